"""The option sets and shapes the reference's OWN scripts render, pinned to the reference's own kernels (oracle/_ref):

  experiments/opt_shape.py:134-159        soft renderer: 24 views of 64^2, logistic / probabilistic / HARD rgb, dist_eps 100, the
                                          sigma of its sweep (np.logspace(-1, -7, 7), :327); hard renderer: dist_func 0,
                                          aggr_alpha_func 0, hard rgb, dist_squared=True, dist_scale 1e-4, dist_eps 1
  experiments/train_reconstruction.py     uniform, tau = 10^-1.5, probabilistic, hard rgb, dist_eps 300 (:518, :557), 64^2
  :181-196

on the benchmark's 1280-face mesh seen from its camera ring.  Small dist_eps is the regime in which the reference's border
test (kernel.cu:747), not the distribution's tail, ends a face's reach -- where a coverage defect hid for a whole round
(VERDICT r4 weak 2).  Gate: the FLAT 1e-5 of tests/pin.py on every element of every tensor, both build variants, forward also
culled == all-pairs bit for bit; the kernels these option sets run on are the specialised ones of gendr_capi.hip's table
(round 5) -- `test_specialised_rows_cover_the_scripts` holds that."""
import numpy as np
import pytest

import parity
import pin

pytestmark = pytest.mark.gpu

SIGMAS = (1e-1, 1e-2, 1e-4, 1e-7)
CASES = [('opt_shape_soft_sigma%g' % s,
          dict(dist_func='logistic', dist_scale=s, dist_squared=False, dist_shape=0., dist_shift=0., dist_eps=100,
               aggr_alpha_func='probabilistic', aggr_alpha_t_conorm_p=0., aggr_rgb_func='hard', double_side=False), 24) for s in SIGMAS]
CASES += [
    ('opt_shape_hard', dict(dist_func=0, dist_scale=1e-4, dist_squared=True, dist_shape=0., dist_shift=0., dist_eps=1,
                            aggr_alpha_func=0, aggr_alpha_t_conorm_p=0., aggr_rgb_func='hard', double_side=False), 24),
    ('train_reconstruction', dict(dist_func='uniform', dist_scale=10 ** -1.5, dist_squared=False, dist_shape=0, dist_shift=0,
                                  dist_eps=300., aggr_alpha_func='probabilistic', aggr_alpha_t_conorm_p=0, aggr_rgb_func='hard',
                                  double_side=False), 8),
    # the same option sets in the regime the fuzz campaign found a defect in: the border test decides
    ('train_reconstruction_eps1', dict(dist_func='uniform', dist_scale=10 ** -1.5, dist_eps=1., aggr_alpha_func='probabilistic',
                                       aggr_rgb_func='hard', double_side=False), 8),
    ('opt_shape_soft_eps3', dict(dist_func='logistic', dist_scale=1e-2, dist_eps=3., aggr_alpha_func='probabilistic',
                                 aggr_rgb_func='hard', double_side=False), 8),
]
ISZ = 64


@pytest.fixture(scope='module')
def ref_kernels():
    parity.require_reference()


def _scene(B):
    from gendr_amd.synthetic import benchmark_scene
    fv, tex = benchmark_scene(B)
    return fv.numpy(), tex.numpy()


@pytest.mark.parametrize("variant", ('default', 'exact'))
@pytest.mark.parametrize("name,opts,B", CASES, ids=[c[0] for c in CASES])
def test_script_option_sets_against_reference_kernels(oracle_mod, native_lib, ref_kernels, name, opts, B, variant):
    fv, tex = _scene(B)
    grad = np.random.RandomState(3).randn(B, 4, ISZ, ISZ).astype(np.float32)
    r = parity.run_reference(fv, tex, ISZ, opts, grad, np.float32)
    c = parity.run_oracle(fv, tex, ISZ, opts, grad, np.float32)
    h = parity.run_hip(fv, tex, ISZ, opts, grad, variant=variant)
    bad = pin.flat_failures(name, pin.measure(h, r, c['abs_faces'], c['abs_textures']), None, section=variant)
    assert not bad, bad
    # alpha: these option sets call at most expf -- on the algebraic ones bit for bit
    if opts['dist_func'] in (0, 'uniform'):
        assert np.array_equal(h['rgba'][:, 3], r['rgba'][:, 3]), 'alpha must be bit-identical to the reference kernels'
    h0 = parity.run_hip(fv, tex, ISZ, dict(opts, cull=0), None, variant=variant)
    for k in ('rgba', 'aggrs_info'):
        assert np.array_equal(h[k], h0[k], equal_nan=True), (name, k, 'culled != all-pairs')


def test_silhouette_path_at_the_scripts_option_sets(oracle_mod, native_lib, ref_kernels):
    """What both scripts consume is channel 3 only (opt_shape.py:257, train_reconstruction.py:230): the alpha-only kernels
    (SURVEY f-4) at the same option sets, bit for bit the full render's alpha and within the flat gate of the reference's."""
    import torch
    from gendr_amd.functional import silhouette as S
    for name, opts, B in CASES:
        fv, tex = _scene(B)
        o = {k: v for k, v in opts.items() if k not in ('aggr_rgb_func', 'double_side')}      # (alpha does not depend on either)
        fvt = torch.from_numpy(fv).cuda()
        a = S.render_silhouette(fvt, image_size=ISZ, **o)
        h = parity.run_hip(fv, tex, ISZ, opts, None)
        assert np.array_equal(a.cpu().numpy(), h['rgba'][:, 3]), name
