"""The vectorised pure-PyTorch restatement (oracle/torch_ref.py) against the C oracle: two independently
structured implementations of the same published arithmetic have to agree -- bit for bit where no libm call
is involved.  Also BASELINE.json config 1 (unit quad, 64x64, uniform / probabilistic, batch 1, CPU only)."""
import numpy as np
import pytest
import torch

import parity
import scenes
from oracle import torch_ref

CASES = [
    ('uniform_prob_softmax', dict(), True),
    ('uniform_prob_hardrgb', dict(aggr_rgb_func='hard'), True),
    ('hard_hard_hard', dict(dist_func='hard', aggr_alpha_func='hard', aggr_rgb_func='hard'), True),
    ('uniform_max_single', dict(aggr_alpha_func='max', double_side=False), True),
    ('uniform_einstein_vertex', dict(aggr_alpha_func='einstein', texture_type='vertex'), True),
    ('uniform_clamp', dict(texel_mode=1), True),
    ('uniform_smalleps', dict(dist_eps=1.5, dist_scale=2e-2), True),
    ('logistic_prob', dict(dist_func='logistic', dist_scale=2e-2), False),
    ('gauss_sq_einstein', dict(dist_func='gaussian', dist_squared=True, dist_scale=3e-3, aggr_alpha_func='einstein'), False),
]


def _run_both(fv, tex, isz, opts, dtype):
    grad = np.random.RandomState(3).randn(fv.shape[0], 4, isz, isz).astype(dtype)
    c = parity.run_oracle(fv.astype(dtype), tex.astype(dtype), isz, opts, grad, dtype)
    kw = {k: v for k, v in opts.items() if k != 'T'}
    t = torch_ref.render(torch.from_numpy(fv.astype(dtype)), torch.from_numpy(tex.astype(dtype)), isz,
                         grad=torch.from_numpy(grad), **kw)
    return c, {k: v.numpy() for k, v in t.items()}


@pytest.mark.parametrize("name,opts,algebraic", CASES, ids=[c[0] for c in CASES])
def test_matches_c_oracle_fp32(oracle_mod, name, opts, algebraic):
    vertex = opts.get('texture_type') == 'vertex'
    for maker in (scenes.soup, scenes.sphere):
        fv, tex = maker(B=2, vertex_tex=vertex) if maker is scenes.sphere else maker(B=2, nf=24, vertex_tex=vertex)
        c, t = _run_both(fv, tex, 24, opts, np.float32)
        if algebraic:
            assert np.array_equal(c['rgba'][:, 3], t['rgba'][:, 3], equal_nan=True), 'alpha must agree bit for bit'
        if algebraic and opts.get('aggr_rgb_func') == 'hard':
            assert np.array_equal(c['rgba'], t['rgba'], equal_nan=True)
            assert np.array_equal(c['aggrs_info'], t['aggrs_info'], equal_nan=True)
        # exp / erfc come from different libms (glibc vs torch's vectorised kernels)
        s = parity.stats(t['rgba'], c['rgba'])
        assert s['p99_rel'] <= 1e-5 and s['frac_gt_1e5'] <= 2e-2, s
        for k, absk in (('grad_faces', 'abs_faces'), ('grad_textures', 'abs_textures')):
            got = t[k].reshape(c[k].shape)
            s = parity.stats(got, c[k], scale=c[absk])
            assert s['p99_rel'] <= 1e-4, (k, s)


@pytest.mark.parametrize("name,opts,algebraic", CASES[:3] + CASES[7:8], ids=[c[0] for c in CASES[:3] + CASES[7:8]])
def test_matches_c_oracle_fp64(oracle_mod, name, opts, algebraic):
    fv, tex = scenes.sphere(B=1)
    c, t = _run_both(fv, tex, 24, opts, np.float64)
    assert np.allclose(t['rgba'], c['rgba'], rtol=1e-9, atol=1e-12, equal_nan=True)
    assert np.allclose(t['grad_faces'].reshape(c['grad_faces'].shape), c['grad_faces'], rtol=1e-6, atol=1e-9)
    assert np.allclose(t['grad_textures'], c['grad_textures'], rtol=1e-6, atol=1e-9)


def _matrix_inputs(opts, scene):
    kw = {}
    if opts.get('texture_type') == 'vertex':
        kw['vertex_tex'] = True
    if 'T' in opts:
        kw['T'] = opts['T']
    if scene == 'soup':
        return scenes.soup(B=2, nf=24, **kw)
    if scene == 'slivers':
        return scenes.slivers(B=1, nf=36, **kw)
    return scenes.sphere(B=2, **kw)


@pytest.mark.parametrize("scene", ['soup', 'sphere', 'slivers'])
@pytest.mark.parametrize("name,opts", scenes.OPTION_MATRIX, ids=[n for n, _ in scenes.OPTION_MATRIX])
def test_whole_option_matrix_fp64(oracle_mod, name, opts, scene):
    """Every distribution, every t-conorm, every texture mode, forward and backward: the two restatements of
    kernel.cu agree in float64, where libm differences (glibc vs torch's vectorised kernels) are 1e-16 and only an
    actual difference in the restated formulas, promotions or skip logic would show."""
    fv, tex = _matrix_inputs(opts, scene)
    c, t = _run_both(fv, tex, 24, opts, np.float64)
    # cauchy calls atanf -- float, whatever scalar_t is (kernel.cu:258): the two libms' float results differ in the last bit
    rtol, gtol = (2e-5, 1e-4) if opts.get('dist_func') == 'cauchy' else (1e-7, 1e-7)
    # pixels that sit exactly on a skip threshold (D <= 1e-6, d^2 >= eps * tau) may flip with a 1e-16 libm difference
    for k in ('rgba', 'aggrs_info'):
        bad = ~np.isclose(t[k], c[k], rtol=rtol, atol=1e-10, equal_nan=True)
        assert bad.mean() <= 2e-3, (k, float(bad.mean()), float(np.nanmax(np.abs(t[k] - c[k]))))
    for k, absk in (('grad_faces', 'abs_faces'), ('grad_textures', 'abs_textures')):
        got = t[k].reshape(c[k].shape)
        s = parity.stats(got, c[k], scale=c[absk])
        assert s['p99_rel'] <= gtol, (k, s)


@pytest.mark.parametrize("name,opts", scenes.OPTION_MATRIX, ids=[n for n, _ in scenes.OPTION_MATRIX])
def test_whole_option_matrix_fp32(oracle_mod, name, opts):
    """float32: same promotions, different libm -- bulk agreement at the 1e-4 level on the well-conditioned scene, or
    (formulas that cancel: 1 - exp(-e^u), 1 - y, Frank) within the oracle's own float32-vs-float64 spread."""
    fv, tex = _matrix_inputs(opts, 'sphere')
    c, t = _run_both(fv, tex, 24, opts, np.float32)
    c64, _ = (parity.run_oracle(fv.astype(np.float64), tex.astype(np.float64), 24, opts,
                                np.random.RandomState(3).randn(fv.shape[0], 4, 24, 24), np.float64), None)
    s = parity.stats(t['rgba'], c['rgba'])
    n = parity.stats(c['rgba'], c64['rgba'])
    assert s['p99_rel'] <= max(1e-4, 4 * n['p99_rel']), (s, n)
    for k, absk in (('grad_faces', 'abs_faces'), ('grad_textures', 'abs_textures')):
        got = t[k].reshape(c[k].shape)
        s = parity.stats(got, c[k], scale=c[absk])
        n = parity.stats(c[k], c64[k], scale=c64[absk])
        assert s['p99_rel'] <= max(1e-3, 4 * n['p99_rel']), (k, s, n)


def test_baseline_config1_unit_quad():
    from gendr_amd.synthetic import unit_quad
    fv, tex = unit_quad()
    out = torch_ref.render(fv, tex, 64)
    a = out['rgba'][0, 3]
    assert float(a.max()) == 1.0 and float(a.min()) == 0.0
    assert abs(float(a.sum()) - (32 * 32 - 32 * 0.25)) < 1e-4
    assert out['rgba'].shape == (1, 4, 64, 64) and out['aggrs_info'].shape == (1, 2, 64, 64)


_REFERENCE_VECTORS = sorted(__import__('glob').glob(__import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)),
                                                                              'golden', 'reference', '*.npz')))


@pytest.mark.parametrize("path", _REFERENCE_VECTORS, ids=[p.split('/')[-1][:-4] for p in _REFERENCE_VECTORS])
def test_torch_restatement_reproduces_reference_vectors_f64(path):
    """The second restatement against outputs of the reference's own kernels (tests/golden/make_reference_golden.py),
    float64: independent of the C oracle."""
    import json
    z = np.load(path)
    opts = json.loads(str(z['options']))
    isz = int(z['image_size'])
    kw = {k: v for k, v in opts.items() if k != 'T'}
    t = torch_ref.render(torch.from_numpy(z['fv'].astype(np.float64)), torch.from_numpy(z['tex'].astype(np.float64)), isz,
                         grad=torch.from_numpy(z['grad'].astype(np.float64)), **kw)
    tol = 5e-5 if opts.get('dist_func') == 'cauchy' else 1e-7
    for k in ('rgba', 'aggrs_info'):
        assert parity.rel_error(t[k].numpy(), z['f64_' + k]).max() <= tol, k
    for k in ('grad_faces', 'grad_textures'):
        ref = z['f64_' + k]
        e = parity.rel_error(t[k].numpy().reshape(ref.shape), ref, floor=parity.GRAD_FLOOR)
        # no per-element sum of magnitudes here: measured against the tensor's largest element
        cauchy = opts.get('dist_func') == 'cauchy'            # float atanf under cancellation
        assert e.max() <= (5e-3 if cauchy else 1e-4) and np.percentile(e, 99) <= max(tol, 1e-7) * 10, (k, float(e.max()))
