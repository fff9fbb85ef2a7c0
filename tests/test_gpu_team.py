"""Team kernels (gendr_amd/csrc/gendr_team.h, gendr_params.team, ABI 7): one workgroup per tile -- the tile's pair list built once in
shared LDS, eight wavefronts evaluating its batches side by side, (forward) a ninth folding their results per pixel in the
reference's order.  Held to: the one-wave kernels and the all-pairs traversal bit for bit (forward), the oracle by the element-wise
rule, the reference's own kernels by the flat 1e-5 gate, on scenes that exercise the team proper (thousands of pairs per tile, several
build phases and chunks), its solo paths (pixel-mode tiles, tiles without a slice of the entry pool) and the pair hints."""
import ctypes

import numpy as np
import pytest
import torch

import criteria
import parity
import pin
import scenes

pytestmark = pytest.mark.gpu

SOFT = dict(dist_func='logistic', dist_scale=1e-2, aggr_alpha_func='probabilistic', aggr_rgb_func='hard', dist_eps=100., double_side=False)


def _uses_team(B, nf, T, isz, opts, silhouette=0):
    from gendr_amd import _native
    o, extra = parity.split_options(opts)
    p = parity.hip_params(isz, o, extra)
    return _native.lib().gendr_uses_team(B, nf, T, ctypes.byref(p), silhouette)


def test_the_rule(native_lib):
    """Automatic: an option set with a team kernel and at most 4096 tiles, or 8192 with a cull radius of 2 pixels and more."""
    assert _uses_team(24, 1280, 1, 64, SOFT) == 1                                   # opt_shape.py: 1536 tiles, 4.4 pixels
    assert _uses_team(24, 1280, 1, 64, dict(SOFT, aggr_rgb_func='softmax')) == 1
    assert _uses_team(24, 1280, 1, 64, SOFT, silhouette=1) == 1
    assert _uses_team(24, 1280, 1, 64, dict(SOFT, team=-1)) == 0
    assert _uses_team(24, 1280, 1, 64, dict(SOFT, dist_scale=1e-4)) == 1           # 0.04 pixels: few tiles are enough
    assert _uses_team(64, 1280, 1, 64, dict(SOFT, dist_scale=1e-4)) == 1 and _uses_team(65, 1280, 1, 64, dict(SOFT, dist_scale=1e-4)) == 0
    assert _uses_team(128, 1280, 1, 64, SOFT) == 1 and _uses_team(129, 1280, 1, 64, SOFT) == 0     # 8192 tiles and one more image
    assert _uses_team(256, 1280, 1, 512, dict(SOFT, dist_eps=1e4)) == 0             # BASELINE config 4: a million tiles
    assert _uses_team(256, 1280, 1, 512, dict(SOFT, dist_eps=1e4, team=1)) == 1     # ... forced
    assert _uses_team(24, 1280, 1, 64, dict(SOFT, dist_func='gaussian')) == 1       # the runtime-dispatch team kernel: light distributions
    assert _uses_team(24, 1280, 1, 64, dict(SOFT, dist_func='gamma', dist_shape=2.)) == 0          # ... not the heavy ones,
    assert _uses_team(24, 1280, 1, 64, dict(SOFT, dist_func='gamma', dist_shape=2., team=1)) == 0
    assert _uses_team(24, 1280, 1, 64, dict(SOFT, aggr_alpha_func='yager', aggr_alpha_t_conorm_p=2.)) == 0   # nor the heavy aggregators,
    assert _uses_team(24, 1280, 3, 64, dict(SOFT, dist_func='gaussian', texture_type='vertex')) == 0         # nor vertex textures
    assert _uses_team(24, 1280, 1, 64, dict(SOFT, dist_func='cauchy')) == 0         # no cull radius: no entry pool, nothing for a team to walk
    assert _uses_team(4, 1280, 1, 256, dict()) == 0 and _uses_team(4, 1280, 1, 256, dict(team=1)) == 1   # BASELINE config 2's option set keeps
    assert _uses_team(24, 1280, 1, 64, dict(SOFT, dist_func='uniform')) == 0        # its specialised one-wave kernel, train_reconstruction.py's too
    assert _uses_team(24, 1280, 1, 64, dict(SOFT, cull=0)) == 0
    assert _uses_team(24, 1280, 1, 64, dict(SOFT, deterministic=1)) == 0


SCENES = [
    ('sphere1280', lambda: parity_scene(4)),
    ('sphere80', lambda: scenes.sphere(B=3)),
    ('soup', lambda: scenes.soup(B=2, nf=96)),
    ('slivers', lambda: scenes.slivers(B=2)),
]


def parity_scene(B):
    from gendr_amd.synthetic import benchmark_scene
    fv, tex = benchmark_scene(B)
    return fv.numpy(), tex.numpy()


def _same_forward(a, b, what):
    assert np.array_equal(a['rgba'], b['rgba'], equal_nan=True), 'rgba: team vs ' + what
    assert np.array_equal(a['aggrs_info'], b['aggrs_info'], equal_nan=True), 'aggrs_info: team vs ' + what


def _close_grads(a, b, what, tol=2e-5):
    for k in ('grad_faces', 'grad_textures'):
        assert float(np.abs(a[k] - b[k]).max()) <= tol * max(1e-30, float(np.abs(b[k]).max())), (k, what)


@pytest.mark.parametrize("rgb", ['hard', 'softmax'])
@pytest.mark.parametrize("name,make", SCENES, ids=[s[0] for s in SCENES])
def test_team_changes_nothing(oracle_mod, native_lib, name, make, rgb):
    fv, tex = make()
    isz = 64
    # sigma 2e-2 on the small scenes: tails of 9 pixels, hundreds to thousands of pairs per tile
    opts = dict(SOFT, aggr_rgb_func=rgb, dist_scale=1e-2 if name == 'sphere1280' else 2e-2, double_side=(name != 'sphere1280'))
    grad = np.random.RandomState(5).randn(fv.shape[0], 4, isz, isz).astype(np.float32)
    assert _uses_team(fv.shape[0], fv.shape[1], 1, isz, dict(opts, team=1)) == 1
    t = parity.run_hip(fv, tex, isz, dict(opts, team=1), grad)
    w = parity.run_hip(fv, tex, isz, dict(opts, team=-1), grad)                # one wave per tile (piece)
    n = parity.run_hip(fv, tex, isz, dict(opts, cull=0), grad)                 # the reference's own traversal
    _same_forward(t, w, 'one-wave kernels')
    _same_forward(t, n, 'all pairs')
    _close_grads(t, w, 'one-wave kernels')
    _close_grads(t, n, 'all pairs')
    bad, _, _ = criteria.check_case(fv, tex, isz, opts, t, grad)
    assert not bad, bad


GENERIC = [
    ('gaussian_prob_hard', dict(dist_func='gaussian', dist_scale=1e-2, aggr_rgb_func='hard')),
    ('gauss_sq_einstein_softmax', dict(dist_func='gaussian', dist_squared=True, dist_scale=3e-4, aggr_alpha_func='einstein', aggr_rgb_func='softmax')),
    ('laplace_hamacher_hard', dict(dist_func='laplace', dist_scale=1e-2, aggr_alpha_func='hamacher', aggr_alpha_t_conorm_p=0.5, aggr_rgb_func='hard')),
    ('cubic_max_softmax', dict(dist_func='cubic_hermite', dist_scale=5e-2, aggr_alpha_func='max', aggr_rgb_func='softmax')),
    ('uniform_max_hard', dict(dist_func='uniform', dist_scale=3e-2, aggr_alpha_func='max', aggr_rgb_func='hard')),
    ('gumbelmin_prob_softmax', dict(dist_func='gumbel_min', dist_scale=1e-2, aggr_rgb_func='softmax')),
]


@pytest.mark.parametrize("name,o", GENERIC, ids=[g[0] for g in GENERIC])
def test_runtime_dispatch_team_kernel(oracle_mod, native_lib, name, o):
    """The team kernel of the 13 light distributions x 5 light aggregators (what opt_shape.py renders with any other --dist-func /
    --aggr-func of those): against the one-wave runtime-dispatch kernels, the all-pairs traversal and the oracle."""
    fv, tex = parity_scene(3)
    isz = 64
    opts = dict(dist_eps=100., double_side=False, **o)
    grad = np.random.RandomState(9).randn(3, 4, isz, isz).astype(np.float32)
    assert _uses_team(3, fv.shape[1], 1, isz, dict(opts, team=1)) == 1         # (forced: BASELINE config 3's option set has a specialised one-wave kernel)
    t = parity.run_hip(fv, tex, isz, dict(opts, team=1), grad)
    w = parity.run_hip(fv, tex, isz, dict(opts, team=-1), grad)
    n = parity.run_hip(fv, tex, isz, dict(opts, cull=0), grad)
    _same_forward(t, w, 'one-wave kernels')
    _same_forward(t, n, 'all pairs')
    _close_grads(t, w, 'one-wave kernels')
    _close_grads(t, n, 'all pairs')
    bad, _, _ = criteria.check_case(fv, tex, isz, opts, t, grad)
    assert not bad, bad


@pytest.mark.parametrize("hints", [1, -1])
def test_team_with_and_without_pair_hints(native_lib, hints):
    fv, tex = parity_scene(3)
    isz = 64
    grad = np.random.RandomState(6).randn(3, 4, isz, isz).astype(np.float32)
    t = parity.run_hip(fv, tex, isz, dict(SOFT, team=1, pair_hints=hints), grad)
    w = parity.run_hip(fv, tex, isz, dict(SOFT, team=-1, pair_hints=-1), grad)
    _same_forward(t, w, 'one-wave kernels')
    _close_grads(t, w, 'one-wave kernels')


def test_team_pixel_mode_and_solo_paths(oracle_mod, native_lib):
    """Pixel-mode tiles (entries of 36 pixels and more on average: sigma 3e-2 on the 80-face sphere) are rendered with lane = pixel,
    an entry per B-wave; tiles without a slice of the entry pool (pool capped at a few entries) by one wave of the team."""
    fv, tex = scenes.sphere(B=2)
    isz = 64
    grad = np.random.RandomState(7).randn(2, 4, isz, isz).astype(np.float32)
    for extra in (dict(), dict(pool_entries_max=512)):
        opts = dict(SOFT, dist_scale=3e-2, aggr_rgb_func='softmax', dist_eps=1e4, **extra)
        t = parity.run_hip(fv, tex, isz, dict(opts, team=1), grad)
        n = parity.run_hip(fv, tex, isz, dict(opts, cull=0), grad)
        _same_forward(t, n, 'all pairs')
        _close_grads(t, n, 'all pairs')
        bad, _, _ = criteria.check_case(fv, tex, isz, opts, t, grad)
        assert not bad, bad


def test_team_at_odd_sizes_and_single_view(native_lib):
    """Image sizes that are no multiple of the tile (clipped tiles, invalid lanes), one view (the 8 queues are bands of one image)."""
    fv, tex = parity_scene(1)
    for isz in (36, 61, 100):
        grad = np.random.RandomState(isz).randn(1, 4, isz, isz).astype(np.float32)
        t = parity.run_hip(fv, tex, isz, dict(SOFT, team=1), grad)
        n = parity.run_hip(fv, tex, isz, dict(SOFT, cull=0), grad)
        _same_forward(t, n, 'all pairs at %d' % isz)
        _close_grads(t, n, 'all pairs at %d' % isz)


def test_team_against_reference_kernels(oracle_mod, native_lib):
    """opt_shape.py's soft renderer at its own shape (24 views, 64^2, 1280 faces), both rgb aggregations, flat 1e-5 against the
    reference's kernels, rgba bit-identical (logistic / probabilistic)."""
    parity.require_reference()
    fv, tex = parity_scene(24)
    isz = 64
    grad = np.random.RandomState(8).randn(24, 4, isz, isz).astype(np.float32)
    for rgb in ('hard', 'softmax'):
        opts = dict(SOFT, aggr_rgb_func=rgb)
        assert _uses_team(24, fv.shape[1], 1, isz, opts) == 1                   # automatic at this shape
        for variant in ('default', 'exact'):
            h = parity.run_hip(fv, tex, isz, opts, grad, variant=variant)
            r = parity.run_reference(fv, tex, isz, opts, grad, np.float32)
            c = parity.run_oracle(fv, tex, isz, opts, grad, np.float32)
            assert np.array_equal(h['rgba'], r['rgba']), (rgb, variant)
            m = pin.measure(h, r, c['abs_faces'], c['abs_textures'])
            assert all(x['max'] <= 1e-5 for x in m.values()), (rgb, variant, m)


def test_team_silhouette(native_lib):
    """The alpha-only entry points: alpha == render()[:, 3] bit for bit, the vertex gradient of an alpha-only loss agrees."""
    from gendr_amd.functional.silhouette import render_silhouette
    from gendr_amd.functional.renderer import render
    import os
    fv, tex = parity_scene(4)
    f = torch.from_numpy(fv).cuda()
    t = torch.from_numpy(tex).cuda()
    isz = 64
    kw = dict(dist_func='logistic', dist_scale=1e-2, dist_eps=100., aggr_alpha_func='probabilistic')
    res = {}
    for team in ('1', '-1'):
        os.environ['GENDR_TEAM'] = team
        try:
            fa = f.clone().requires_grad_(True)
            al = render_silhouette(fa, isz, **kw)
            al.square().sum().backward()
            fb = f.clone().requires_grad_(True)
            full = render(fb, t, isz, [0., 0., 0.], aggr_rgb_func='hard', double_side=False, **kw)
            full[:, 3].square().sum().backward()
            res[team] = (al.detach().cpu().numpy(), fa.grad.cpu().numpy(), full[:, 3].detach().cpu().numpy(), fb.grad.cpu().numpy())
        finally:
            del os.environ['GENDR_TEAM']
    a1, g1, f1, h1 = res['1']
    a0, g0, f0, h0 = res['-1']
    assert np.array_equal(a1, a0) and np.array_equal(a1, f1) and np.array_equal(f1, f0)
    for x, y in ((g1, g0), (g1, h1), (h1, h0)):
        assert float(np.abs(x - y).max()) <= 2e-5 * float(np.abs(y).max())
