"""Round-3 additions of the C ABI (gendr_params, ABI 5): the deterministic backward, the caller's limit on the coverage
pool (a pool that runs out must change nothing but speed), aggrs_info left unwritten for unlisted tiles."""
import ctypes

import numpy as np
import pytest
import torch

import criteria
import parity
import scenes

pytestmark = pytest.mark.gpu

CASES = [(n, o) for n, o in scenes.OPTION_MATRIX if n in (
    'uniform_prob_softmax', 'uniform_prob_hardrgb', 'gauss_sq_einstein', 'logistic_prob', 'gamma_yager_vertex', 'cauchy_dombi',
    'uniform_T4', 'uniform_T9_clamp', 'hard_hard_hard', 'uniform_singleside')]


def _inputs(opts, maker=scenes.sphere):
    kw = {}
    if opts.get('texture_type') == 'vertex':
        kw['vertex_tex'] = True
    if 'T' in opts:
        kw['T'] = opts['T']
    return maker(**kw)


@pytest.mark.parametrize("name,opts", CASES, ids=[n for n, _ in CASES])
def test_deterministic_backward_is_bit_reproducible_and_correct(oracle_mod, native_lib, name, opts):
    """gendr_params.deterministic = 1: two calls return bit-identical gradients (the default path's float atomics do not
    promise that, experiments/train_reconstruction.py:582-586), the forward pass is untouched, and the gradients pass the
    same element-wise rule against the oracle as the default path's."""
    for maker, isz in ((scenes.sphere, 64), (scenes.soup, 48)):
        fv, tex = _inputs(opts, maker)
        grad = np.random.RandomState(1).randn(fv.shape[0], 4, isz, isz).astype(np.float32)
        d1 = parity.run_hip(fv, tex, isz, dict(opts, deterministic=1), grad)
        d2 = parity.run_hip(fv, tex, isz, dict(opts, deterministic=1), grad)
        for k in ('grad_faces', 'grad_textures'):
            assert np.array_equal(d1[k], d2[k], equal_nan=True), (name, k)
        a = parity.run_hip(fv, tex, isz, opts, grad)
        assert np.array_equal(a['rgba'], d1['rgba'], equal_nan=True) and np.array_equal(a['aggrs_info'], d1['aggrs_info'], equal_nan=True)
        bad, rep, refs = criteria.check_case(fv, tex, isz, opts, d1, grad)
        assert not bad, (name, bad)
        o32 = refs['o32']
        for k, ak in (('grad_faces', 'abs_faces'), ('grad_textures', 'abs_textures')):
            s = parity.stats(d1[k], a[k], scale=o32[ak].reshape(a[k].shape))
            assert s['max_rel'] <= 5e-6, (name, k, s)        # the two paths differ by summation order only


def test_deterministic_backward_is_independent_of_the_batch(native_lib):
    """One writer and one summation order per gradient element: an item's gradients do not depend on what else is in the
    batch (with the clamped texel mode: the reference's texel-index overflow reads the next item's texel otherwise)."""
    fv, tex = scenes.sphere(B=2)
    grad = np.random.RandomState(1).randn(2, 4, 64, 64).astype(np.float32)
    o = dict(deterministic=1, texel_mode=1)
    both = parity.run_hip(fv, tex, 64, o, grad)
    for i in range(2):
        one = parity.run_hip(fv[i:i + 1], tex[i:i + 1], 64, o, grad[i:i + 1])
        assert np.array_equal(one['grad_faces'][0], both['grad_faces'][i])
        assert np.array_equal(one['grad_textures'][0], both['grad_textures'][i])


def test_deterministic_through_the_autograd_function(native_lib, monkeypatch):
    import gendr_amd
    from gendr_amd.synthetic import benchmark_scene
    monkeypatch.setenv('GENDR_DETERMINISTIC', '1')
    fv0, tex0 = benchmark_scene(4, subdivisions=2, device='cuda:0')
    g = torch.randn(4, 4, 96, 96, device='cuda', generator=torch.Generator('cuda').manual_seed(3))
    outs = []
    for _ in range(2):
        fv, tex = fv0.clone().requires_grad_(True), tex0.clone().requires_grad_(True)
        img = gendr_amd.functional.render(fv, tex, image_size=96)
        img.backward(g)
        outs.append((fv.grad.clone(), tex.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("limit", [8, 1000])
@pytest.mark.parametrize("name,opts", CASES[:5], ids=[n for n, _ in CASES[:5]])
def test_exhausted_entry_pool_changes_nothing_but_speed(native_lib, name, opts, limit):
    """gendr_params.pool_entries_max: tiles that find the pool exhausted take the render kernels' own walk over all faces;
    the allocation counter cannot wrap or hand out another region's slice (64-bit, advanced only while the request fits)."""
    fv, tex = _inputs(opts)
    grad = np.random.RandomState(1).randn(fv.shape[0], 4, 64, 64).astype(np.float32)
    a = parity.run_hip(fv, tex, 64, opts, grad)
    b = parity.run_hip(fv, tex, 64, dict(opts, pool_entries_max=limit), grad)
    assert np.array_equal(a['rgba'], b['rgba'], equal_nan=True)
    assert np.array_equal(a['aggrs_info'], b['aggrs_info'], equal_nan=True)
    for k in ('grad_faces', 'grad_textures'):
        scale = max(1e-30, float(np.nanmax(np.abs(a[k]))))
        assert float(np.nanmax(np.abs(a[k] - b[k]))) <= 2e-5 * scale, k


def test_unlisted_aux_is_left_alone_when_asked(native_lib):
    """skip_unlisted_aux = 1 (what the autograd Function sets): RGBA identical, aggrs_info identical wherever a face
    reaches the pixel's tile, untouched (the sentinel survives) where none does; gradients identical."""
    from gendr_amd.functional import renderer as R
    fv, tex = scenes.sphere()
    B, nf = fv.shape[:2]
    isz = 128
    o, extra = parity.split_options({})
    faces = torch.from_numpy(fv).reshape(B, nf, 9).cuda()
    textures = torch.from_numpy(tex).cuda()
    p0 = parity.hip_params(isz, o, extra)
    rgba0, aux0, rec0 = R.native_forward(faces, textures, p0)
    p1 = parity.hip_params(isz, o, dict(extra, skip_unlisted_aux=1))
    sentinel = torch.full((B, 2, isz, isz), -777.0, device='cuda')
    rgba1, aux1, rec1 = R.native_forward(faces, textures, p1, aggrs_info=sentinel.clone())
    assert torch.equal(rgba0, rgba1)
    untouched = aux1 == -777.0
    assert bool(untouched.any()) and bool((~untouched).any())
    assert torch.equal(aux0[~untouched], aux1[~untouched])
    assert bool((rgba0[:, 3][untouched[:, 0]] == 0).all())                  # only pixels no face reaches
    g = torch.randn(B, 4, isz, isz, device='cuda', generator=torch.Generator('cuda').manual_seed(1))
    gf0, gt0 = R.native_backward(faces, textures, rgba0, aux0, rec0, g, p0)
    gf1, gt1 = R.native_backward(faces, textures, rgba1, aux1, rec1, g, p1)
    assert float((gf0 - gf1).abs().max()) <= 2e-5 * float(gf0.abs().max())
    assert float((gt0 - gt1).abs().max()) <= 2e-5 * float(gt0.abs().max())


HINT_CASES = [(n, o) for n, o in scenes.OPTION_MATRIX if n in (
    'uniform_prob_softmax', 'uniform_prob_hardrgb', 'hard_hard_hard', 'gauss_sq_einstein', 'logistic_prob', 'gamma_yager_vertex',
    'uniform_T4', 'uniform_singleside', 'uniform_smalleps', 'laplace_frank')]


@pytest.mark.parametrize("name,opts", HINT_CASES, ids=[n for n, _ in HINT_CASES])
def test_pair_hints_change_nothing_but_speed(oracle_mod, native_lib, name, opts):
    """gendr_params.pair_hints (ABI 6): with the forward kernel's hints the backward kernel evaluates the ONE edge the
    closest-point search selected in the forward pass instead of repeating the search -- the same float operations for that
    edge.  Forward results are untouched; the gradients equal those of the hint-less call up to the order of the float
    atomics (both calls use them), and pass the element-wise rule against the oracle."""
    for maker, isz in ((scenes.sphere, 64), (scenes.soup, 48), (scenes.slivers, 48)):
        fv, tex = _inputs(opts, maker)
        grad = np.random.RandomState(2).randn(fv.shape[0], 4, isz, isz).astype(np.float32)
        on = parity.run_hip(fv, tex, isz, dict(opts, pair_hints=1), grad)
        off = parity.run_hip(fv, tex, isz, dict(opts, pair_hints=-1), grad)
        assert np.array_equal(on['rgba'], off['rgba'], equal_nan=True) and np.array_equal(on['aggrs_info'], off['aggrs_info'], equal_nan=True)
        bad, rep, refs = criteria.check_case(fv, tex, isz, opts, on, grad)
        assert not bad, (name, bad)
        o32 = refs['o32']
        for k, ak in (('grad_faces', 'abs_faces'), ('grad_textures', 'abs_textures')):
            s = parity.stats(on[k], off[k], scale=o32[ak].reshape(on[k].shape))
            assert s['max_rel'] <= 2e-6, (name, k, s)


def test_pair_hints_survive_a_pair_no_hint_can_describe(oracle_mod, native_lib):
    """A pixel inside a face whose three closest-point candidates all lie 1e4 units away: none of them beats the running
    minimum 1e8 of kernel.cu:86,112 and the reference leaves dx = dy = t = 0.  No 2-bit hint says that; the forward kernel
    flags the tile queue instead and backward repeats the whole search there.  Same gradients as without hints, and the
    oracle's."""
    fv, tex = scenes.sphere(B=2)
    fv = fv.copy()
    fv[:, 0, :, :2] = [[-3e4, -3e4], [3e4, -3e4], [0., 3e4]]       # covers everything, every edge farther than sqrt(1e8)
    fv[:, 0, :, 2] = 50.0                                           # behind the sphere
    isz = 64
    grad = np.random.RandomState(3).randn(2, 4, isz, isz).astype(np.float32)
    for opts in (dict(), dict(aggr_rgb_func='hard'), dict(dist_func='gaussian', dist_scale=3e-3, dist_squared=True, aggr_alpha_func='einstein')):
        on = parity.run_hip(fv, tex, isz, dict(opts, pair_hints=1), grad)
        off = parity.run_hip(fv, tex, isz, dict(opts, pair_hints=-1), grad)
        assert np.array_equal(on['rgba'], off['rgba'], equal_nan=True)
        bad, rep, refs = criteria.check_case(fv, tex, isz, opts, on, grad)
        assert not bad, bad
        o32 = refs['o32']
        for k, ak in (('grad_faces', 'abs_faces'), ('grad_textures', 'abs_textures')):
            s = parity.stats(on[k], off[k], scale=o32[ak].reshape(on[k].shape))
            assert s['max_rel'] <= 2e-6, (k, s)


def test_backward_after_face_setup_alone_takes_no_stale_hints(native_lib):
    """The hints belong to the gendr_forward call that filled the workspace.  The pybind-shaped backward_render rebuilds the
    workspace with gendr_face_setup (no forward render): the setup kernel clears the per-queue flag, so backward must not
    read the hints an EARLIER forward call left in the same memory for a different mesh."""
    from gendr_amd.functional import renderer as R
    from gendr_amd import _native
    L = _native.lib()
    isz = 64
    o, extra = parity.split_options(dict(pair_hints=1))
    p = parity.hip_params(isz, o, extra)
    fa, ta = scenes.sphere(B=2)
    fb, tb = scenes.soup(B=2, nf=fa.shape[1])
    dev = 'cuda:0'
    A = torch.from_numpy(fa).reshape(2, -1, 9).to(dev).contiguous(); TA = torch.from_numpy(ta).to(dev).contiguous()
    Bf = torch.from_numpy(fb).reshape(2, -1, 9).to(dev).contiguous(); TB = torch.from_numpy(tb).to(dev).contiguous()
    g = torch.randn(2, 4, isz, isz, device=dev, generator=torch.Generator(dev).manual_seed(5))
    rgba_b, aux_b, ws_b = R.native_forward(Bf, TB, p)
    want_f, want_t = R.native_backward(Bf, TB, rgba_b, aux_b, ws_b, g, p)
    # a forward call on mesh A leaves ITS hints in a workspace; then only the setup stage runs on mesh B in the same memory
    rgba_a, aux_a, ws = R.native_forward(A, TA, p)
    torch.cuda.synchronize()
    st = torch.cuda.current_stream().cuda_stream
    e = L.gendr_face_setup(ctypes.c_void_p(Bf.data_ptr()), ctypes.c_void_p(TB.data_ptr()), ctypes.c_void_p(ws.data_ptr()),
                           2, Bf.shape[1], 1, ctypes.byref(p), ctypes.c_void_p(st))
    assert e == 0
    got_f, got_t = R.native_backward(Bf, TB, rgba_b, aux_b, ws, g, p)
    scale = float(want_f.abs().max())
    assert float((got_f - want_f).abs().max()) <= 1e-5 * scale
    assert float((got_t - want_t).abs().max()) <= 1e-5 * max(1e-30, float(want_t.abs().max()))


LOOSE_CASES = [(n, o) for n, o in scenes.OPTION_MATRIX if n in (
    'uniform_prob_softmax', 'uniform_prob_hardrgb', 'hard_hard_hard', 'gauss_sq_einstein', 'logistic_prob', 'uniform_smalleps',
    'uniform_singleside', 'gamma_yager_vertex')]


@pytest.mark.parametrize("name,opts", LOOSE_CASES, ids=[n for n, _ in LOOSE_CASES])
def test_loose_face_boxes_change_nothing(native_lib, name, opts):
    """gendr_params.loose_faces (ABI 6): a face whose cull box is loose (error bound >> cull radius: the sliver and soup scenes are
    full of them) is evaluated by the coverage kernel on the pixels of every tile its loose box meets and keeps exactly the
    pixels that can contribute -- instead of all 64 of every such tile.  On by default at every size since round 4: forward
    results bit-identical to the call without it (loose_faces = -1) AND to the all-pairs traversal; gradients equal up to
    atomics order."""
    for maker, isz in ((scenes.slivers, 64), (scenes.soup, 48), (scenes.sphere, 64)):
        fv, tex = _inputs(opts, maker)
        grad = np.random.RandomState(4).randn(fv.shape[0], 4, isz, isz).astype(np.float32)
        on = parity.run_hip(fv, tex, isz, dict(opts, loose_faces=1), grad)
        off = parity.run_hip(fv, tex, isz, dict(opts, loose_faces=-1), grad)
        # loose_faces = 2: the per-image lists + loose_faces_kernel + the binning kernel narrowing by the live pixels' box -- the path
        # images of 1024^2 and more (BASELINE config 5) take by default, forced onto these small images
        lists = parity.run_hip(fv, tex, isz, dict(opts, loose_faces=2), grad)
        allp = parity.run_hip(fv, tex, isz, dict(opts, cull=0))
        for k in ('rgba', 'aggrs_info'):
            assert np.array_equal(on[k], off[k], equal_nan=True), (name, maker.__name__, k)
            assert np.array_equal(on[k], allp[k], equal_nan=True), (name, maker.__name__, k, 'all pairs')
            assert np.array_equal(lists[k], allp[k], equal_nan=True), (name, maker.__name__, k, 'list path vs all pairs')
        for k in ('grad_faces', 'grad_textures'):
            scale = max(1e-30, float(np.abs(off[k]).max()))
            assert float(np.abs(on[k] - off[k]).max()) <= 2e-5 * scale, (name, maker.__name__, k)
            assert float(np.abs(lists[k] - off[k]).max()) <= 2e-5 * scale, (name, maker.__name__, k, 'list path')


@pytest.mark.parametrize("mode", [1, 2])
def test_loose_faces_survive_graph_replay(native_lib, mode):
    """Loose faces are resolved inside the coverage kernel (mode 1); images of 1024^2 and more -- and mode 2 at any size -- also keep
    per-image lists in the workspace that the binning kernel empties after use, since a replayed graph (same kernel arguments, same
    workspace, over and over) would otherwise make them grow: forty replays of a captured forward call return the first call's
    image."""
    from gendr_amd.functional import renderer as R
    fv, tex = scenes.slivers()
    isz = 64
    o, extra = parity.split_options(dict(loose_faces=mode))
    p = parity.hip_params(isz, o, extra)
    Bn, nf = fv.shape[:2]
    faces = torch.from_numpy(fv).reshape(Bn, nf, 9).cuda().contiguous()
    t = torch.from_numpy(tex).cuda().contiguous()
    want, _, _ = R.native_forward(faces, t, p)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        R.native_forward(faces, t, p)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        rgba, aux, ws = R.native_forward(faces, t, p)
    for _ in range(40):
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(rgba, want)
