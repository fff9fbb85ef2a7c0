"""Host-side behaviour of the drop-in Python surface (no GPU needed)."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_product_never_touches_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use oracle/."""
    for base, _dirs, files in os.walk(os.path.join(ROOT, 'gendr_amd')):
        for f in files:
            if f.endswith(('.py', '.h', '.hip', '.cpp', '.c')):
                text = open(os.path.join(base, f)).read()
                assert not re.search(r'^\s*(import|from)\s+oracle\b', text, flags=re.M), os.path.join(base, f)
                assert 'gendr_oracle' not in text, os.path.join(base, f)
    text = open(os.path.join(ROOT, 'include', 'gendr_hip.h')).read()
    assert 'oracle' not in text.lower()


def test_name_and_id_maps_match_reference_tables():
    from gendr_amd.functional import renderer as R
    assert R.DIST_FUNC_IDS['hard'] == R.DIST_FUNC_IDS['heaviside'] == 0
    assert R.DIST_FUNC_IDS['hyperbolic_secant'] == R.DIST_FUNC_IDS['gudermannian'] == 7
    assert sorted(set(R.DIST_FUNC_IDS.values())) == list(range(18))
    assert sorted(R.AGGR_ALPHA_FUNC_IDS.values()) == list(range(10))
    assert R.AGGR_RGB_FUNC_IDS == {'hard': 0, 'softmax': 1}
    assert R.TEXTURE_TYPE_IDS == {'surface': 0, 'vertex': 1}


def test_make_params_normalisation():
    from gendr_amd.functional.renderer import make_params
    p = make_params(128, [0.1, 0.2, 0.3], 6, np.float64(0.03), False, None, None, 300, 'yager', np.float32(2.0),
                    'hard', 1e-3, 1e-3, 1, 100, False, 'vertex')
    assert (p.image_size, p.dist_func, p.aggr_alpha_func, p.aggr_rgb_func, p.texture_type) == (128, 6, 6, 0, 1)
    assert p.dist_shape == 0.0 and p.dist_shift == 0.0 and abs(p.dist_scale - 0.03) < 1e-8
    assert abs(p.background[1] - 0.2) < 1e-7 and p.cull == 1 and p.texel_mode == 0
    with pytest.raises(AssertionError):
        make_params(64, [0, 0, 0], 1, -1.0, False, None, None, 1e4, 2, None, 1, 1e-3, 1e-3, 1, 100, True, 'surface')
    with pytest.raises(AssertionError):
        make_params(64, [0, 0, 0], 1, 1e-2, False, None, None, 0.5, 2, None, 1, 1e-3, 1e-3, 1, 100, True, 'surface')
    with pytest.raises(KeyError):
        make_params(64, [0, 0, 0], 'triangular', 1e-2, False, None, None, 1e4, 2, None, 1, 1e-3, 1e-3, 1, 100, True, 'surface')


def test_gendr_module_surface():
    import gendr_amd
    r = gendr_amd.GenDR()
    assert (r.image_size, r.dist_func, r.dist_scale, r.aggr_alpha_func, r.aggr_rgb_func) == (256, 'uniform', 1e-2, 'probabilistic', 'softmax')
    assert r.double_side is False and r.texture_type == 'surface' and r.anti_aliasing is False and r.dist_eps == 1e4
    with pytest.raises(ValueError):
        gendr_amd.GenDR(aggr_rgb_func='median')
    with pytest.raises(ValueError):
        gendr_amd.GenDR(texture_type='atlas')
    r.dist_scale = 0.5                       # options are plain attributes, read at call time
    assert r._options()['dist_scale'] == 0.5
    assert gendr_amd.GenDR(dist_scale=None).dist_scale is None   # opt_shape.py:139 constructs with None


def test_cpu_tensors_are_rejected():
    from gendr_amd.functional import render
    fv = torch.zeros(1, 2, 3, 3)
    tex = torch.zeros(1, 2, 1, 3)
    with pytest.raises(TypeError):
        render(fv, tex, image_size=16)


def test_render_signature_matches_reference_order():
    import inspect
    from gendr_amd.functional import render, GenDRFunction, soft_rasterize
    names = list(inspect.signature(render).parameters)
    assert names == ['face_vertices', 'textures', 'image_size', 'background_color', 'dist_func', 'dist_scale',
                     'dist_squared', 'dist_shape', 'dist_shift', 'dist_eps', 'aggr_alpha_func',
                     'aggr_alpha_t_conorm_p', 'aggr_rgb_func', 'aggr_rgb_eps', 'aggr_rgb_gamma', 'near', 'far',
                     'double_side', 'texture_type']
    assert list(inspect.signature(GenDRFunction.forward).parameters)[1:] == names
    assert inspect.signature(render).parameters['double_side'].default is True
    assert soft_rasterize is render


def test_geometry_helpers():
    from gendr_amd.functional.geometry import look_at, perspective, face_vertices, get_points_from_angles, vertex_normals
    v = torch.tensor([[[0., 0., 0.], [0.1, 0., 0.], [0., 0.1, 0.]]])
    eye = get_points_from_angles(2.0, 0.0, 0.0)
    assert np.allclose(eye, (0.0, 0.0, -2.0))
    cam = look_at(v, eye)
    assert torch.allclose(cam[0, 0], torch.tensor([0., 0., 2.]), atol=1e-6)
    assert torch.allclose(cam[0, 1], torch.tensor([0.1, 0., 2.]), atol=1e-6)
    p = perspective(cam, angle=45.)
    assert torch.allclose(p[0, 1], torch.tensor([0.05, 0., 2.]), atol=1e-6)
    f = torch.tensor([[[0, 1, 2]]])
    assert face_vertices(v, f).shape == (1, 1, 3, 3)
    n = vertex_normals(v, f)
    assert torch.allclose(n[0, 0].abs(), torch.tensor([0., 0., 1.]), atol=1e-6)
    e = get_points_from_angles(torch.tensor([2.0]), torch.tensor([30.0]), torch.tensor([-15.0]))
    assert e.shape == (1, 3)


def test_synthetic_scene_shape():
    from gendr_amd.synthetic import benchmark_scene, icosphere
    v, f = icosphere(3)
    assert v.shape == (642, 3) and f.shape == (1280, 3)
    fv, tex = benchmark_scene(3, subdivisions=1)
    assert fv.shape == (3, 80, 3, 3) and tex.shape == (3, 80, 1, 3)
    assert fv[..., :2].abs().max() < 1.0 and fv[..., 2].min() > 1.0


def test_vec3_broadcast_rules():
    from gendr_amd.functional.geometry import _as_vec3, look_at
    assert _as_vec3([0, 0, 1], 'cpu', 4).shape == (4, 3)
    assert _as_vec3(torch.zeros(1, 3), 'cpu', 4).shape == (4, 3)           # [1,3] broadcasts (reference behaviour)
    assert _as_vec3(torch.zeros(4, 3), 'cpu', 4).shape == (4, 3)
    assert _as_vec3(torch.zeros(2, 3), 'cpu', 4).shape == (2, 3)           # left to the caller's shape check
    v = torch.rand(3, 5, 3)
    a = look_at(v, torch.tensor([[0.0, 0.0, -2.7]]))
    b = look_at(v, [0.0, 0.0, -2.7])
    assert torch.equal(a, b)


def test_cull_radius_is_an_upper_bound_for_every_distribution(native_lib):
    """gendr_cull_radius bisects D(-x) against half the skip threshold; the bisection finds A crossing, culling
    needs that NO outside pixel beyond the radius passes the reference's skip test (kernel.cu:784: D <= 1e-6), i.e.
    D(-x) <= 1e-6 for every x >= r.  Scan that directly for every dist_func, scale, shape and shift -- it does not
    rely on D(-x) being monotone (monotonicity only held empirically for the truncated gamma series)."""
    import ctypes
    import numpy as np
    from gendr_amd.functional.renderer import make_params
    L = native_lib
    for tau in (1e-4, 1e-2, 3e-2, 1.0):
        for fid in range(1, 18):
            for squared in (False, True):
                for shape, shift in ((0.0, 0.0), (0.5, 0.8), (1.5, 0.0), (2.0, 0.0), (2.0, 1.0), (3.5, 0.25)):
                    p = make_params(64, [0, 0, 0], fid, tau, squared, shape, shift, 1e4, 2, 0.0, 1, 1e-3, 1e-3, 1, 100, True, 0)
                    r = L.gendr_cull_radius(ctypes.byref(p))
                    assert r == r and r >= 0
                    r_eps = float(np.sqrt(np.float32(1e4) * np.float32(tau)))       # kernel.cu:769 cuts there anyway
                    if not np.isfinite(r) or r >= r_eps:
                        continue
                    ds = np.unique(np.concatenate([r * (1 + np.logspace(-7, 1.5, 400)), np.linspace(r, min(r_eps, 4 * r + 1e-3), 400)]))
                    for d in ds[ds < r_eps]:
                        d = np.float32(d)
                        x = float(d * d) if squared else float(d)
                        v = L.gendr_sigmoid_forward(fid, -1.0, x, tau, shape, shift)
                        assert v != v or v <= 1e-6, (fid, tau, squared, shape, shift, float(d), r, v)


def test_workspace_of_radiusless_option_sets_has_no_pool(native_lib):
    """Distributions whose tail never falls below the contribution threshold inside the image (cauchy, reciprocal at the
    default dist_eps) list every face in every tile: the coverage pool would be its worst case (1.3 GiB at C2, 20 GiB at
    C4 in round 2) and add nothing -- such option sets get no pool (and no tile masks); and a caller's pool_entries_max
    caps the pool of any option set."""
    import ctypes
    from gendr_amd.functional import renderer as R
    bg = [0., 0., 0.]
    hints = 0

    def ws(B, nf, isz, **kw):
        o = dict(dist_func='uniform', dist_scale=1e-2, dist_squared=False, dist_shape=None, dist_shift=None, dist_eps=1e4,
                 aggr_alpha_func='probabilistic', aggr_alpha_t_conorm_p=None, aggr_rgb_func='softmax', aggr_rgb_eps=1e-3,
                 aggr_rgb_gamma=1e-3, near=1, far=100, double_side=False, texture_type='surface')
        limit = kw.pop('pool_entries_max', 0)
        nonlocal hints
        saved = hints
        if kw.pop('pair_hints_off', False):
            hints = -1
        o.update(kw)
        p = R.make_params(isz, bg, *[o[k] for k in ('dist_func', 'dist_scale', 'dist_squared', 'dist_shape', 'dist_shift', 'dist_eps',
                                                   'aggr_alpha_func', 'aggr_alpha_t_conorm_p', 'aggr_rgb_func', 'aggr_rgb_eps',
                                                   'aggr_rgb_gamma', 'near', 'far', 'double_side', 'texture_type')])
        p.pool_entries_max = limit
        p.pair_hints = hints
        hints = saved
        return int(native_lib.gendr_workspace_bytes(B, nf, 1, ctypes.byref(p)))

    hints = -1
    base = ws(64, 1280, 256)
    assert base < 200e6                                         # C2: records + masks + pool
    hints = 0                                                   # automatic: C2's 1.3-pixel cull radius gets pair hints (ABI 6),
    with_hints = ws(64, 1280, 256)                              # one 16-byte slot per pool entry
    assert base < with_hints < 300e6
    assert ws(256, 1280, 512, dist_func='logistic') == ws(256, 1280, 512, dist_func='logistic', pair_hints_off=True)   # C4 (37 pixels): none
    for dist in ('cauchy', 'reciprocal'):
        assert ws(64, 1280, 256, dist_func=dist) < 40e6         # records, queues, queue records only
        assert ws(256, 1280, 512, dist_func=dist) < 200e6       # C4 size: was 20 GiB
    capped = ws(64, 1280, 256, pool_entries_max=1000)
    assert capped < base and base - capped > 50e6


def test_reference_code_object_manifest():
    """oracle/_ref (the reference's device code, oracle/build_ref.py): when it is built, the manifest names every kernel
    of OBJECTS in float and double, the pin build is the one without contraction, and -- where the reference tree is
    present -- it records the hashes of the files it was compiled from.  Nothing of the translated sources stays."""
    import hashlib
    import json
    from oracle import build_ref
    if not build_ref.available():
        pytest.skip('oracle/_ref not built')
    man = json.load(open(build_ref.manifest_path()))
    assert set(man['objects']) == set(build_ref.OBJECTS)
    for name, (src, _, kernels) in build_ref.OBJECTS.items():
        o = man['objects'][name]
        assert os.path.exists(os.path.join(build_ref.REF_DIR, o['file'])) and o['source'] == src
        for k in kernels:
            assert k + '<float>' in o['kernels'] and k + '<double>' in o['kernels']
    assert '-ffp-contract=off' in man['objects']['render']['flags']
    assert '-ffp-contract=off' not in man['objects']['render_fma']['flags']
    if os.path.isdir(build_ref.REF_CUDA_DIR):
        for f, sha in man['reference_sha256'].items():
            assert sha == hashlib.sha256(open(os.path.join(build_ref.REF_CUDA_DIR, f), 'rb').read()).hexdigest()
    assert sorted(os.listdir(build_ref.REF_DIR)) == sorted(['manifest.json'] + [o['file'] for o in man['objects'].values()])


def test_option_cache_is_keyed_on_values_not_objects():
    """tensors hash by identity; an in-place update of a tensor-valued dist_scale must not hit the old entry."""
    import numpy as np
    import torch
    from gendr_amd.functional import renderer as R
    opts = lambda s, sq=False, df='uniform': (64, [0, 0, 0], df, s, sq, None, None, 1e4, 'probabilistic', None, 'softmax', 1e-3, 1e-3, 1, 100, True, 'surface')
    s = torch.nn.Parameter(torch.tensor(1e-2))
    a = R._params_bytes(*opts(s))[0]
    with torch.no_grad():
        s.mul_(10)
    b = R._params_bytes(*opts(s))[0]
    assert a != b
    assert b == R._params_bytes(*opts(float(s)))[0]
    assert R._params_bytes(*opts(np.float64(0.25)))[0] == R._params_bytes(*opts(0.25))[0]
    assert R._params_bytes(*opts(0.25, sq=1))[0] == R._params_bytes(*opts(0.25, sq=True))[0]
    R._params_bytes(*opts(0.25, df=1))
    with pytest.raises(ValueError):                     # a cached id 1 must not let the bool True through (True == 1 as a dict key)
        R._params_bytes(*opts(0.25, df=True))
