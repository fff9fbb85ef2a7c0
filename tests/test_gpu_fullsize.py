"""BASELINE.json's full sizes (256x256, 1280 faces) through size-independent properties, plus one frame
checked element-wise against the oracle."""
import numpy as np
import pytest
import torch

import criteria
import parity

pytestmark = pytest.mark.gpu

C2 = dict(dist_func='uniform', dist_scale=1e-2, aggr_alpha_func='probabilistic', aggr_rgb_func='softmax', double_side=False)
C3 = dict(dist_func='gaussian', dist_scale=1e-4, dist_squared=True, aggr_alpha_func='einstein', double_side=False)


C4 = dict(dist_func='logistic', dist_scale=1e-2, aggr_alpha_func='probabilistic', aggr_rgb_func='softmax', double_side=False)
C5 = dict(dist_func='gamma', dist_shape=2.0, dist_scale=1e-2, aggr_alpha_func='yager', aggr_alpha_t_conorm_p=2.0,
          aggr_rgb_func='softmax', texture_type='vertex', double_side=False)


def _scene(B, texture='surface'):
    from gendr_amd.synthetic import benchmark_scene
    fv, tex = benchmark_scene(B, texture=texture)
    return fv.numpy(), tex.numpy()


@pytest.mark.parametrize("opts", [C2, C3], ids=['C2', 'C3'])
def test_one_full_frame_against_oracle(oracle_mod, native_lib, opts):
    fv, tex = _scene(3)
    fv, tex = fv[2:3], tex[2:3]
    res, h, r = parity.compare(fv, tex, 256, opts)
    if opts is C2:
        assert np.array_equal(h['rgba'][:, 3], r['rgba'][:, 3])
        assert res['rgba']['max_rel'] <= 1e-5
        assert res['grad_faces_cond']['max_rel'] <= 1e-5 and res['grad_textures_cond']['max_rel'] <= 1e-5
    grad = np.random.RandomState(1).randn(1, 4, 256, 256).astype(np.float32)
    bad, _, _ = criteria.check_case(fv, tex, 256, opts, h, grad, oracle_f32=r, key='C2' if opts is C2 else 'C3')
    assert not bad, bad


@pytest.mark.parametrize("name,opts,isz", [('C4', C4, 512), ('C5', C5, 768)])
def test_large_frames_against_oracle(oracle_mod, native_lib, name, opts, isz):
    """C4 at its full size (long face lists: a 37-pixel cull radius); C5's option set (its own kernel, vertex colours)
    at 768^2, large enough for backward to take the scalar phase-A walk (>= 400 image pixels per face)."""
    fv, tex = _scene(2, texture='vertex' if name == 'C5' else 'surface')
    fv, tex = fv[1:2], tex[1:2]
    res, h, r = parity.compare(fv, tex, isz, opts)
    grad = np.random.RandomState(1).randn(1, 4, isz, isz).astype(np.float32)
    bad, _, _ = criteria.check_case(fv, tex, isz, opts, h, grad, oracle_f32=r, n_jitter=6, key=name)
    assert not bad, bad


@pytest.mark.parametrize("opts", [C2, C3], ids=['C2', 'C3'])
def test_culling_exact_at_full_size(native_lib, opts):
    fv, tex = _scene(4)
    grad = np.random.RandomState(1).randn(4, 4, 256, 256).astype(np.float32)
    a = parity.run_hip(fv, tex, 256, opts, grad)
    b = parity.run_hip(fv, tex, 256, dict(opts, cull=0), grad)
    assert np.array_equal(a['rgba'], b['rgba']) and np.array_equal(a['aggrs_info'], b['aggrs_info'])
    for k in ('grad_faces', 'grad_textures'):
        assert np.abs(a[k] - b[k]).max() <= 2e-5 * np.abs(b[k]).max()


def test_batch_of_64_equals_items_rendered_alone(native_lib):
    """Batch items are independent -- except through the reference's texel-index overflow (kernel.cu:179-184),
    where the LAST face of item i reads the first texel of item i+1.  So: alpha always, everything with the
    clamped texel mode."""
    fv, tex = _scene(64)
    full = parity.run_hip(fv, tex, 256, C2)
    full_clamp = parity.run_hip(fv, tex, 256, dict(C2, texel_mode=1))
    for i in (0, 17, 63):
        one = parity.run_hip(fv[i:i + 1], tex[i:i + 1], 256, C2)
        assert np.array_equal(full['rgba'][i, 3], one['rgba'][0, 3])
        one = parity.run_hip(fv[i:i + 1], tex[i:i + 1], 256, dict(C2, texel_mode=1))
        assert np.array_equal(full_clamp['rgba'][i], one['rgba'][0])


def test_backward_is_linear_in_the_upstream_gradient(native_lib):
    fv, tex = _scene(2)
    rs = np.random.RandomState(0)
    g1 = rs.randn(2, 4, 256, 256).astype(np.float32)
    g2 = rs.randn(2, 4, 256, 256).astype(np.float32)
    a = parity.run_hip(fv, tex, 256, C2, g1)
    b = parity.run_hip(fv, tex, 256, C2, g2)
    c = parity.run_hip(fv, tex, 256, C2, g1 + g2)
    for k in ('grad_faces', 'grad_textures'):
        s = a[k] + b[k]
        assert np.abs(c[k] - s).max() <= 1e-4 * np.abs(s).max()


def test_alpha_is_invariant_to_face_order_for_max(native_lib):
    """max t-conorm is order independent exactly; the fold order only matters for rounding otherwise."""
    fv, tex = _scene(1)
    perm = np.random.RandomState(0).permutation(fv.shape[1])
    o = dict(C2, aggr_alpha_func='max')
    a = parity.run_hip(fv, tex, 256, o)
    b = parity.run_hip(fv[:, perm], tex[:, perm], 256, o)
    assert np.array_equal(a['rgba'][:, 3], b['rgba'][:, 3])
