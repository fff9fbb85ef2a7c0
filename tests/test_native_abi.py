"""The C-ABI library loads without a GPU and exports exactly what include/gendr_hip.h declares;
option validation follows the reference's asserts and device-side parameter checks."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, 'include', 'gendr_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(gendr_[a-z_0-9]+)\s*\(', text)))


def test_every_declared_symbol_is_exported(native_lib):
    names = _declared_functions()
    assert len(names) >= 14
    for n in names:
        assert hasattr(native_lib, n), n


def test_python_binding_lists_the_same_symbols():
    from gendr_amd import _native
    assert sorted(_native.EXPORTS) == _declared_functions()


def test_struct_layout_and_version(native_lib):
    from gendr_amd import _native
    assert native_lib.gendr_abi_version() == _native.ABI_VERSION
    assert native_lib.gendr_params_size() == ctypes.sizeof(_native.GendrParams)


def _params(**kw):
    from gendr_amd.functional.renderer import make_params
    base = dict(image_size=64, background_color=[0, 0, 0], dist_func='uniform', dist_scale=1e-2, dist_squared=False,
                dist_shape=None, dist_shift=None, dist_eps=1e4, aggr_alpha_func='probabilistic',
                aggr_alpha_t_conorm_p=None, aggr_rgb_func='softmax', aggr_rgb_eps=1e-3, aggr_rgb_gamma=1e-3,
                near=1, far=100, double_side=True, texture_type='surface')
    base.update(kw)
    return make_params(**base)


@pytest.mark.parametrize("kw,code", [
    (dict(), 0),
    (dict(dist_func=18), -3), (dict(dist_func=-1), -3),
    (dict(aggr_alpha_func=10), -4),
    (dict(aggr_rgb_func=2), -5),
    (dict(dist_func='gamma', dist_shape=-0.5), -7),
    (dict(aggr_alpha_func='hamacher', aggr_alpha_t_conorm_p=-1.0), -8),
    (dict(aggr_alpha_func='frank', aggr_alpha_t_conorm_p=1.0), -8),
    (dict(aggr_alpha_func='frank'), -8),                       # None -> 0.0, invalid for frank
    (dict(aggr_alpha_func='yager', aggr_alpha_t_conorm_p=0.0), -8),
    (dict(aggr_alpha_func='aczel_alsina', aggr_alpha_t_conorm_p=-1.0), -8),
    (dict(aggr_alpha_func='dombi'), -8),
    (dict(aggr_alpha_func='schweizer_sklar', aggr_alpha_t_conorm_p=0.5), -8),
    (dict(aggr_alpha_func='schweizer_sklar', aggr_alpha_t_conorm_p=-0.5), 0),
    (dict(aggr_alpha_func='hamacher'), 0),                     # p = 0 is valid for hamacher (opt_shape.py:106)
])
def test_validate_codes(native_lib, kw, code):
    p = _params(**kw)
    assert native_lib.gendr_validate(ctypes.byref(p), 2, 10, 1) == code
    assert isinstance(native_lib.gendr_error_string(code), bytes)


def test_validate_shapes(native_lib):
    p = _params()
    assert native_lib.gendr_validate(ctypes.byref(p), 2, 10, 0) == -2          # T < 1
    assert native_lib.gendr_validate(ctypes.byref(p), -1, 10, 1) == -2
    pv = _params(texture_type='vertex')
    assert native_lib.gendr_validate(ctypes.byref(pv), 2, 10, 3) == 0
    assert native_lib.gendr_validate(ctypes.byref(pv), 2, 10, 4) == -6         # vertex colours need T == 3
    assert native_lib.gendr_validate(None, 1, 1, 1) == -1


def test_workspace_bytes(native_lib):
    p = _params(image_size=256)
    # bin record 16 B (the cull box; 64 B until round 6) + record 224 B per face, masks 8 B per (8x8 tile, 64-face chunk), tile queue 4 B and queue record 16 B per tile, the entry pool (16 B per slot: 32 per tile + 64 per face, at most tiles * faces),
    # the heavy-first copy of the queue records (16 B per tile, for 16 .. 2^19 tiles), control block (24 counters, 4 KiB apart); every part 256-byte aligned
    n = native_lib.gendr_workspace_bytes(2, 1280, 1, ctypes.byref(p))
    control = 24 * 1024 * 4
    tiles = 2 * 32 * 32
    pool = 2 * (32 * tiles + 64 * 2 * 1280) * 16          # radius of 1.3 pixels: 64 per face; fewer than 8 batch items: doubled
    # + the pair hints (ABI 6; automatic: on for this 1.3-pixel cull radius): one 16-byte slot per pool entry
    # (the per-image lists of faces with a loose cull box exist from 1024^2 only -- loose_faces_kernel; below that the coverage kernel
    # resolves them without any buffer of its own: round 4)
    assert n == 2 * 1280 * 16 + 2 * 1280 * 224 + tiles * 20 * 8 + tiles * 4 + tiles * 16 + pool + pool + tiles * 16 + control
    # 1024^2: a flag (4 B) and a pixel box (16 B) per face, a list of 16 ints per image
    big1 = native_lib.gendr_workspace_bytes(2, 1280, 1, ctypes.byref(_params(image_size=1024)))
    q1 = _params(image_size=1016)
    assert big1 - native_lib.gendr_workspace_bytes(2, 1280, 1, ctypes.byref(q1)) >= 2 * 1280 * 4 + 2 * 1280 * 16 + 256
    q = _params(image_size=256)
    q.pair_hints = -1
    assert native_lib.gendr_workspace_bytes(2, 1280, 1, ctypes.byref(q)) == n - pool
    # above 2^19 tiles the render kernels walk the queue records in the binning order: no copy
    big = native_lib.gendr_workspace_bytes(32, 1280, 1, ctypes.byref(_params(image_size=2048)))
    big_tiles = 32 * 256 * 256
    assert big < 32 * 1280 * (64 + 224 + 20) + big_tiles * (20 * 8 + 4 + 16) + 2 * 16 * (32 * big_tiles + 20000 * 32 * 1280) + control + 12 * 256 + 32 * 64
    # tiny problems: the pool never exceeds one slot per (tile, face)
    small = native_lib.gendr_workspace_bytes(1, 2, 1, ctypes.byref(_params(image_size=8)))
    assert small == 256 * 4 + 512 + 256 + 256 + control    # four sub-256-byte parts, 2 records (448 B), an 8-slot pool and its hint slots, the counters
    assert native_lib.gendr_workspace_bytes(2, 1280, 3, ctypes.byref(_params(image_size=256, texture_type='vertex'))) > n
    assert native_lib.gendr_workspace_bytes(2, 1280, 0, ctypes.byref(p)) == 0


def test_cull_radius_is_conservative(native_lib, oracle_mod):
    """Beyond the reported radius the oracle's own CDF is below the skip threshold (kernel.cu:784)."""
    import math
    for fid in range(18):
        for sq in (False, True):
            p = _params(dist_func=fid, dist_scale=2e-2, dist_squared=sq, dist_shape=2.0, dist_shift=0.3)
            r = native_lib.gendr_cull_radius(ctypes.byref(p))
            assert r >= 0
            if math.isinf(r) or r >= 8:
                continue
            for k in (1.0, 1.01, 1.5, 4.0):
                d = r * k + 1e-7
                x = d * d if sq else d
                assert oracle_mod.sigmoid_forward(fid, -1.0, x, 2e-2, 2.0, 0.3) <= 1e-6, (fid, sq, r, k)
    p = _params(dist_func='uniform')
    p.cull = 0
    assert math.isinf(native_lib.gendr_cull_radius(ctypes.byref(p)))


def test_null_pointers_are_rejected_not_dereferenced(native_lib):
    p = _params()
    assert native_lib.gendr_forward(None, None, None, None, None, 1, 1, 1, ctypes.byref(p), None) == -1
    assert native_lib.gendr_face_info(None, None, 1, 1, None) == -1


def test_cpp_autograd_node_builds_and_binds(native_lib):
    """gendr_amd/_gendr_torch.so (csrc/gendr_torch.cpp): loads without a GPU, takes the C-ABI entry points of the loaded library by
    address (it links none of them), refuses another ABI, and raises the reference-shaped TypeError for CPU tensors."""
    import ctypes
    import torch
    from gendr_amd import build, _native
    build.build_torch_ext()
    ext = _native.torch_ext()
    assert ext is not None and hasattr(ext, 'bind') and hasattr(ext, 'render')
    slot = _native.torch_slot()
    assert slot == _native.torch_slot()                       # bound once per variant
    with pytest.raises(RuntimeError):
        ext.bind(0, 0, 0, 0, 0, native_lib.gendr_params_size(), native_lib.gendr_abi_version() + 1)
    with pytest.raises(TypeError):
        ext.render(torch.zeros(1, 2, 3, 3), torch.zeros(1, 2, 1, 3), bytes(ctypes.sizeof(_native.GendrParams)), slot, True)
