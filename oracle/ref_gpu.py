"""Launches the REFERENCE's own render kernels (oracle/_ref/*.co, built by oracle/build_ref.py from
/root/reference/gendr/cuda/generalized_renderer_cuda_kernel.cu) on the GPU.  TEST INFRASTRUCTURE ONLY: used by the
`-m gpu` pin tests to hold the CPU restatement -- and the HIP product -- against outputs of the reference's kernels.

The host side the code object lacks is restated here, and it is only buffer set-up and launch shapes:
  * buffers as functional/renderer.py:133-151 / :188-194 creates them (faces_info zeros [B,nf,27], aggrs_info zeros
    [B,2,is,is], soft_colors ones times the background, gradients zero);
  * launches as kernel.cu:1099-1150 (face preprocessing over B*nf threads, forward over B*is*is threads) and
    :1186-1222 (backward over B*is*is threads), 256 threads per block (NUM_THREADS, kernel.cu:18);
  * texture_size = textures.size(2), texture_res = int(sqrt(texture_size)) (kernel.cu:1097-1098).
Arguments go through hipModuleLaunchKernel's kernelParams in the order of the kernels' parameter lists (kernel.cu:620,
:680, :866).  Nothing here reads /root/reference.
"""
import ctypes
import json
import math
import os

import numpy as np
import torch

from . import build_ref

NUM_THREADS = 256          # kernel.cu:18
_hip = None
_modules = {}


def available(name=None):
    return build_ref.available(name)


def _rt():
    global _hip
    if _hip is None:
        _hip = ctypes.CDLL('libamdhip64.so')
        _hip.hipModuleLoad.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_char_p]
        _hip.hipModuleGetFunction.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_char_p]
        _hip.hipModuleLaunchKernel.argtypes = [ctypes.c_void_p] + [ctypes.c_uint] * 6 + [ctypes.c_uint, ctypes.c_void_p,
                                                                                         ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p]
    return _hip


def _check(code, what):
    if code != 0:
        raise RuntimeError('%s failed with HIP error %d' % (what, code))


def _kernels(variant):
    """variant: a code object of oracle/build_ref.py OBJECTS -- 'render' (no contraction: the pin build), 'render_fma',
    'voxelization', 'load_textures', 'create_texture_image'."""
    if variant not in _modules:
        with open(build_ref.manifest_path()) as f:
            man = json.load(f)['objects'][variant]
        torch.cuda.init()
        hip = _rt()
        mod = ctypes.c_void_p()
        _check(hip.hipModuleLoad(ctypes.byref(mod), os.path.join(build_ref.REF_DIR, man['file']).encode()), 'hipModuleLoad')
        fns = {}
        for key, sym in man['kernels'].items():
            fn = ctypes.c_void_p()
            _check(hip.hipModuleGetFunction(ctypes.byref(fn), mod, sym.encode()), 'hipModuleGetFunction(%s)' % key)
            fns[key] = fn
        _modules[variant] = (mod, fns)
    return _modules[variant][1]


def _launch(fn, n_threads, args, threads=NUM_THREADS):
    hip = _rt()
    blocks = (n_threads - 1) // threads + 1
    arr = (ctypes.c_void_p * len(args))(*[ctypes.cast(ctypes.pointer(a), ctypes.c_void_p) for a in args])
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _check(hip.hipModuleLaunchKernel(fn, blocks, 1, 1, threads, 1, 1, 0, stream, arr, None), 'hipModuleLaunchKernel')


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _option_args(p, texture_size):
    """The scalar tail shared by the forward and the backward kernel (kernel.cu:686-705 / :875-894); `p` is the
    normalised option set (gendr_amd.functional.renderer.make_params: names -> ids, None -> 0)."""
    return [ctypes.c_int(texture_size), ctypes.c_int(int(math.sqrt(texture_size))),
            ctypes.c_int(p.dist_func), ctypes.c_float(p.dist_scale), ctypes.c_bool(bool(p.dist_squared)),
            ctypes.c_float(p.dist_shape), ctypes.c_float(p.dist_shift), ctypes.c_float(p.dist_eps),
            ctypes.c_int(p.aggr_alpha_func), ctypes.c_float(p.aggr_alpha_t_conorm_p),
            ctypes.c_int(p.aggr_rgb_func), ctypes.c_float(p.aggr_rgb_eps), ctypes.c_float(p.aggr_rgb_gamma),
            ctypes.c_float(p.near_), ctypes.c_float(p.far_), ctypes.c_bool(bool(p.double_side)), ctypes.c_int(p.texture_type)]


def render(fv, tex, image_size, p, grad=None, dtype=np.float32, variant='render', device='cuda:0', pad_textures=True,
           background=None):
    """fv [B,nf,3,3], tex [B,nf,T,3] numpy; p: normalised options.  -> dict of numpy arrays (rgba, aggrs_info, faces_info
    and, with `grad` [B,4,is,is], grad_faces / grad_textures), computed by the reference's kernels in `dtype`.

    pad_textures: the reference's texel index can run past a face's texels (kernel.cu:153-176 with a clipped weight of
    exactly 1: index T instead of T - 1), i.e. into the next face's -- and for the last face of the tensor past the end of
    the tensor, which is undefined behaviour.  The overflowing index is exactly T -- w = (0, 1, 0): wy = R, wx = 0, index
    wy R + wx (:179-183); the mirrored branch cannot overflow -- so the read lands on the first texel behind the tensor.  The
    textures therefore live at the front of a larger buffer whose tail repeats what the restatement and the product DEFINE for
    that read: the last face's own texel at the clamped index, (R - 1) R = T - R (the only texel for T = 1; round 5: the tail
    used to repeat the tensor's last texel, T - 1, which is the same thing only for R = 1 -- found by the reference-arbitrated
    fuzz test on T = 4 / 9 draws).
    background: three floats (default: p.background, i.e. rounded to float as functional/renderer.py:147-149 does)."""
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    scalar = 'float' if dtype == np.float32 else 'double'
    k = _kernels(variant)
    B, nf = fv.shape[:2]
    T = tex.shape[2]
    faces = torch.from_numpy(np.ascontiguousarray(fv, dtype)).reshape(B, nf, 9).to(device)
    tex_t = torch.from_numpy(np.ascontiguousarray(tex, dtype)).to(device)
    if pad_textures:
        R = int(np.sqrt(T)) if int(p.texture_type) == 0 else 1
        own = tex_t.reshape(-1, 3)[tex_t.numel() // 3 - (R if int(p.texture_type) == 0 else 1)]       # last face, texel T - R
        store = own.repeat(tex_t.numel() // 3 + 1366)[:tex_t.numel() + 4098].contiguous()
        store[:tex_t.numel()] = tex_t.reshape(-1)
        textures = store[:tex_t.numel()].view(B, nf, T, 3)
    else:
        textures = tex_t
    faces_info = torch.zeros(B, nf, 27, dtype=tdt, device=device)
    aggrs_info = torch.zeros(B, 2, image_size, image_size, dtype=tdt, device=device)
    soft_colors = torch.ones(B, 4, image_size, image_size, dtype=tdt, device=device)
    bg = [float(p.background[c]) for c in range(3)] if background is None else [float(v) for v in background]
    for c in range(3):
        soft_colors[:, c] *= bg[c]
    ib, inf_, iis = ctypes.c_int(B), ctypes.c_int(nf), ctypes.c_int(image_size)
    tail = _option_args(p, T)
    _launch(k['forward_render_inv_cuda_kernel<%s>' % scalar], B * nf, [_ptr(faces), _ptr(faces_info), ib, inf_, iis])
    _launch(k['forward_render_cuda_kernel<%s>' % scalar], B * image_size * image_size,
            [_ptr(faces), _ptr(textures), _ptr(faces_info), _ptr(aggrs_info), _ptr(soft_colors), ib, inf_, iis] + tail)
    out = dict(rgba=soft_colors.cpu().numpy(), aggrs_info=aggrs_info.cpu().numpy(), faces_info=faces_info.cpu().numpy())
    if grad is not None:
        g = torch.from_numpy(np.ascontiguousarray(grad, dtype)).to(device)
        grad_faces = torch.zeros_like(faces)
        if pad_textures:
            gstore = torch.zeros(tex_t.numel() + 4096, dtype=tdt, device=device)
            grad_textures = gstore[:tex_t.numel()].view(B, nf, T, 3)
        else:
            grad_textures = torch.zeros_like(textures)
        _launch(k['backward_render_cuda_kernel<%s>' % scalar], B * image_size * image_size,
                [_ptr(faces), _ptr(textures), _ptr(soft_colors), _ptr(faces_info), _ptr(aggrs_info), _ptr(grad_faces),
                 _ptr(grad_textures), _ptr(g), ib, inf_, iis] + tail)
        out['grad_faces'] = grad_faces.cpu().numpy().reshape(B, nf, 3, 3)
        out['grad_textures'] = grad_textures.cpu().numpy()
    torch.cuda.synchronize()
    return out


def time_step(fv, tex, image_size, p, steps=10, warmup=2, variant='render_fma', device='cuda:0'):
    """ms per forward + backward of the reference's kernels (float) on the given inputs, HIP events around `steps`
    repetitions: the reference's own design timed on the same GPU.  The buffer set-up (zeros / ones) is outside."""
    k = _kernels(variant)
    B, nf = fv.shape[:2]
    T = tex.shape[2]
    faces = torch.from_numpy(np.ascontiguousarray(fv, np.float32)).reshape(B, nf, 9).to(device)
    store = torch.zeros(tex.size + 4096, dtype=torch.float32, device=device)
    store[:tex.size] = torch.from_numpy(np.ascontiguousarray(tex, np.float32)).reshape(-1).to(device)
    textures = store[:tex.size].view(B, nf, T, 3)
    g = torch.randn(B, 4, image_size, image_size, device=device)
    ib, inf_, iis = ctypes.c_int(B), ctypes.c_int(nf), ctypes.c_int(image_size)
    tail = _option_args(p, T)
    gstore = torch.zeros(tex.size + 4096, dtype=torch.float32, device=device)

    def step():
        faces_info = torch.zeros(B, nf, 27, device=device)
        aggrs_info = torch.zeros(B, 2, image_size, image_size, device=device)
        soft_colors = torch.ones(B, 4, image_size, image_size, device=device)
        grad_faces = torch.zeros_like(faces)
        gstore.zero_()
        _launch(k['forward_render_inv_cuda_kernel<float>'], B * nf, [_ptr(faces), _ptr(faces_info), ib, inf_, iis])
        _launch(k['forward_render_cuda_kernel<float>'], B * image_size * image_size,
                [_ptr(faces), _ptr(textures), _ptr(faces_info), _ptr(aggrs_info), _ptr(soft_colors), ib, inf_, iis] + tail)
        _launch(k['backward_render_cuda_kernel<float>'], B * image_size * image_size,
                [_ptr(faces), _ptr(textures), _ptr(soft_colors), _ptr(faces_info), _ptr(aggrs_info), _ptr(grad_faces),
                 _ptr(gstore), _ptr(g), ib, inf_, iis] + tail)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


# ------------------------------------------------------------------------------------------------------------
# voxelization (voxelization_cuda_kernel.cu) and the texture-atlas kernels: the same arrangement
# ------------------------------------------------------------------------------------------------------------
def voxel_sub1(faces, size, dim, device='cuda:0'):
    """voxelize_sub1 of functional/voxelization.py:11-19 around voxelize_sub1_kernel (launch: voxelization_cuda_kernel.cu
    :204-217, 512 threads over B * vs * vs rays).  faces [B,nf,3,3] float32 in voxel units -> int32 [B,vs,vs,vs]."""
    k = _kernels('voxelization')
    f = torch.from_numpy(np.ascontiguousarray(faces, np.float32)).to(device)
    B, nf = f.shape[:2]
    if dim == 0:
        f = f[:, :, :, [2, 1, 0]].contiguous()
    elif dim == 1:
        f = f[:, :, :, [0, 2, 1]].contiguous()
    vox = torch.zeros(B, size, size, size, dtype=torch.int32, device=device)
    _launch(k['voxelize_sub1_kernel<float>'], B * size * size, [_ptr(f), _ptr(vox), ctypes.c_int(B), ctypes.c_int(nf), ctypes.c_int(size)], 512)
    torch.cuda.synchronize()
    return vox.transpose(dim + 1, -1).contiguous().cpu().numpy()


def voxel_sub2(faces, size, device='cuda:0'):
    k = _kernels('voxelization')
    f = torch.from_numpy(np.ascontiguousarray(faces, np.float32)).to(device)
    B, nf = f.shape[:2]
    vox = torch.zeros(B, size, size, size, dtype=torch.int32, device=device)
    _launch(k['voxelize_sub2_kernel<float>'], B * nf, [_ptr(f), _ptr(vox), ctypes.c_int(B), ctypes.c_int(nf), ctypes.c_int(size)], 512)
    torch.cuda.synchronize()
    return vox.cpu().numpy()


def voxel_fill(voxels, device='cuda:0', max_sweeps=100000):
    """voxelize_sub3 of functional/voxelization.py:29-44: sub3 once, sub4 until the visible count stops changing.
    voxels int32 [B,vs,vs,vs] (0/1) -> (1 - visible, number of sub4 sweeps)."""
    k = _kernels('voxelization')
    vox = torch.from_numpy(np.ascontiguousarray(voxels, np.int32)).to(device)
    B, vs = vox.shape[0], vox.shape[1]
    visible = torch.zeros_like(vox)
    n = B * vs * vs * vs
    args = [_ptr(vox), _ptr(visible), ctypes.c_int(B), ctypes.c_int(vs)]
    _launch(k['voxelize_sub3_kernel<float>'], n, args, 512)
    total = int(visible.sum())
    sweeps = 0
    while sweeps < max_sweeps:
        _launch(k['voxelize_sub4_kernel<float>'], n, args, 512)
        sweeps += 1
        now = int(visible.sum())
        if now == total:
            break
        total = now
    return (1 - visible).cpu().numpy(), sweeps


def voxelization(faces, size, normalize=False, device='cuda:0'):
    """functional/voxelization.py:46-62 with the reference's kernels."""
    f = np.array(faces, np.float32, copy=True)
    if not normalize:
        f *= np.float32(size)
    v = voxel_sub1(f, size, 0, device) + voxel_sub1(f, size, 1, device) + voxel_sub1(f, size, 2, device) + voxel_sub2(f, size, device)
    return voxel_fill((v > 0).astype(np.int32), device)[0]


def load_textures(image, face_uv, is_update, textures, device='cuda:0'):
    """load_textures_cuda (load_textures_cuda_kernel.cu:74-107: 1024 threads over numel / 3 texels; texture_res =
    sqrt(textures.size(1)) converted to size_t).  image [H,W,3], face_uv [nf,3,2], is_update [nf] int32, textures
    [nf,R*R,3] -> the updated textures."""
    k = _kernels('load_textures')
    d = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(device, dt).contiguous()
    img, uv, upd, tex = d(image, torch.float32), d(face_uv, torch.float32), d(is_update, torch.int32), d(textures, torch.float32)
    numel = tex.numel()
    res = int(math.sqrt(tex.shape[1]))
    sz = ctypes.c_size_t
    _launch(k['load_textures_cuda_kernel<float>'], numel // 3, [_ptr(img), _ptr(uv), _ptr(upd), _ptr(tex), sz(numel), sz(res),
                                                                sz(img.shape[0]), sz(img.shape[1])], 1024)
    torch.cuda.synchronize()
    return tex.cpu().numpy()


def create_texture_image_kernel(face_uv, textures, image, eps=1e-5, device='cuda:0'):
    """create_texture_image_cuda (create_texture_image_cuda_kernel.cu:79-111: tile_width = int(sqrt(nf - 1)) + 1,
    texture_res_out = image.size(1) / tile_width, 1024 threads over numel / 3 pixels).  face_uv [nf,3,2] in pixels,
    textures [nf,R*R,3], image [rows,cols,3] -> the painted image."""
    k = _kernels('create_texture_image')
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(device).contiguous()
    uv, tex, img = d(face_uv), d(textures), d(image)
    nf = tex.shape[0]
    res_in = int(math.sqrt(tex.shape[1]))
    tile_width = int(math.sqrt(nf - 1)) + 1
    res_out = img.shape[1] // tile_width
    numel = img.numel()
    sz = ctypes.c_size_t
    _launch(k['create_texture_image_cuda_kernel<float>'], numel // 3,
            [_ptr(uv), _ptr(tex), _ptr(img), sz(numel), sz(nf), sz(res_in), sz(res_out), sz(tile_width), ctypes.c_float(eps)], 1024)
    torch.cuda.synchronize()
    return img.cpu().numpy()
