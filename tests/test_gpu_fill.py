"""Background fill of the forward pass.  A 64x64-pixel super-tile no face reaches is one entry of the unlisted-tile
queue and is written with 256-byte row segments when the planes allow 16-byte stores (bin_faces_kernel /
render_forward_body); otherwise tile by tile.  Every route must leave the pixels the oracle leaves: small object in the
middle of images of 192 (3 x 3 super-tiles, all inside), 200 (last super-tile column cut by the border) and 130 pixels
(rows not a multiple of four: no super-tile entries), soft and hard colour aggregation, the alpha-only kernels, planes at
a 4-byte offset, and the background taken from the output buffer (the pybind-shaped entry point of the scripts)."""
import ctypes

import numpy as np
import pytest
import torch

import parity

pytestmark = pytest.mark.gpu


def _scene(B=2, nf=10, seed=3):
    rs = np.random.RandomState(seed)
    fv = np.zeros((B, nf, 3, 3), np.float32)
    c = rs.uniform(-0.12, 0.12, (B, nf, 1, 2))
    fv[..., :2] = c + 0.1 * rs.uniform(-1, 1, (B, nf, 3, 2))
    fv[..., 2] = rs.uniform(1.5, 5.0, (B, nf, 3))
    tex = rs.uniform(0, 1, (B, nf, 1, 3)).astype(np.float32)
    return fv, tex


OPTS = [
    ('soft', dict(background=(0.2, 0.5, 0.7))),
    ('hard_rgb', dict(aggr_rgb_func='hard', background=(0.3, 0.1, 0.9))),
    ('hard_all', dict(dist_func='hard', aggr_alpha_func='hard', aggr_rgb_func='hard')),
]


@pytest.mark.parametrize("isz", [192, 200, 130, 64])
@pytest.mark.parametrize("name,opts", OPTS, ids=[n for n, _ in OPTS])
def test_background_pixels_equal_the_oracle(native_lib, isz, name, opts):
    fv, tex = _scene()
    h = parity.run_hip(fv, tex, isz, opts)
    r = parity.run_oracle(fv, tex, isz, opts)
    far = np.ones((isz, isz), bool)
    m = int(0.3 * isz)
    far[m:isz - m, m:isz - m] = False                      # the object and its soft halo live in the middle
    for k in ('rgba', 'aggrs_info'):
        assert np.array_equal(h[k][..., far], r[k][..., far]), (k, name, isz)       # untouched pixels: bit for bit
        np.testing.assert_allclose(h[k], r[k], rtol=1e-5, atol=1e-6)


def _forward_into(fv, tex, isz, opts, offset_floats):
    from gendr_amd import _native
    from gendr_amd.functional import renderer as R
    L = _native.lib()
    o, extra = parity.split_options(opts)
    p = parity.hip_params(isz, o, extra)
    B, nf = fv.shape[:2]
    faces = torch.from_numpy(fv).reshape(B, nf, 9).cuda()
    textures = torch.from_numpy(tex).cuda()
    P = isz * isz
    big_rgba = torch.full((B * 4 * P + 8,), float('nan'), device='cuda')
    big_aux = torch.full((B * 2 * P + 8,), float('nan'), device='cuda')
    rgba = big_rgba[offset_floats:offset_floats + B * 4 * P].view(B, 4, isz, isz)
    aux = big_aux[offset_floats:offset_floats + B * 2 * P].view(B, 2, isz, isz)
    assert rgba.data_ptr() % 16 == (4 * offset_floats) % 16
    R.native_forward(faces, textures, p, rgba=rgba, aggrs_info=aux)
    torch.cuda.synchronize()
    return rgba.cpu().numpy(), aux.cpu().numpy(), big_rgba.cpu().numpy(), big_aux.cpu().numpy()


@pytest.mark.parametrize("name,opts", OPTS[:2], ids=[n for n, _ in OPTS[:2]])
def test_planes_at_a_four_byte_offset(native_lib, name, opts):
    fv, tex = _scene()
    a = _forward_into(fv, tex, 192, opts, 0)
    b = _forward_into(fv, tex, 192, opts, 1)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    for big, off in ((b[2], 1), (b[3], 1)):                # nothing written outside the planes
        assert np.isnan(big[:off]).all() and np.isnan(big[len(big) - (8 - off):]).all()
    assert not np.isnan(a[0]).any() and not np.isnan(a[1]).any()


def test_background_from_the_output_buffer(native_lib):
    """forward_render of the pybind-shaped module: soft_colors arrives pre-filled with a per-pixel background."""
    from gendr_amd.cuda import generalized_renderer as G
    fv, tex = _scene()
    B, nf = fv.shape[:2]
    isz = 192
    faces = torch.from_numpy(fv).reshape(B, nf, 9).cuda()
    textures = torch.from_numpy(tex).cuda()
    rs = np.random.RandomState(0)
    bg = rs.uniform(0, 1, (B, 3, isz, isz)).astype(np.float32)
    outs = {}
    for rgb_func in (0, 1):
        soft = torch.ones(B, 4, isz, isz, device='cuda')
        soft[:, :3] = torch.from_numpy(bg).cuda()
        faces_info = torch.zeros(B, nf, 27, device='cuda')
        aggrs = torch.zeros(B, 2, isz, isz, device='cuda')
        G.forward_render(faces, textures, faces_info, aggrs, soft, isz, 1, 1e-2, False, 0., 0., 1e4, 1, 0.,
                         rgb_func, 1e-3, 1e-3, 1., 100., True, 0)
        outs[rgb_func] = soft.cpu().numpy()
    m = int(0.3 * isz)
    far = np.ones((isz, isz), bool)
    far[m:isz - m, m:isz - m] = False
    # hard colour aggregation keeps the buffer's background where no face is; softmax returns (bg * s) / s
    assert np.array_equal(outs[0][:, :3][..., far], bg[..., far])
    np.testing.assert_allclose(outs[1][:, :3][..., far], bg[..., far], rtol=3e-7)     # (bg * s) / s: within an ulp of bg
    for v in outs.values():
        assert (v[:, 3][..., far] == 0).all()


def test_alpha_only_fill(native_lib):
    from gendr_amd.functional import render, render_silhouette
    fv, tex = _scene()
    for isz in (192, 200, 130):
        f = torch.from_numpy(fv).cuda()
        full = render(f, torch.from_numpy(tex).cuda(), image_size=isz)
        sil = render_silhouette(f, image_size=isz)
        assert torch.equal(full[:, 3], sil)
