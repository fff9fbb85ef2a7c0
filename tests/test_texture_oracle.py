"""The texture-atlas oracle (oracle/texture_ref.py) against closed-form cases.  (The reference has no vectors for
this path: the pin to its own kernels is tests/test_gpu_reference_pin_aux.py.)"""
import numpy as np

from oracle import texture_ref as T


def _uv(nf, seed=0):
    return np.random.default_rng(seed).uniform(0.05, 0.95, (nf, 3, 2)).astype(np.float32)


def test_constant_and_linear_images():
    nf, R, H, W = 7, 4, 33, 47
    uv = _uv(nf)
    tex0 = np.full((nf, R * R, 3), 0.25, np.float32)
    const = np.broadcast_to(np.float32([0.2, 0.5, 0.9]), (H, W, 3))
    out = T.load_textures(const, uv, np.ones(nf, np.int32), tex0)
    np.testing.assert_allclose(out, np.broadcast_to(np.float32([0.2, 0.5, 0.9]), out.shape), atol=2e-7)
    # image = (u, v, 0): bilinear sampling of a linear image is exact, the texel value is the barycentric mix of uv
    ramp = np.zeros((H, W, 3), np.float32)
    ramp[..., 0] = np.arange(W, dtype=np.float32)[None, :] / (W - 1)
    ramp[..., 1] = np.arange(H, dtype=np.float32)[:, None] / (H - 1)
    out = T.load_textures(ramp, uv, np.ones(nf, np.int32), tex0)
    i = np.arange(R * R)
    wy, wx = i // R, i % R
    lower = wx + wy < R
    w0 = np.where(lower, (wx + 1 / 3) / R, (R - 1 - wx + 2 / 3) / R)
    w1 = np.where(lower, (wy + 1 / 3) / R, (R - 1 - wy + 2 / 3) / R)
    w = np.stack([w0, w1, 1 - w0 - w1], 1)                               # [RR,3]
    np.testing.assert_allclose(out[..., :2], np.einsum('tk,fkc->ftc', w, uv.astype(np.float64)), atol=3e-6)
    np.testing.assert_array_equal(out[..., 2], 0)


def test_is_update_masks_faces_and_last_pixel_is_safe():
    nf, R = 5, 2
    uv = _uv(nf, 1)
    uv[0] = 1.0                                                          # samples exactly the last row / column
    img = np.random.default_rng(2).random((9, 11, 3)).astype(np.float32)
    tex0 = np.full((nf, R * R, 3), 0.5, np.float32)
    upd = np.array([1, 0, 1, 0, 1], np.int32)
    out = T.load_textures(img, uv, upd, tex0)
    np.testing.assert_array_equal(out[[1, 3]], tex0[[1, 3]])
    np.testing.assert_allclose(out[0], np.broadcast_to(img[-1, -1], (R * R, 3)), atol=1e-6)
    assert np.isfinite(out).all()


def test_atlas_layout_and_constant_tiles():
    nf, R, res = 11, 3, 8
    tw, th, uv = T.atlas_layout(nf, res)
    assert (tw, th) == (4, 3)
    colours = np.random.default_rng(3).random((nf, 1, 3)).astype(np.float32)
    tex = np.broadcast_to(colours, (nf, R * R, 3)).copy()
    img, uv01 = T.create_texture_image(tex, res)
    assert img.shape == (th * res, tw * res, 3) and uv01.shape == (nf, 3, 2)
    up = img[::-1]                                                       # undo the vertical flip
    for fn in range(nf):
        r, c = fn // tw, fn % tw
        np.testing.assert_array_equal(up[r * res:(r + 1) * res, c * res:(c + 1) * res], np.broadcast_to(colours[fn, 0], (res, res, 3)))
    np.testing.assert_array_equal(up[2 * res:, 3 * res:], 1.0)            # the tile after the last face keeps the fill value
    assert uv01.min() >= 0 and uv01.max() <= 1
    np.testing.assert_allclose(uv01[:, :, 0] * (tw * res - 1), uv[:, :, 0], atol=1e-4)


def test_atlas_round_trip_recovers_per_face_colours():
    nf, R, res = 10, 4, 16
    colours = np.random.default_rng(4).random((nf, 1, 3)).astype(np.float32)
    tex = np.broadcast_to(colours, (nf, R * R, 3)).copy()
    img, uv01 = T.create_texture_image(tex, res)
    back = T.load_textures(img[::-1], uv01, np.ones(nf, np.int32), np.zeros_like(tex))
    np.testing.assert_allclose(back, tex, atol=1e-6)


def test_atlas_texel_pattern_inside_a_tile():
    # one face, R=2: the four texels appear as the lower-left / mirrored upper-right halves of the clipped barycentrics
    tex = np.float32([[[1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 0]]])
    img, _ = T.create_texture_image(tex, 16)
    up = img[::-1]
    seen = {tuple(px) for px in up.reshape(-1, 3)}
    assert seen <= {(1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0)} and len(seen) >= 3
