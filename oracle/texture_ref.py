"""CPU oracle for the texture-atlas kernels.  TEST INFRASTRUCTURE ONLY -- nothing under
gendr_amd/ may import this.

numpy restatement (fp32 with the reference's double sub-expressions, no contraction) of
  load_textures_cuda_kernel         gendr/cuda/load_textures_cuda_kernel.cu:14-72
  create_texture_image_cuda_kernel  gendr/cuda/create_texture_image_cuda_kernel.cu:16-75
PARITY PINNED to outputs of the reference's own kernels (oracle/build_ref.py -> oracle/_ref, oracle/ref_gpu.py): bit for
bit on the GPU box (tests/test_gpu_reference_pin_aux.py; samples that make the reference read outside its image are left
to the clamped definition below).  On the CPU side: closed-form
cases in tests/test_texture_oracle.py (constant images, exact texel centres, atlas round trip).
Where the reference reads outside its buffers (weight-0 neighbour past the last row / column; texel index outside
the face's block) the index is clamped, as in the HIP kernels.
"""
import numpy as np

F = np.float32


def load_textures(image, face_uv, is_update, textures):
    """image [H,W,3], face_uv [nf,3,2], is_update [nf], textures [nf,R*R,3] -> updated copy of textures."""
    image, face_uv = np.asarray(image, F), np.asarray(face_uv, F)
    out = np.array(textures, dtype=F, copy=True)
    nf, RR = out.shape[:2]
    R = int(np.sqrt(RR))
    H, W = image.shape[:2]
    i = np.arange(nf * RR)
    w_y, w_x = (i % RR) // R, i % R
    lower = w_x + w_y < R
    w0 = np.where(lower, (w_x + 1. / 3.) / R, ((R - 1. - w_x) + 2. / 3.) / R).astype(F)      # double -> float
    w1 = np.where(lower, (w_y + 1. / 3.) / R, ((R - 1. - w_y) + 2. / 3.) / R).astype(F)
    w2 = (1. - w0.astype(np.float64) - w1.astype(np.float64)).astype(F)
    f = face_uv[i // RR]                                                                     # [n,3,2]
    pos_x = ((f[:, 0, 0] * w0 + f[:, 1, 0] * w1) + f[:, 2, 0] * w2) * F(W - 1)
    pos_y = ((f[:, 0, 1] * w0 + f[:, 1, 1] * w1) + f[:, 2, 1] * w2) * F(H - 1)
    xi, yi = pos_x.astype(np.int64), pos_y.astype(np.int64)                                  # (int) truncates
    wx1 = pos_x - xi.astype(F)
    wx0 = F(1) - wx1
    wy1 = pos_y - yi.astype(F)
    wy0 = F(1) - wy1
    x0, x1 = np.clip(xi, 0, W - 1), np.clip(xi + 1, 0, W - 1)
    y0, y1 = np.clip(yi, 0, H - 1), np.clip((pos_y + F(1)).astype(np.int64), 0, H - 1)
    c = np.zeros((nf * RR, 3), F)
    c = c + image[y0, x0] * (wx0 * wy0)[:, None]
    c = c + image[y1, x0] * (wx0 * wy1)[:, None]
    c = c + image[y0, x1] * (wx1 * wy0)[:, None]
    c = c + image[y1, x1] * (wx1 * wy1)[:, None]
    upd = np.asarray(is_update)[i // RR] != 0
    flat = out.reshape(-1, 3)
    flat[upd] = c[upd]
    return out


def atlas_layout(num_faces, texture_res):
    """Tile grid and per-face atlas triangle in pixels (functional/save_obj.py:14-27)."""
    tile_width = int((num_faces - 1.) ** 0.5) + 1
    tile_height = int((num_faces - 1.) / tile_width) + 1
    fn = np.arange(num_faces)
    column, row = fn % tile_width, fn // tile_width
    uv = np.zeros((num_faces, 3, 2), F)
    uv[:, 0, 0] = column * texture_res + texture_res / 2
    uv[:, 0, 1] = row * texture_res + 1
    uv[:, 1, 0] = column * texture_res + 1
    uv[:, 1, 1] = (row + 1) * texture_res - 1 - 1
    uv[:, 2, 0] = (column + 1) * texture_res - 1 - 1
    uv[:, 2, 1] = (row + 1) * texture_res - 1 - 1
    return tile_width, tile_height, uv


def create_texture_image_kernel(face_uv, textures, image, tile_width, eps=1e-5):
    """face_uv [nf,3,2] (pixels), textures [nf,R*R,3], image [rows,cols,3] -> painted copy of image."""
    uv, tex = np.asarray(face_uv, F), np.asarray(textures, F)
    img = np.array(image, dtype=F, copy=True)
    nf, RR = tex.shape[:2]
    R = int(np.sqrt(RR))
    rows, cols = img.shape[:2]
    R_out = cols // tile_width
    eps = F(eps)
    i = np.arange(rows * cols)
    x, y = i % cols, i // cols
    fn = x // R_out + (y // R_out) * tile_width
    live = fn < nf
    i, x, y, fn = i[live], x[live], y[live], fn[live]
    p = uv[fn]
    p0x, p0y, p1x, p1y, p2x, p2y = p[:, 0, 0], p[:, 0, 1], p[:, 1, 0], p[:, 1, 1], p[:, 2, 0], p[:, 2, 1]
    den = ((p2x * (p0y - p1y) + p0x * (p1y - p2y)) + p1x * (p2y - p0y)) + eps
    inv = [(p1y - p2y) / den, (p2x - p1x) / den, (p1x * p2y - p2x * p1y) / den,
           (p2y - p0y) / den, (p0x - p2x) / den, (p2x * p0y - p0x * p2y) / den,
           (p0y - p1y) / den, (p1x - p0x) / den, (p0x * p1y - p1x * p0y) / den]
    xf, yf = x.astype(F), y.astype(F)
    w, w_sum = [], np.zeros(len(i), F)
    for k in range(3):
        wk = (inv[3 * k] * xf + inv[3 * k + 1] * yf) + inv[3 * k + 2]
        wk = np.maximum(np.minimum(wk.astype(np.float64), 1.), 0.).astype(F)
        w.append(wk)
        w_sum = w_sum + wk
    wn0, wn1 = w[0] / (w_sum + eps), w[1] / (w_sum + eps)
    w_x, w_y = (wn0 * F(R)).astype(np.int64), (wn1 * F(R)).astype(np.int64)
    lower = ((wn0 + wn1) * F(R) - w_x.astype(F)) - w_y.astype(F) <= F(1)
    texel = np.where(lower, w_y * R + w_x, (R - 1 - w_y) * R + (R - 1 - w_x))
    texel = np.clip(texel, 0, RR - 1)
    img.reshape(-1, 3)[i] = tex[fn, texel]
    return img


def create_texture_image(textures, texture_res=16):
    """functional/save_obj.py:13-41: returns (image flipped vertically [rows,cols,3], face uv in [0,1] [nf,3,2])."""
    textures = np.asarray(textures, F)
    nf = textures.shape[0]
    tile_width, tile_height, uv = atlas_layout(nf, texture_res)
    image = np.ones((tile_height * texture_res, tile_width * texture_res, 3), F)
    image = create_texture_image_kernel(uv, textures, image, tile_width, 1e-5)
    uv = uv.copy()
    uv[:, :, 0] /= F(image.shape[1] - 1)
    uv[:, :, 1] /= F(image.shape[0] - 1)
    return image[::-1], uv
