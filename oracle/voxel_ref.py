"""CPU oracle for voxelization.  TEST INFRASTRUCTURE ONLY -- nothing under gendr_amd/ may
import this.

numpy restatement, stage by stage and in fp32 without contraction, of
  voxelize_sub1_kernel  gendr/cuda/voxelization_cuda_kernel.cu:36-97   (axis-parallel rays, 2x2 neighbourhood)
  voxelize_sub2_kernel  :100-125                                        (vertex voxels)
  voxelize_sub3_kernel  :127-146                                        (empty boundary voxels are visible)
  voxelize_sub4_kernel  :148-194                                        (one flood sweep over interior voxels)
  and their driver      gendr/functional/voxelization.py:11-62          (axis permutations, union, sweep-until-stable)

PARITY PINNED to outputs of the reference's own kernels (the device half of voxelization_cuda_kernel.cu compiled for gfx950
by oracle/build_ref.py, launched by oracle/ref_gpu.py): every stage and the whole pipeline agree bit for bit on the GPU box
(tests/test_gpu_reference_pin_aux.py).  On the CPU side: closed-form cases in tests/test_voxel_oracle.py (axis-aligned box, nested shells, open
surfaces) and an independent connected-components formulation of the flood fill (scipy.ndimage.label).
"""
import numpy as np

F = np.float32


def _sub1(faces, vs, dim):
    """faces [B,nf,3,3] fp32 (voxel units), rays along original axis `dim` -> int32 [B,vs,vs,vs] in ORIGINAL axis order."""
    B, nf = faces.shape[:2]
    perm = {0: [2, 1, 0], 1: [0, 2, 1], 2: [0, 1, 2]}[dim]                   # voxelization.py:14-17
    f = faces[:, :, :, perm].reshape(B, nf, 9)
    vox = np.zeros((B, vs, vs, vs), np.int32)                               # kernel's [b, y, x, z]
    y = np.arange(vs, dtype=F)[:, None]                                     # thread (y, x): i % vs, (i / vs) % vs
    x = np.arange(vs, dtype=F)[None, :]
    yi0, xi0 = np.meshgrid(np.arange(vs), np.arange(vs), indexing='ij')
    with np.errstate(all='ignore'):
        for b in range(B):
            for fn in range(nf):
                c = f[b, fn]
                y1d, x1d, z1d = c[3] - c[0], c[4] - c[1], c[5] - c[2]
                y2d, x2d, z2d = c[6] - c[0], c[7] - c[1], c[8] - c[2]
                det = x1d * y2d - x2d * y1d
                if det == 0:
                    continue
                ypd, xpd = y - c[0], x - c[1]
                t1 = (y2d * xpd - x2d * ypd) / det
                t2 = (-y1d * xpd + x1d * ypd) / det
                hit = ~(t1 < 0) & ~(t2 < 0) & ~(F(1) < t1 + t2)
                if not hit.any():
                    continue
                zf = np.floor(t1 * z1d + t2 * z2d + c[2])
                zi = np.where(np.isfinite(zf), zf, -1).astype(np.int64)
                ok = hit & (zi >= 0) & (zi < vs)
                for dy, dx in ((0, 0), (-1, 0), (0, -1), (-1, -1)):
                    yy, xx = yi0 + dy, xi0 + dx
                    m = ok & (yy >= 0) & (xx >= 0)
                    vox[b, yy[m], xx[m], zi[m]] = 1
    return np.swapaxes(vox, dim + 1, 3)                                     # .transpose(dim + 1, -1), voxelization.py:19


def _sub2(faces, vs):
    B, nf = faces.shape[:2]
    vox = np.zeros((B, vs, vs, vs), np.int32)
    idx = np.floor(faces.reshape(B, nf * 3, 3)).astype(np.int64)
    for b in range(B):
        i = idx[b]
        m = ((i >= 0) & (i < vs)).all(axis=1)
        vox[b, i[m, 0], i[m, 1], i[m, 2]] = 1
    return vox


def surface(faces, size, normalize=False):
    faces = np.asarray(faces, dtype=F).copy()
    if not normalize:
        faces *= F(size)                                                    # voxelization.py:49-52 (normalize=True is a no-op there)
    v = _sub1(faces, size, 0) + _sub1(faces, size, 1) + _sub1(faces, size, 2) + _sub2(faces, size)
    return (v > 0).astype(np.int32)


def fill(voxels):
    """sub3 + sub4 swept until the visible count stops changing (voxelization.py:28-44); returns 1 - visible."""
    vs = voxels.shape[1]
    edge = np.zeros(voxels.shape[1:], bool)
    edge[0], edge[-1], edge[:, 0], edge[:, -1], edge[:, :, 0], edge[:, :, -1] = (True,) * 6
    empty = voxels == 0
    visible = empty & edge[None]
    interior = ~edge[None]
    while True:
        nb = np.zeros_like(visible)
        nb[:, 1:] |= visible[:, :-1]
        nb[:, :-1] |= visible[:, 1:]
        nb[:, :, 1:] |= visible[:, :, :-1]
        nb[:, :, :-1] |= visible[:, :, 1:]
        nb[:, :, :, 1:] |= visible[:, :, :, :-1]
        nb[:, :, :, :-1] |= visible[:, :, :, 1:]
        new = visible | (interior & empty & nb)
        if new.sum() == visible.sum():
            break
        visible = new
    return (1 - visible.astype(np.int32)).astype(np.int32)


def voxelization(faces, size, normalize=False):
    return fill(surface(faces, size, normalize))
