"""Wavefront OBJ reading / writing on the host (plain Python + torch).

Counterparts of the reference's ``gendr/functional/load_obj.py:108-172`` and ``save_obj.py:52-106`` for
geometry and per-vertex colours.  Surface texture atlases go through the reference's ``load_textures`` /
``create_texture_image`` CUDA kernels (not rebuilt yet): asking for them raises
``NotImplementedError`` instead of silently returning something else.  No scikit-image import at module load.
"""
import os

import numpy as np
import torch


def _default_device():
    return torch.device('cuda') if torch.cuda.is_available() else torch.device('cpu')


def load_obj(filename_obj, normalization=False, load_texture=False, texture_res=4, texture_type='surface'):
    """Returns (vertices [nv,3] float32, faces [nf,3] int32[, textures]).  Polygons are fan-triangulated;
    only the vertex index of ``f a/b/c`` entries is used."""
    assert texture_type in ['surface', 'vertex']
    positions, colours, triangles = [], [], []
    with open(filename_obj) as fh:
        for line in fh:
            parts = line.split()
            if not parts:
                continue
            if parts[0] == 'v':
                positions.append([float(x) for x in parts[1:4]])
                colours.append([float(x) for x in parts[4:7]])
            elif parts[0] == 'f':
                idx = [int(tok.split('/')[0]) for tok in parts[1:]]
                for k in range(1, len(idx) - 1):
                    triangles.append((idx[0], idx[k], idx[k + 1]))
    dev = _default_device()
    vertices = torch.from_numpy(np.asarray(positions, dtype=np.float32)).to(dev)
    faces = torch.from_numpy(np.asarray(triangles, dtype=np.int32)).to(dev) - 1

    textures = None
    if load_texture and texture_type == 'surface':
        raise NotImplementedError('surface texture atlases need the load_textures kernel (SURVEY.md f-3), not rebuilt yet')
    if load_texture and texture_type == 'vertex':
        textures = torch.from_numpy(np.asarray(colours, dtype=np.float32)).to(dev)

    if normalization:                       # into a unit cube centred at zero (load_obj.py:162-166)
        vertices -= vertices.min(0)[0][None, :]
        vertices /= torch.abs(vertices).max()
        vertices *= 2
        vertices -= vertices.max(0)[0][None, :] / 2

    return (vertices, faces, textures) if load_texture else (vertices, faces)


def save_obj(filename, vertices, faces, textures=None, texture_res=16, texture_type='surface'):
    assert vertices.ndimension() == 2 and faces.ndimension() == 2
    assert texture_type in ['surface', 'vertex']
    assert texture_res >= 2
    if textures is not None and texture_type == 'surface':
        raise NotImplementedError('surface texture atlases need the create_texture_image kernel (SURVEY.md f-3), not rebuilt yet')
    v = vertices.detach().cpu().numpy()
    f = faces.detach().cpu().numpy()
    with open(filename, 'w') as fh:
        fh.write('# %s\n#\n\n' % os.path.basename(filename))
        if textures is not None:
            c = textures.detach().cpu().numpy()
            for p, col in zip(v, c):
                fh.write('v %.8f %.8f %.8f %.8f %.8f %.8f\n' % (p[0], p[1], p[2], col[0], col[1], col[2]))
        else:
            for p in v:
                fh.write('v %.8f %.8f %.8f\n' % (p[0], p[1], p[2]))
        fh.write('\n')
        for tri in f:
            fh.write('f %d %d %d\n' % (tri[0] + 1, tri[1] + 1, tri[2] + 1))


def save_voxel(filename, voxel):
    """Occupied voxels as a vertex-only OBJ (``save_obj.py:95-106``)."""
    occ = np.argwhere(np.asarray(voxel) == 1).astype(np.float32)
    if occ.size:
        occ = occ / np.asarray(voxel.shape[:3], dtype=np.float32)
    return save_obj(filename, torch.from_numpy(occ.reshape(-1, 3)), torch.zeros((0, 3), dtype=torch.int32))


def voxelization(faces, size, normalize=False):
    """Mesh -> ``[B, size, size, size]`` int32 occupancy (1 = surface or enclosed).  ``faces`` ``[B, nf, 3, 3]`` in
    unit-cube coordinates (scaled by ``size`` unless ``normalize``; reference ``functional/voxelization.py:47-62``).
    Two HIP launches (``csrc/gendr_voxel.h``) instead of the reference's four kernels + host-synchronised sweep loop;
    CUDA tensors only -- there is no CPU path."""
    from .. import _native
    from .renderer import check
    if not torch.is_tensor(faces) or not faces.is_cuda:
        raise TypeError('voxelization only supports CUDA Tensors')
    if faces.dim() == 3 and faces.shape[2] == 9:
        faces = faces.reshape(faces.shape[0], faces.shape[1], 3, 3)
    if faces.dim() != 4 or faces.shape[2:] != (3, 3):
        raise ValueError('faces must be [B, nf, 3, 3]; got %s' % (tuple(faces.shape),))
    size = int(size)
    faces = faces.detach().to(torch.float32).clone()
    if not normalize:
        faces *= size
    B, nf = faces.shape[:2]
    lib = _native.lib()
    voxels = torch.empty(B, size, size, size, dtype=torch.int32, device=faces.device)
    ws_bytes = lib.gendr_voxelize_workspace_bytes(B, size)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=faces.device) if ws_bytes else None
    with torch.cuda.device(faces.device):
        check(lib.gendr_voxelize(faces.data_ptr(), voxels.data_ptr(), ws.data_ptr() if ws is not None else None,
                                 B, nf, size, torch.cuda.current_stream(faces.device).cuda_stream), 'gendr_voxelize')
    return voxels
