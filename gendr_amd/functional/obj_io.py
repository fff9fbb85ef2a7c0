"""Wavefront OBJ reading / writing on the host (plain Python + torch).

Counterparts of the reference's ``gendr/functional/load_obj.py:108-172`` and ``save_obj.py:52-106`` for
geometry and per-vertex colours.  Surface texture atlases (``load_textures`` ``load_obj.py:32-107``,
``create_texture_image`` ``save_obj.py:13-41``) run the two HIP kernels of ``csrc/gendr_texture.h`` (SURVEY.md row
f-3) and therefore need a GPU; images are read / written with Pillow (the reference uses scikit-image, absent here).
"""
import os

import numpy as np
import torch


def _default_device():
    return torch.device('cuda') if torch.cuda.is_available() else torch.device('cpu')


def _native_call():
    from .. import _native
    from .renderer import check
    return _native.lib(), check


def _need_cuda(what):
    if not torch.cuda.is_available():
        raise RuntimeError('%s runs a HIP kernel and needs a GPU (there is no CPU path)' % what)
    return torch.device('cuda')


def load_mtl(filename_mtl):
    """Diffuse colours (Kd) and texture file names (map_Kd) per material (``load_obj.py:14-29``)."""
    texture_filenames, colors, material_name = {}, {}, ''
    with open(filename_mtl) as fh:
        for line in fh:
            parts = line.split()
            if not parts:
                continue
            if parts[0] == 'newmtl':
                material_name = parts[1]
            elif parts[0] == 'map_Kd':
                texture_filenames[material_name] = parts[1]
            elif parts[0] == 'Kd':
                colors[material_name] = np.array([float(x) for x in parts[1:4]])
    return colors, texture_filenames


def _read_image(path):
    from PIL import Image
    with Image.open(path) as im:
        if im.mode not in ('L', 'RGB', 'RGBA'):
            im = im.convert('RGB')
        image = np.asarray(im).astype(np.float32) / 255.
    if image.ndim == 2:                                  # grey -> three channels
        image = np.stack((image,) * 3, -1)
    if image.shape[2] == 4:                              # alpha is ignored
        image = image[:, :, :3]
    return image


def load_textures(filename_obj, filename_mtl, texture_res):
    """Per-face texel blocks ``[nf, texture_res**2, 3]`` sampled from the materials' images (``load_obj.py:32-107``)."""
    dev = _need_cuda('load_textures')
    uvs, tri_uv, material_names, material_name = [], [], [], ''

    def vt_index(tok):
        return int(tok.split('/')[1]) if '/' in tok and '//' not in tok else 0

    with open(filename_obj) as fh:
        lines = fh.readlines()
    for line in lines:
        parts = line.split()
        if parts and parts[0] == 'vt':
            uvs.append([float(x) for x in parts[1:3]])
    for line in lines:
        parts = line.split()
        if not parts:
            continue
        if parts[0] == 'f':
            idx = [vt_index(tok) for tok in parts[1:]]
            for k in range(1, len(idx) - 1):
                tri_uv.append((idx[0], idx[k], idx[k + 1]))
                material_names.append(material_name)
        elif parts[0] == 'usemtl':
            material_name = parts[1]
    uvs = np.asarray(uvs, dtype=np.float32).reshape(-1, 2)
    face_uv = torch.from_numpy(uvs[np.asarray(tri_uv, dtype=np.int32).reshape(-1, 3) - 1]).to(dev)     # [nf,3,2]
    face_uv[1 < face_uv] = face_uv[1 < face_uv] % 1                                                     # wrap, load_obj.py:72
    face_uv = face_uv.contiguous()
    nf = face_uv.shape[0]

    colors, texture_filenames = load_mtl(filename_mtl)
    textures = torch.ones(nf, texture_res ** 2, 3, dtype=torch.float32, device=dev)
    names = np.array(material_names)
    for name, color in colors.items():
        sel = torch.from_numpy(names == name).to(dev)
        textures[sel] = torch.from_numpy(color.astype(np.float32)).to(dev)[None, None, :]

    lib, check = _native_call()
    for name, filename_texture in texture_filenames.items():
        image = _read_image(os.path.join(os.path.dirname(filename_obj), filename_texture))
        image = torch.from_numpy(image[::-1].copy()).to(dev).contiguous()                              # flipped, load_obj.py:104
        is_update = torch.from_numpy((names == name).astype(np.int32)).to(dev)
        with torch.cuda.device(dev):
            check(lib.gendr_load_textures(image.data_ptr(), face_uv.data_ptr(), is_update.data_ptr(), textures.data_ptr(),
                                          nf, int(texture_res), image.shape[0], image.shape[1],
                                          torch.cuda.current_stream(dev).cuda_stream), 'gendr_load_textures')
    return textures


def create_texture_image(textures, texture_res=16):
    """Texel blocks ``[nf, R*R, 3]`` -> (atlas image ``[rows, cols, 3]`` numpy, flipped vertically; per-face atlas
    triangle ``[nf, 3, 2]`` numpy in [0,1]) (``save_obj.py:13-41``)."""
    dev = _need_cuda('create_texture_image')
    textures = textures.detach().to(dev, torch.float32).contiguous()
    nf = textures.shape[0]
    R_in = int(np.sqrt(textures.shape[1]))
    tile_width = int((nf - 1.) ** 0.5) + 1
    tile_height = int((nf - 1.) / tile_width) + 1
    image = torch.ones(tile_height * texture_res, tile_width * texture_res, 3, dtype=torch.float32, device=dev)
    fn = torch.arange(nf)
    column, row = fn % tile_width, fn // tile_width
    uv = torch.zeros(nf, 3, 2, dtype=torch.float32)
    uv[:, 0, 0] = column * texture_res + texture_res / 2
    uv[:, 0, 1] = row * texture_res + 1
    uv[:, 1, 0] = column * texture_res + 1
    uv[:, 1, 1] = (row + 1) * texture_res - 1 - 1
    uv[:, 2, 0] = (column + 1) * texture_res - 1 - 1
    uv[:, 2, 1] = (row + 1) * texture_res - 1 - 1
    uv_d = uv.to(dev).contiguous()
    lib, check = _native_call()
    with torch.cuda.device(dev):
        check(lib.gendr_create_texture_image(uv_d.data_ptr(), textures.data_ptr(), image.data_ptr(), nf, R_in,
                                             image.shape[0], image.shape[1], tile_width, 1e-5,
                                             torch.cuda.current_stream(dev).cuda_stream), 'gendr_create_texture_image')
    uv[:, :, 0] /= (image.shape[1] - 1)
    uv[:, :, 1] /= (image.shape[0] - 1)
    return image.cpu().numpy()[::-1, ::1], uv.numpy()


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
        with open(filename_obj) as fh:
            for line in fh:
                if line.startswith('mtllib'):
                    filename_mtl = os.path.join(os.path.dirname(filename_obj), line.split()[1])
                    textures = load_textures(filename_obj, filename_mtl, texture_res)
        if textures is None:
            raise Exception('Failed to load textures.')
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
    v = vertices.detach().cpu().numpy()
    f = faces.detach().cpu().numpy()
    if textures is not None and texture_type == 'surface':           # atlas png + mtl + vt records (save_obj.py:49-93)
        from PIL import Image
        filename_mtl, filename_texture = filename[:-4] + '.mtl', filename[:-4] + '.png'
        texture_image, face_uv = create_texture_image(textures, texture_res)
        Image.fromarray((texture_image.clip(0, 1) * 255).astype('uint8')).save(filename_texture)
        with open(filename, 'w') as fh:
            fh.write('# %s\n#\n\n' % os.path.basename(filename))
            fh.write('mtllib %s\n\n' % os.path.basename(filename_mtl))
            for p in v:
                fh.write('v %.8f %.8f %.8f\n' % (p[0], p[1], p[2]))
            fh.write('\n')
            for t in face_uv.reshape((-1, 2)):
                fh.write('vt %.8f %.8f\n' % (t[0], t[1]))
            fh.write('\nusemtl material_1\n')
            for i, tri in enumerate(f):
                fh.write('f %d/%d %d/%d %d/%d\n' % (tri[0] + 1, 3 * i + 1, tri[1] + 1, 3 * i + 2, tri[2] + 1, 3 * i + 3))
            fh.write('\n')
        with open(filename_mtl, 'w') as fh:
            fh.write('newmtl material_1\n')
            fh.write('map_Kd %s\n' % os.path.basename(filename_texture))
        return
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
