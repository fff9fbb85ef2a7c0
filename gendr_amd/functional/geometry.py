"""Camera and mesh helpers in plain PyTorch (the steps right before the hot path).

Behavioural counterparts of the reference's ``gendr/functional/look_at.py:11-68``,
``look.py``, ``get_points_from_angles.py:9-29``, ``face_vertices.py:9-27`` and
``gendr/transform.py:14-47`` (``perspective`` / ``orthogonal``).  These are not
kernels: a handful of elementwise / gather ops; SURVEY.md row f-1 lists fusing
them; ``projection.py`` is that fusion (HIP), this module stays as the CPU-capable unfused composition.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def _as_vec3(v, device, batch):
    """list / tuple / ndarray / tensor -> float32 tensor [batch, 3] on device."""
    if isinstance(v, (list, tuple)):
        v = torch.tensor(v, dtype=torch.float32, device=device)
    elif isinstance(v, np.ndarray):
        v = torch.from_numpy(v).to(device)
    elif torch.is_tensor(v):
        v = v.to(device)
    if v.dim() == 1:
        v = v[None, :]
    if v.dim() == 2 and v.shape[0] == 1 and batch != 1:      # [1,3] broadcasts over the batch, as in the reference
        v = v.expand(batch, v.shape[1])
    return v


def get_points_from_angles(distance, elevation, azimuth, degrees=True):
    """Spherical (distance, elevation, azimuth) -> eye position; scalars give a tuple,
    tensors a ``[N, 3]`` tensor (``get_points_from_angles.py:9-29``)."""
    if isinstance(distance, (float, int)):
        if degrees:
            elevation, azimuth = math.radians(elevation), math.radians(azimuth)
        ce = math.cos(elevation)
        return (distance * ce * math.sin(azimuth), distance * math.sin(elevation), -distance * ce * math.cos(azimuth))
    if degrees:
        elevation = math.pi / 180. * elevation
        azimuth = math.pi / 180. * azimuth
    ce = torch.cos(elevation)
    return torch.stack([distance * ce * torch.sin(azimuth),
                        distance * torch.sin(elevation),
                        -distance * ce * torch.cos(azimuth)]).transpose(1, 0)


def _camera_rotation(z_axis, up):
    # eps = 1e-5 as in the reference (look_at.py:52-56)
    z_axis = F.normalize(z_axis, eps=1e-5)
    x_axis = F.normalize(torch.cross(up, z_axis, dim=1), eps=1e-5)
    y_axis = F.normalize(torch.cross(z_axis, x_axis, dim=1), eps=1e-5)
    return torch.stack((x_axis, y_axis, z_axis), dim=1)      # [B, 3, 3], rows = camera axes


def look_at(vertices, eye, at=[0, 0, 0], up=[0, 1, 0], only_rotate=False):
    """World -> camera coordinates for a camera at ``eye`` looking at ``at``."""
    if vertices.ndimension() != 3:
        raise ValueError('vertices Tensor should have 3 dimensions')
    B, dev = vertices.shape[0], vertices.device
    eye, at, up = _as_vec3(eye, dev, B), _as_vec3(at, dev, B), _as_vec3(up, dev, B)
    rot = _camera_rotation(at - eye, up)
    if not only_rotate:
        vertices = vertices - eye[:, None, :]
    return torch.matmul(vertices, rot.transpose(1, 2))


def look(vertices, eye, direction=[0, 1, 0], up=None):
    """World -> camera coordinates for a camera at ``eye`` looking along ``direction`` (``look.py``)."""
    if vertices.ndimension() != 3:
        raise ValueError('vertices Tensor should have 3 dimensions')
    B, dev = vertices.shape[0], vertices.device
    if up is None:
        up = [0., 1., 0.]
    eye, direction, up = _as_vec3(eye, dev, B), _as_vec3(direction, dev, B), _as_vec3(up, dev, B)
    rot = _camera_rotation(direction, up)
    return torch.matmul(vertices - eye[:, None, :], rot.transpose(1, 2))


def perspective(vertices, angle=30.):
    """x,y divided by z * tan(angle) (``transform.py:14-29``); z is kept."""
    if vertices.ndimension() != 3:
        raise ValueError('vertices Tensor should have 3 dimensions')
    width = torch.tan(torch.tensor(angle / 180 * math.pi, dtype=torch.float32, device=vertices.device))
    z = vertices[:, :, 2]
    return torch.stack((vertices[:, :, 0] / z / width, vertices[:, :, 1] / z / width, z), dim=2)


def orthogonal(vertices, scale=1.):
    if vertices.ndimension() != 3:
        raise ValueError('vertices Tensor should have 3 dimensions')
    return torch.stack((vertices[:, :, 0] * scale, vertices[:, :, 1] * scale, vertices[:, :, 2]), dim=2)


def face_vertices(vertices, faces):
    """Gather ``[B, nv, 3]`` vertices by ``[B, nf, 3]`` indices -> ``[B, nf, 3, 3]`` (``face_vertices.py:9-27``)."""
    assert vertices.ndimension() == 3 and faces.ndimension() == 3
    assert vertices.shape[0] == faces.shape[0] and vertices.shape[2] == 3 and faces.shape[2] == 3
    B, nv = vertices.shape[:2]
    offset = (torch.arange(B, device=vertices.device) * nv)[:, None, None]
    return vertices.reshape(B * nv, 3)[(faces.long() + offset)]


def vertex_normals(vertices, faces):
    """Area-weighted vertex normals by scatter-add of face cross products (``vertex_normals.py``)."""
    assert vertices.ndimension() == 3 and faces.ndimension() == 3
    B, nv = vertices.shape[:2]
    flat = vertices.reshape(B * nv, 3)
    idx = (faces.long() + (torch.arange(B, device=vertices.device) * nv)[:, None, None]).reshape(-1, 3)
    tri = flat[idx]                                          # [B*nf, 3, 3]
    normals = torch.zeros(B * nv, 3, dtype=vertices.dtype, device=vertices.device)
    for a, b, c in ((1, 2, 0), (2, 0, 1), (0, 1, 2)):
        normals.index_add_(0, idx[:, a], torch.cross(tri[:, b] - tri[:, a], tri[:, c] - tri[:, a], dim=1))
    return F.normalize(normals, eps=1e-6, dim=1).reshape(B, nv, 3)
