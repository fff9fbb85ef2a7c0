"""Camera transforms as ``nn.Module``s mapping a Mesh to a Mesh (plain PyTorch glue; reference
``gendr/transform.py:49-168``)."""
import math
import os

import numpy as np
import torch
import torch.nn as nn

from . import functional as Fn
from .functional.geometry import perspective, orthogonal     # noqa: F401  (re-exported names)
from .mesh import Mesh


class Transform(nn.Module):
    def transform(self, vertices):
        raise NotImplementedError()

    def forward(self, mesh):
        return Mesh(self.transform(mesh.vertices), mesh.faces, mesh.textures, mesh.texture_res, mesh.texture_type)


class Projection(Transform):
    """3x4 projection matrix + radial / tangential distortion (``transform.py:64-105``)."""

    def __init__(self, P, dist_coeffs=None, orig_size=512):
        super().__init__()
        if isinstance(P, np.ndarray):
            P = torch.from_numpy(P).to('cuda' if torch.cuda.is_available() else 'cpu')
        if P is None or P.ndimension() != 3 or P.shape[1] != 3 or P.shape[2] != 4:
            raise ValueError('You need to provide a valid (batch_size)x3x4 projection matrix')
        if dist_coeffs is None:
            dist_coeffs = torch.zeros(P.shape[0], 5, dtype=torch.float32, device=P.device)
        self.P, self.dist_coeffs, self.orig_size = P, dist_coeffs, orig_size

    def transform(self, vertices):
        hom = torch.cat([vertices, torch.ones_like(vertices[:, :, :1])], dim=-1)
        cam = torch.bmm(hom, self.P.transpose(2, 1))
        x, y, z = cam[:, :, 0], cam[:, :, 1], cam[:, :, 2]
        x_, y_ = x / (z + 1e-5), y / (z + 1e-5)
        k1, k2, p1, p2, k3 = (self.dist_coeffs[:, None, i] for i in range(5))
        r2 = x_ ** 2 + y_ ** 2
        radial = 1 + k1 * r2 + k2 * r2 ** 2 + k3 * r2 ** 3
        xd = x_ * radial + 2 * p1 * x_ * y_ + p2 * (r2 + 2 * x_ ** 2)
        yd = y_ * radial + p1 * (r2 + 2 * y_ ** 2) + 2 * p2 * x_ * y_
        half = self.orig_size / 2.
        return torch.stack([2 * (xd - half) / self.orig_size, 2 * (yd - half) / self.orig_size, z], dim=-1)


class _ProjectedMesh(Mesh):
    """Mesh seen through a camera.  ``face_vertices`` (what the renderer consumes) comes from the fused HIP
    projection kernel (``functional/projection.py``); ``vertices`` is still available and is
    computed by the unfused PyTorch composition on first use."""

    def __init__(self, mesh, camera, eye, target, target_is_direction):
        self.__dict__.update(mesh.__dict__)
        self._source_vertices = mesh.vertices
        self._vertices = None
        self._camera, self._eye_t, self._target, self._is_dir = camera, eye, target, target_is_direction

    @property
    def vertices(self):
        if self._vertices is None:
            self._vertices = self._camera.transform(self._source_vertices)
        return self._vertices

    @property
    def face_vertices(self):
        cam = self._camera
        fn = Fn.look_faces if self._is_dir else Fn.look_at_faces
        return fn(self._source_vertices, self._faces, self._eye_t, self._target, [0., 1., 0.],
                  cam.perspective, cam.viewing_angle, cam.viewing_scale)

    @property
    def surface_normals(self):
        self.vertices
        return Mesh.surface_normals.fget(self)

    @property
    def vertex_normals(self):
        self.vertices
        return Mesh.vertex_normals.fget(self)


def _fused_ok(mesh):
    return (os.environ.get('GENDR_FUSED_PROJECTION', '1') != '0' and mesh.vertices.is_cuda
            and mesh.vertices.dtype == torch.float32)


class _Camera(Transform):
    def __init__(self, perspective=True, viewing_angle=30, viewing_scale=1.0, eye=None):
        super().__init__()
        self.perspective = perspective
        self.viewing_angle = viewing_angle
        self.viewing_scale = viewing_scale
        self._eye = eye if eye is not None else [0, 0, -(1. / math.tan(math.radians(viewing_angle)) + 1)]

    def set_eyes(self, eyes):
        self._eye = eyes

    @property
    def eyes(self):
        return self._eye

    def _project(self, vertices):
        if self.perspective:
            return perspective(vertices, angle=self.viewing_angle)
        return orthogonal(vertices, scale=self.viewing_scale)


class LookAt(_Camera):
    def set_eyes_from_angles(self, distances, elevations, azimuths):
        self._eye = Fn.get_points_from_angles(distances, elevations, azimuths)

    def transform(self, vertices):
        return self._project(Fn.look_at(vertices, self._eye))

    def forward(self, mesh):
        if not _fused_ok(mesh):
            return super().forward(mesh)
        return _ProjectedMesh(mesh, self, self._eye, [0., 0., 0.], False)


class Look(_Camera):
    def __init__(self, camera_direction=[0, 0, 1], perspective=True, viewing_angle=30, viewing_scale=1.0, eye=None):
        super().__init__(perspective, viewing_angle, viewing_scale, eye)
        self.camera_direction = camera_direction

    def transform(self, vertices):
        return self._project(Fn.look(vertices, self._eye, self.camera_direction))

    def forward(self, mesh):
        if not _fused_ok(mesh):
            return super().forward(mesh)
        return _ProjectedMesh(mesh, self, self._eye, self.camera_direction, True)
