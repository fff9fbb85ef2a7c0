"""CPU oracle for the fused projection.  TEST INFRASTRUCTURE ONLY -- nothing under
gendr_amd/ may import this.

numpy restatement of the reference's tensor chain, stage by stage:
  camera_rotation  gendr/functional/look_at.py:52-59 (F.normalize with eps=1e-5, cross products)
  look_at          gendr/functional/look_at.py:61-67 (subtract eye, matmul by R^T)
  perspective      gendr/transform.py:14-29           (x / z / tan(angle), y likewise, z kept)
  orthogonal       gendr/transform.py:32-47           (x * scale, y * scale, z kept)
  face_vertices    gendr/functional/face_vertices.py:24-27 (gather by face index)
Pinned by tests/golden/glue/glue.npz (outputs of the reference's own look_at / look / face_vertices modules);
perspective / orthogonal live in a module that imports the CUDA extension and cannot be imported here, they
are restated from the source text (three expressions).
"""
import math

import numpy as np


def _normalize(v, eps=1e-5):
    n = np.sqrt((v * v).sum(axis=1, keepdims=True))
    return v / np.maximum(n, np.asarray(eps, dtype=v.dtype))


def camera_rotation(z_axis, up):
    z = _normalize(z_axis)
    x = _normalize(np.cross(up, z))
    y = _normalize(np.cross(z, x))
    return np.stack([x, y, z], axis=1)                      # [B,3,3], rows = axes


def look_at(vertices, eye, at=(0, 0, 0), up=(0, 1, 0), dtype=np.float32):
    v = np.asarray(vertices, dtype=dtype)
    B = v.shape[0]
    bc = lambda a: np.broadcast_to(np.asarray(a, dtype=dtype).reshape(-1, 3), (B, 3))
    eye, at, up = bc(eye), bc(at), bc(up)
    R = camera_rotation(at - eye, up)
    d = v - eye[:, None, :]
    # out[n,i] = sum_j d[n,j] R[i,j], summed j = 0,1,2 left to right like the HIP kernel
    return (d[:, :, None, 0] * R[:, None, :, 0] + d[:, :, None, 1] * R[:, None, :, 1]) + d[:, :, None, 2] * R[:, None, :, 2]


def look(vertices, eye, direction=(0, 1, 0), up=(0, 1, 0), dtype=np.float32):
    v = np.asarray(vertices, dtype=dtype)
    B = v.shape[0]
    bc = lambda a: np.broadcast_to(np.asarray(a, dtype=dtype).reshape(-1, 3), (B, 3))
    eye, direction, up = bc(eye), bc(direction), bc(up)
    R = camera_rotation(direction, up)
    d = v - eye[:, None, :]
    return (d[:, :, None, 0] * R[:, None, :, 0] + d[:, :, None, 1] * R[:, None, :, 1]) + d[:, :, None, 2] * R[:, None, :, 2]


def perspective(vertices, angle=30.):
    dt = vertices.dtype
    width = np.tan(np.asarray(angle / 180 * math.pi, dtype=np.float32)).astype(dt)
    z = vertices[:, :, 2]
    return np.stack([vertices[:, :, 0] / z / width, vertices[:, :, 1] / z / width, z], axis=2)


def orthogonal(vertices, scale=1.):
    s = np.asarray(scale, dtype=vertices.dtype)
    return np.stack([vertices[:, :, 0] * s, vertices[:, :, 1] * s, vertices[:, :, 2]], axis=2)


def face_vertices(vertices, faces):
    B = vertices.shape[0]
    f = np.broadcast_to(faces, (B,) + faces.shape[1:]).astype(np.int64)
    return np.stack([vertices[b][f[b]] for b in range(B)])   # [B,nf,3,3]


def look_at_faces(vertices, faces, eye, at=(0, 0, 0), up=(0, 1, 0), perspective_=True, viewing_angle=30.,
                  viewing_scale=1., dtype=np.float32):
    cam = look_at(vertices, eye, at, up, dtype)
    proj = perspective(cam, viewing_angle) if perspective_ else orthogonal(cam, viewing_scale)
    return face_vertices(proj, np.asarray(faces))
