"""Synthetic scenes for the benchmark and the tests (no dataset or network access).

The headline workload of BASELINE.json is "~1k faces at 256^2, batch 64": a procedurally generated
1280-face icosphere (the topology class of the reference's ``experiments/data/sphere_642.obj``) with an
anisotropic scale, seen from the camera ring of ``experiments/train_reconstruction.py:284-285,343``
(distance 2.732, elevation 30 deg, azimuth -15 deg * i, ``LookAt(viewing_angle=15)``).
"""
import math

import numpy as np
import torch

from .functional.geometry import face_vertices, get_points_from_angles, look_at, perspective


def icosphere(subdivisions=3, radius=0.5):
    """Returns (vertices [nv,3] float32, faces [nf,3] int64); 3 subdivisions = 642 v / 1280 f."""
    t = (1.0 + math.sqrt(5.0)) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t),
         (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    verts = [np.array(p, dtype=np.float64) / np.linalg.norm(p) for p in v]
    faces = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2),
             (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5),
             (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    for _ in range(subdivisions):
        cache, out = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = verts[a] + verts[b]
                verts.append(m / np.linalg.norm(m))
                cache[key] = len(verts) - 1
            return cache[key]

        for a, b, c in faces:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            out += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        faces = out
    return (np.asarray(verts) * radius).astype(np.float32), np.asarray(faces, dtype=np.int64)


def ring_cameras(n, distance=2.732, elevation=30.0):
    """Eye positions of the reconstruction experiment's camera ring: azimuth -15 deg * (i mod 24)."""
    az = torch.tensor([-15.0 * (i % 24) for i in range(n)], dtype=torch.float32)
    return get_points_from_angles(torch.full((n,), float(distance)), torch.full((n,), float(elevation)), az)


def benchmark_scene(batch, subdivisions=3, axes=(1.0, 0.35, 0.8), viewing_angle=15.0, texture='surface',
                    seed=0, device='cpu'):
    """face_vertices [B,nf,3,3] and textures [B,nf,T,3] of the synthetic headline workload."""
    g = torch.Generator().manual_seed(seed)
    verts, faces = icosphere(subdivisions)
    v = torch.from_numpy(verts) * torch.tensor(axes, dtype=torch.float32)
    # fixed-seed smooth radial displacement so that the views are not symmetric
    k = torch.randn(3, generator=g)
    v = v * (1.0 + 0.08 * torch.sin(4.0 * (v @ k)))[:, None]
    v = v[None].repeat(batch, 1, 1)
    f = torch.from_numpy(faces)[None].repeat(batch, 1, 1)
    cam = perspective(look_at(v, ring_cameras(batch)), angle=viewing_angle)
    fv = face_vertices(cam, f).contiguous()
    nf = f.shape[1]
    if texture == 'vertex':
        tex = torch.rand(batch, nf, 3, 3, generator=g)
    else:
        tex = torch.rand(batch, nf, 1, 3, generator=g)
    return fv.to(device), tex.to(device)


def unit_quad(device='cpu'):
    """BASELINE config 1: two faces, vertices (+-0.5, +-0.5, z=2)."""
    v = torch.tensor([[-.5, -.5, 2.], [.5, -.5, 2.], [.5, .5, 2.], [-.5, .5, 2.]])
    f = torch.tensor([[0, 1, 2], [0, 2, 3]])
    fv = v[f][None].contiguous()
    tex = torch.tensor([[[1., 0., 0.]], [[0., 1., 0.]]])[None].contiguous()
    return fv.to(device), tex.to(device)
