"""Shared inputs for the voxelization tests (unit-cube coordinates, like Mesh.voxelize feeds them)."""
import numpy as np

from gendr_amd.synthetic import icosphere


def box_faces(lo=0.25, hi=0.75):
    """12 triangles of an axis-aligned box -> [1,12,3,3] fp32."""
    c = np.array([[x, y, z] for x in (lo, hi) for y in (lo, hi) for z in (lo, hi)], np.float32)
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    tris = [(q[0], q[1], q[2]) for q in quads] + [(q[0], q[2], q[3]) for q in quads]
    return c[np.array(tris)][None]


def sphere_faces(batch=3, level=2, radius=0.7, seed=0, jitter=0.02):
    v, f = icosphere(level)
    rng = np.random.default_rng(seed)
    out = []
    for b in range(batch):
        vb = v * radius * (1.0 + 0.3 * b / max(batch - 1, 1)) + 0.5 + jitter * rng.standard_normal(v.shape)
        out.append(vb.astype(np.float32)[f])
    return np.stack(out)


def nested_shells():
    a = sphere_faces(1, 2, 0.9, 1, 0.0)
    b = sphere_faces(1, 2, 0.4, 2, 0.0)
    return np.concatenate([a, b], axis=1)


def soup(batch=2, nf=200, seed=3):
    rng = np.random.default_rng(seed)
    centre = rng.uniform(-0.1, 1.1, (batch, nf, 1, 3))
    return (centre + 0.15 * rng.standard_normal((batch, nf, 3, 3))).astype(np.float32)
