"""The lighting oracle (oracle/light_ref.py) against vectors produced by the reference's own functional/lighting.py
(tests/golden/glue/glue.npz)."""
import os

import numpy as np

from oracle import light_ref as L

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'glue', 'glue.npz'))


def test_ambient_and_directional_match_reference_vectors():
    light = L.ambient(np.zeros_like(G['ambient']), 0.4, (1.0, 0.9, 0.8))
    np.testing.assert_allclose(light, G['ambient'], rtol=0, atol=1e-7)
    light = L.directional(light, G['vertex_normals'], 0.6, (0.7, 1.0, 1.0), (0.3, 1.0, -0.2))
    np.testing.assert_allclose(light, G['directional'], rtol=0, atol=2e-7)


def test_surface_normals_are_unit_and_orthogonal_to_the_face():
    n = L.surface_normals(G['vertices'], G['faces'])
    fv = G['face_vertices']
    assert np.abs(np.linalg.norm(n, axis=-1) - 1).max() < 1e-6
    assert np.abs((n * (fv[:, :, 0] - fv[:, :, 1])).sum(-1)).max() < 1e-6
    # a degenerate face keeps a finite normal (the clamp at 1e-6)
    v = np.zeros((1, 3, 3), np.float32)
    assert np.isfinite(L.surface_normals(v, np.array([[[0, 1, 2]]]))).all()


def test_light_faces_composition():
    tex = np.random.default_rng(0).random((3, 320, 4, 3)).astype(np.float32)
    out = L.light_faces(G['vertices'], G['faces'], tex, 0.5, (1, 1, 1), [(0.5, (1, 1, 1), (0, 1, 0))])
    n = L.surface_normals(G['vertices'], G['faces'])
    want = tex * (0.5 + 0.5 * np.maximum(n[..., 1], 0))[:, :, None, None]
    np.testing.assert_allclose(out, want, rtol=0, atol=1e-6)
