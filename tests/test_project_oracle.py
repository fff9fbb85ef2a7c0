"""The projection oracle (oracle/project_ref.py) against vectors produced by the reference's own modules
(tests/golden/glue/glue.npz: look_at.py, look.py, face_vertices.py run from /root/reference)."""
import os

import numpy as np

from oracle import project_ref as P

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'glue', 'glue.npz'))


def test_look_at_matches_reference_vectors():
    out = P.look_at(G['vertices'], G['eyes'])
    assert out.dtype == np.float32
    np.testing.assert_allclose(out, G['look_at'], rtol=0, atol=1e-6)
    out1 = P.look_at(G['vertices'], [0.0, 0.0, -2.732])
    np.testing.assert_allclose(out1, G['look_at_single_eye'], rtol=0, atol=1e-6)


def test_look_matches_reference_vectors():
    out = P.look(G['vertices'], G['eyes'], direction=[0.2, -0.1, 1.0], up=[0.0, 1.0, 0.0])
    np.testing.assert_allclose(out, G['look'], rtol=0, atol=1e-6)


def test_face_vertices_matches_reference_vectors():
    np.testing.assert_array_equal(P.face_vertices(G['vertices'], G['faces']), G['face_vertices'])
    # a shared [1,nf,3] index tensor addresses every batch item alike
    np.testing.assert_array_equal(P.face_vertices(G['vertices'], G['faces'][:1]), G['face_vertices'])


def test_perspective_and_orthogonal_expressions():
    cam = G['look_at']
    p = P.perspective(cam, 30.)
    w = np.float32(np.tan(np.float32(np.pi / 6)))
    np.testing.assert_array_equal(p[..., 2], cam[..., 2])
    np.testing.assert_array_equal(p[..., 0], cam[..., 0] / cam[..., 2] / w)
    o = P.orthogonal(cam, 0.5)
    np.testing.assert_array_equal(o[..., 1], cam[..., 1] * np.float32(0.5))


def test_f32_composition_close_to_f64():
    a = P.look_at_faces(G['vertices'], G['faces'], G['eyes'])
    b = P.look_at_faces(G['vertices'], G['faces'], G['eyes'], dtype=np.float64)
    assert np.abs(a - b).max() < 2e-6
