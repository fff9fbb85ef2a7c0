"""Plain-PyTorch glue (camera, mesh, lighting, losses) against golden vectors generated from the REFERENCE's
own pure-Python modules (tests/golden/make_glue_golden.py imports them from /root/reference by file path)."""
import os

import numpy as np
import pytest
import torch

Z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'glue', 'glue.npz'))
V = torch.from_numpy(Z['vertices'])
Fc = torch.from_numpy(Z['faces'])


def close(a, b, tol=1e-6):
    return np.allclose(a.detach().numpy() if torch.is_tensor(a) else a, b, rtol=tol, atol=tol)


def test_look_at_and_look():
    from gendr_amd.functional import look_at, look
    assert close(look_at(V, torch.from_numpy(Z['eyes'])), Z['look_at'])
    assert close(look_at(V, [0.0, 0.0, -2.732]), Z['look_at_single_eye'])
    assert close(look(V, torch.from_numpy(Z['eyes']), direction=[0.2, -0.1, 1.0], up=torch.tensor([0.0, 1.0, 0.0])), Z['look'])


def test_points_from_angles():
    from gendr_amd.functional import get_points_from_angles
    d, e, a = (torch.from_numpy(x) for x in Z['angles'])
    assert close(get_points_from_angles(d, e, a), Z['points_from_angles'])
    assert close(np.asarray(get_points_from_angles(2.732, 30.0, -15.0)), Z['points_from_angles_scalar'], 1e-12)


def test_face_vertices_and_normals():
    from gendr_amd.functional import face_vertices, vertex_normals
    assert np.array_equal(face_vertices(V, Fc).numpy(), Z['face_vertices'])
    assert close(vertex_normals(V, Fc), Z['vertex_normals'])


def test_lighting_functions_and_module():
    from gendr_amd.functional import ambient_lighting, directional_lighting, vertex_normals
    light = torch.zeros(V.shape[0], V.shape[1], 3)
    light = ambient_lighting(light, 0.4, (1.0, 0.9, 0.8))
    assert close(light, Z['ambient'])
    light = directional_lighting(light, vertex_normals(V, Fc), 0.6, (0.7, 1.0, 1.0), (0.3, 1.0, -0.2))
    assert close(light, Z['directional'])
    import gendr_amd
    m = gendr_amd.Mesh(V, Fc, texture_type='vertex')
    lit = gendr_amd.Lighting(0.4, [1.0, 0.9, 0.8], 0.6, [0.7, 1.0, 1.0], [0.3, 1.0, -0.2])(m)
    assert close(lit.textures, Z['directional'])           # textures start as ones
    assert lit.face_textures.shape == (V.shape[0], Fc.shape[1], 3, 3)


def test_losses():
    import gendr_amd
    lap = gendr_amd.LaplacianLoss(V[0], Fc[0].long())
    assert close(lap(V), Z['laplacian_loss'], 1e-5)
    assert close(gendr_amd.LaplacianLoss(V[0], Fc[0].long(), average=True)(V), Z['laplacian_loss_avg'], 1e-5)
    assert close(gendr_amd.FlattenLoss(Fc[0].long())(V), Z['flatten_loss'], 1e-4)
    # rotated index order: the reference drops edges that are (2,0) in both faces; same edge count, same loss
    fr = torch.from_numpy(Z['faces_rotated'])
    fl = gendr_amd.FlattenLoss(fr)
    assert fl.v0s.shape[0] == int(Z['flatten_edges_rotated']) < 3 * fr.shape[0] // 2
    assert close(fl(V), Z['flatten_loss_rotated'], 1e-4)


def test_mesh_transform_chain_and_alias_package(tmp_path):
    import gendr
    import gendr.cuda.generalized_renderer as native
    assert gendr.GenDR is __import__('gendr_amd').GenDR and callable(native.forward_render)
    m = gendr.Mesh(V.numpy()[0], Fc.numpy()[0]) if not torch.cuda.is_available() else gendr.Mesh(V[:1], Fc[:1])
    assert m.batch_size == 1 and m.num_faces == Fc.shape[1] and m.texture_res == 1
    cam = gendr.LookAt(viewing_angle=15)
    cam.set_eyes_from_angles(torch.tensor([2.732]), torch.tensor([30.0]), torch.tensor([-15.0]))
    out = cam(gendr.Lighting()(m))
    fv = out.face_vertices
    assert fv.shape == (1, Fc.shape[1], 3, 3) and float(fv[..., :2].abs().max()) < 1.0 and float(fv[..., 2].min()) > 1.0
    # OBJ round trip (geometry + vertex colours)
    path = str(tmp_path / 'mesh.obj')
    mv = gendr.Mesh(V[:1], Fc[:1], textures=torch.rand(1, V.shape[1], 3), texture_type='vertex')
    mv.save_obj(path, save_texture=True)
    back = gendr.Mesh.from_obj(path, load_texture=True, texture_type='vertex')
    assert np.allclose(back.vertices.cpu().numpy(), V[:1].numpy(), atol=1e-6)
    assert np.array_equal(back.faces.cpu().numpy(), Fc[:1].numpy())
    assert np.allclose(back.textures.cpu().numpy(), mv.textures.numpy(), atol=1e-6)
    with pytest.raises(TypeError):                 # HIP only, like the renderer: no CPU path
        gendr.functional.voxelization(fv, 32)


def test_projection_transform():
    import gendr_amd
    P = torch.tensor([[[500., 0., 256., 0.], [0., 500., 256., 0.], [0., 0., 1., 0.]]])
    pts = torch.tensor([[[0.0, 0.0, 2.0], [0.2, -0.1, 4.0]]])
    out = gendr_amd.Projection(P, orig_size=512).transform(pts)
    x = (500 * 0.2 / 4.0 + 256)
    assert torch.allclose(out[0, 0], torch.tensor([0.0, 0.0, 2.0]), atol=1e-4)
    assert abs(float(out[0, 1, 0]) - 2 * (x - 256) / 512) < 1e-4
    with pytest.raises(ValueError):
        gendr_amd.Projection(torch.zeros(3, 4))
