"""Fused camera transform + face gather (HIP) against the numpy oracle and against the unfused
PyTorch composition whose stages are pinned to the reference's modules by tests/golden/glue."""
import os

import numpy as np
import pytest
import torch

import gendr_amd as gendr
from gendr_amd import functional as Fn
from gendr_amd.synthetic import icosphere
from oracle import project_ref as P

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'glue', 'glue.npz'))
TOL = 2e-6          # a few fp32 ulps of O(1) coordinates: rocBLAS' matmul summation order is unspecified


def _inputs(shared_faces=False):
    v = torch.from_numpy(G['vertices']).cuda()
    f = torch.from_numpy(G['faces']).cuda()
    e = torch.from_numpy(G['eyes']).cuda()
    return v, (f[:1] if shared_faces else f), e


@pytest.mark.parametrize('perspective', [True, False])
@pytest.mark.parametrize('shared_faces', [False, True])
def test_forward_vs_oracle_and_unfused(perspective, shared_faces):
    v, f, e = _inputs(shared_faces)
    out = Fn.look_at_faces(v, f, e, perspective=perspective, viewing_angle=25., viewing_scale=0.7)
    ref = P.look_at_faces(G['vertices'], G['faces'], G['eyes'], perspective_=perspective, viewing_angle=25., viewing_scale=0.7)
    assert out.shape == (3, 320, 3, 3)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=0, atol=TOL)
    cam = Fn.look_at(v, e)
    cam = Fn.perspective(cam, 25.) if perspective else Fn.orthogonal(cam, 0.7)
    unf = Fn.face_vertices(cam, f.expand(3, -1, -1))
    np.testing.assert_allclose(out.cpu().numpy(), unf.cpu().numpy(), rtol=0, atol=TOL)


def test_reference_vectors_through_fused_kernel():
    # orthogonal, scale 1 == face_vertices(look_at(...)): both stages are reference outputs in glue.npz
    v, f, e = _inputs()
    out = Fn.look_at_faces(v, f, e, perspective=False, viewing_scale=1.0).cpu().numpy()
    np.testing.assert_allclose(out, P.face_vertices(G['look_at'], G['faces']), rtol=0, atol=TOL)
    out = Fn.look_faces(v, f, e, direction=[0.2, -0.1, 1.0], perspective=False).cpu().numpy()
    np.testing.assert_allclose(out, P.face_vertices(G['look'], G['faces']), rtol=0, atol=TOL)


@pytest.mark.parametrize('perspective', [True, False])
def test_backward_vs_autograd_of_unfused(perspective):
    v, f, e = _inputs()
    g = torch.randn(3, 320, 3, 3, device='cuda', generator=torch.Generator('cuda').manual_seed(3))
    grads = []
    for fused in (True, False):
        vv, ee = v.clone().requires_grad_(True), e.clone().requires_grad_(True)
        if fused:
            out = Fn.look_at_faces(vv, f, ee, perspective=perspective)
        else:
            cam = Fn.look_at(vv, ee)
            out = Fn.face_vertices(Fn.perspective(cam) if perspective else Fn.orthogonal(cam), f)
        (out * g).sum().backward()
        grads.append((vv.grad.cpu().double(), ee.grad.cpu().double()))
    (gv_a, ge_a), (gv_b, ge_b) = grads
    assert (gv_a - gv_b).abs().max() <= 1e-5 * max(1.0, gv_b.abs().max().item())
    # the eye gradient sums ~1000 terms of mixed sign, also through the rotation
    assert (ge_a - ge_b).abs().max() <= 1e-4 * max(1.0, ge_b.abs().max().item())


def test_camera_parameter_gradients_vs_autograd():
    # eye, at and up all learnable; look (direction) as well
    v, f, e = _inputs()
    gen = torch.Generator('cuda').manual_seed(11)
    g = torch.randn(3, 320, 3, 3, device='cuda', generator=gen)
    at0 = 0.2 * torch.randn(3, 3, device='cuda', generator=gen)
    up0 = torch.tensor([[0.1, 1.0, 0.0], [0.0, 1.0, 0.2], [-0.3, 0.9, 0.1]], device='cuda')
    for mode in ('look_at', 'look'):
        res = []
        for fused in (True, False):
            ee, aa, uu = (t.clone().requires_grad_(True) for t in (e, at0, up0))
            tgt = aa if mode == 'look_at' else aa - e                     # a direction roughly towards the mesh
            if fused:
                fn = Fn.look_at_faces if mode == 'look_at' else Fn.look_faces
                out = fn(v, f, ee, tgt, uu)
            else:
                cam = Fn.look_at(v, ee, tgt, uu) if mode == 'look_at' else Fn.look(v, ee, tgt, uu)
                out = Fn.face_vertices(Fn.perspective(cam), f)
            (out * g).sum().backward()
            res.append([t.grad.cpu().double() for t in (ee, aa, uu)] + [out.detach().cpu()])
        np.testing.assert_allclose(res[0][3].numpy(), res[1][3].numpy(), rtol=0, atol=TOL)
        for ga, gb, name in zip(res[0][:3], res[1][:3], ('eye', 'target', 'up')):
            assert (ga - gb).abs().max() <= 2e-4 * max(1.0, gb.abs().max().item()), (mode, name, ga, gb)


def test_project_faces_with_explicit_rotation():
    from gendr_amd.functional.geometry import _camera_rotation
    v, f, e = _inputs()
    rot = _camera_rotation(-e, torch.tensor([[0., 1., 0.]], device='cuda').expand(3, 3))
    a = Fn.project_faces(v, f, rot, e)
    b = Fn.look_at_faces(v, f, e)
    np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=0, atol=TOL)


def test_degenerate_up_uses_normalize_eps():
    # up parallel to the viewing direction: x_raw = 0 -> F.normalize clamps the norm at 1e-5 (look_at.py:54)
    v, f, _ = _inputs()
    e = torch.tensor([[0., 3., 0.]], device='cuda').expand(3, 3).contiguous()
    a = Fn.look_at_faces(v, f, e, perspective=False)
    b = Fn.face_vertices(Fn.orthogonal(Fn.look_at(v, e)), f)
    np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=0, atol=TOL)


def test_backward_vs_fp64_finite_differences():
    v, f, e = _inputs()
    g = np.random.default_rng(0).standard_normal((3, 320, 3, 3))
    vv = v.clone().requires_grad_(True)
    (Fn.look_at_faces(vv, f, e) * torch.from_numpy(g).float().cuda()).sum().backward()
    base = G['vertices'].astype(np.float64)
    loss = lambda x: (P.look_at_faces(x, G['faces'], G['eyes'], dtype=np.float64) * g).sum()
    h = 1e-6
    for (b, n, k) in [(0, 0, 0), (1, 17, 1), (2, 161, 2), (0, 80, 2)]:
        xp, xm = base.copy(), base.copy()
        xp[b, n, k] += h
        xm[b, n, k] -= h
        fd = (loss(xp) - loss(xm)) / (2 * h)
        assert abs(vv.grad[b, n, k].item() - fd) <= 1e-4 * max(1.0, abs(fd))


def test_camera_module_routes_through_fused_kernel_and_matches_unfused(monkeypatch):
    v0, f0 = icosphere(2)
    mesh = gendr.Mesh(np.tile(v0[None], (2, 1, 1)), np.tile(f0[None], (2, 1, 1)))
    cam = gendr.LookAt(viewing_angle=20)
    cam.set_eyes_from_angles(torch.tensor([2.7, 3.0]), torch.tensor([27.9, 11.3]), torch.tensor([41.3, -21.7]))
    fused_mesh = cam(mesh)
    assert type(fused_mesh).__name__ == '_ProjectedMesh'
    fv = fused_mesh.face_vertices
    monkeypatch.setenv('GENDR_FUSED_PROJECTION', '0')
    plain = cam(mesh)
    assert type(plain) is gendr.Mesh
    np.testing.assert_allclose(fv.cpu().numpy(), plain.face_vertices.cpu().numpy(), rtol=0, atol=TOL)
    np.testing.assert_allclose(fused_mesh.vertices.cpu().numpy(), plain.vertices.cpu().numpy(), rtol=0, atol=0)
    renderer = gendr.GenDR(image_size=64, dist_func='logistic', dist_scale=1e-2)
    a, b = renderer(fused_mesh), renderer(plain)
    # 1-ulp differences in the projected vertices are amplified without bound by faces seen edge-on
    # (|det| ~ 1e-7 at a sphere's silhouette), so compare the images in bulk, not by their worst pixel
    d = (a - b).abs()
    assert d.mean().item() < 1e-5 and (d > 1e-4).float().mean().item() < 5e-3


def test_errors():
    v, f, e = _inputs()
    with pytest.raises(RuntimeError):
        Fn.look_at_faces(v.cpu(), f.cpu(), e.cpu())
    bad = f.clone()
    bad[0, 0, 0] = 10 ** 6
    with pytest.raises(IndexError):
        Fn.look_at_faces(v, bad, e)
    out = Fn.look_at_faces(v, f[:, :0], e)
    assert out.shape == (3, 0, 3, 3)


def test_out_of_range_index_is_memory_safe_even_when_the_host_check_is_skipped(monkeypatch):
    from gendr_amd.functional import projection as PJ
    v, f, e = _inputs()
    bad = f.clone()
    bad[1, 5, 2] = 162                      # == nv: one past the end
    with pytest.raises(IndexError):
        Fn.look_at_faces(v, bad, e)
    monkeypatch.setattr(PJ, '_check_indices', lambda faces, nv: None)      # as during HIP-graph capture
    vv = v.clone().requires_grad_(True)
    out = Fn.look_at_faces(vv, bad, e)
    assert torch.isnan(out[1, 5, 2]).all() and torch.isfinite(out[0]).all() and torch.isfinite(out[1, 5, :2]).all()
    torch.nan_to_num(out).sum().backward()
    assert torch.isfinite(vv.grad).all()


def test_broadcastable_eye_at_up_like_the_reference():
    """A [1,3] eye / at / up broadcasts over the batch in the reference's look_at and in the unfused glue; the fused
    path must accept it too (it raised ValueError)."""
    v, f, _ = _inputs()
    eye1 = torch.tensor([[0.3, 0.4, -2.7]], device='cuda')
    up1 = torch.tensor([[0.0, 1.0, 0.0]], device='cuda')
    out = Fn.look_at_faces(v, f, eye1, at=torch.zeros(1, 3, device='cuda'), up=up1, viewing_angle=20.)
    unf = Fn.face_vertices(Fn.perspective(Fn.look_at(v, eye1, at=torch.zeros(1, 3, device='cuda'), up=up1), 20.), f)
    np.testing.assert_allclose(out.cpu().numpy(), unf.cpu().numpy(), rtol=0, atol=TOL)
    cam = gendr.LookAt(viewing_angle=20.)
    cam.set_eyes(eye1)
    m = gendr.Mesh(v, f)
    np.testing.assert_allclose(cam(m).face_vertices.cpu().numpy(), unf.cpu().numpy(), rtol=0, atol=TOL)
    with pytest.raises(ValueError):
        Fn.look_at_faces(v, f, torch.zeros(2, 3, device='cuda'))        # genuinely incompatible: B = 3
