"""Fused lighting of surface textures (HIP) against the numpy oracle and the PyTorch composition it replaces."""
import os

import numpy as np
import pytest
import torch

import gendr_amd as gendr
from gendr_amd import functional as Fn
from oracle import light_ref as L

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'glue', 'glue.npz'))
LIGHTS = [(0.6, (0.7, 1.0, 1.0), (0.3, 1.0, -0.2)), (0.3, (1.0, 0.2, 0.1), (-1.0, 0.1, 0.4))]


@pytest.mark.parametrize('T', [1, 4])
@pytest.mark.parametrize('shared_faces', [False, True])
def test_forward_bit_exact_against_oracle(T, shared_faces):
    tex = np.random.default_rng(T).random((3, 320, T, 3)).astype(np.float32)
    v, f = torch.from_numpy(G['vertices']).cuda(), torch.from_numpy(G['faces']).cuda()
    out = Fn.light_faces(v, f[:1] if shared_faces else f, torch.from_numpy(tex).cuda(), 0.4, (1.0, 0.9, 0.8), LIGHTS)
    np.testing.assert_array_equal(out.cpu().numpy(), L.light_faces(G['vertices'], G['faces'], tex, 0.4, (1.0, 0.9, 0.8), LIGHTS))


def test_module_routes_through_the_kernel_and_matches_the_tensor_chain(monkeypatch):
    tex = torch.rand(3, 320, 4, 3, device='cuda', generator=torch.Generator('cuda').manual_seed(0))
    g = torch.randn(3, 320, 4, 3, device='cuda', generator=torch.Generator('cuda').manual_seed(1))
    lighting = gendr.Lighting(0.4, [1.0, 0.9, 0.8], 0.6, [0.7, 1.0, 1.0], [0.3, 1.0, -0.2])
    res = []
    for fused in ('1', '0'):
        monkeypatch.setenv('GENDR_FUSED_LIGHTING', fused)
        v = torch.from_numpy(G['vertices']).cuda().requires_grad_(True)
        t = tex.clone().requires_grad_(True)
        mesh = gendr.Mesh(v, torch.from_numpy(G['faces']).cuda(), t, texture_type='surface')
        out = lighting(mesh).textures
        (out * g).sum().backward()
        res.append((out.detach().cpu().numpy(), v.grad.cpu().numpy(), t.grad.cpu().numpy()))
    (o1, gv1, gt1), (o0, gv0, gt0) = res
    np.testing.assert_allclose(o1, o0, rtol=3e-7, atol=1e-7)          # torch.sqrt / F.normalize are not bit-identical to IEEE sqrt
    np.testing.assert_allclose(gt1, gt0, rtol=3e-7, atol=1e-7)
    assert np.abs(gv1 - gv0).max() <= 1e-5 * max(1.0, np.abs(gv0).max())


def test_vertex_gradient_against_fp64_finite_differences():
    tex = np.random.default_rng(2).random((3, 320, 1, 3))
    g = np.random.default_rng(3).standard_normal((3, 320, 1, 3))
    v = torch.from_numpy(G['vertices']).cuda().requires_grad_(True)
    out = Fn.light_faces(v, torch.from_numpy(G['faces']).cuda(), torch.from_numpy(tex).float().cuda(), 0.4, (1.0, 0.9, 0.8), LIGHTS)
    (out * torch.from_numpy(g).float().cuda()).sum().backward()

    def loss(vert):
        fv = np.stack([vert[b][G['faces'][b]] for b in range(3)])
        raw = np.cross(fv[:, :, 2] - fv[:, :, 1], fv[:, :, 0] - fv[:, :, 1])
        n = raw / np.maximum(np.linalg.norm(raw, axis=-1, keepdims=True), 1e-6)
        light = 0.4 * np.array([1.0, 0.9, 0.8])[None, None]
        for inten, col, direc in LIGHTS:
            light = light + inten * np.array(col)[None, None] * np.maximum((n * np.array(direc)).sum(-1), 0)[..., None]
        return (tex * light[:, :, None, :] * g).sum()
    base = G['vertices'].astype(np.float64)
    for (b, n_, k) in [(0, 0, 0), (1, 40, 1), (2, 161, 2)]:
        p, m = base.copy(), base.copy()
        p[b, n_, k] += 1e-6
        m[b, n_, k] -= 1e-6
        fd = (loss(p) - loss(m)) / 2e-6
        assert abs(v.grad[b, n_, k].item() - fd) <= 1e-3 * max(1.0, abs(fd))


def test_fallbacks_and_errors(monkeypatch):
    v, f = torch.from_numpy(G['vertices']), torch.from_numpy(G['faces'])
    tex = torch.rand(3, 320, 1, 3)
    with pytest.raises(RuntimeError):
        Fn.light_faces(v, f, tex)                                           # CPU tensors: no fused path
    out = gendr.Lighting()(gendr.Mesh(v, f, tex, texture_type='surface'))   # ... but the module falls back to PyTorch
    assert out.textures.shape == tex.shape
    bad = f.clone().cuda()
    bad[0, 0, 0] = 999
    lit = Fn.light_faces(v.cuda(), bad, tex.cuda())
    assert torch.isnan(lit[0, 0]).all() and torch.isfinite(lit[0, 1:]).all()
