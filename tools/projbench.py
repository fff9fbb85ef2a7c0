"""Fused projection (row f-1) vs the unfused PyTorch composition, C2-shaped input: B=64, icosphere(3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gendr_amd import functional as Fn
from gendr_amd.synthetic import icosphere

v0, f0 = icosphere(3)
B = 64
v = torch.from_numpy(v0).cuda()[None].repeat(B, 1, 1).contiguous()
f = torch.from_numpy(f0).cuda().int()[None].repeat(B, 1, 1).contiguous()
eye = Fn.get_points_from_angles(torch.full((B,), 2.732).cuda(), torch.linspace(0, 40, B).cuda(), torch.linspace(-90, 90, B).cuda())
g = torch.randn(B, f.shape[1], 3, 3, device='cuda')


def fused():
    vv, ee = v.clone().requires_grad_(True), eye.clone().requires_grad_(True)
    (Fn.look_at_faces(vv, f, ee)).backward(g)


def unfused():
    vv, ee = v.clone().requires_grad_(True), eye.clone().requires_grad_(True)
    Fn.face_vertices(Fn.perspective(Fn.look_at(vv, ee)), f).backward(g)


for name, fn in (('fused', fused), ('unfused', unfused)):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(200):
        fn()
    e.record()
    torch.cuda.synchronize()
    print('%-8s fwd+bwd %.1f us per call (B=%d, nv=%d, nf=%d)' % (name, s.elapsed_time(e) * 1000 / 200, B, v.shape[1], f.shape[1]))
