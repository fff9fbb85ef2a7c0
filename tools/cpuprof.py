"""Host-side cost of one fwd+bwd step through the autograd Function (cProfile, C2 shapes)."""
import sys, os, cProfile, pstats, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

from gendr_amd.functional import render
from gendr_amd.synthetic import benchmark_scene

cfg = bench.CONFIGS['c2']
B = cfg['batch']
opts = dict(cfg['opts'], image_size=cfg['image_size'], double_side=False)
fv, tex = benchmark_scene(B, subdivisions=cfg['subdiv'], texture=cfg['texture'], seed=0)
fv, tex = fv.cuda(), tex.cuda()
g = torch.randn(B, 4, cfg['image_size'], cfg['image_size'], device='cuda')


def step():
    a = fv.detach().requires_grad_(True)
    t = tex.detach().requires_grad_(True)
    out = render(a, t, **opts)
    out.backward(g)


for _ in range(20):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('enqueue %.1f us/step, drained after %.1f us/step' % ((t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
