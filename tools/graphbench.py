"""C2 step (forward + backward through the autograd Function) eager vs replayed from a HIP graph."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gendr_amd.functional import render
from gendr_amd.synthetic import benchmark_scene

cfg = bench.CONFIGS['c2']
B = cfg['batch']
opts = dict(cfg['opts'], image_size=cfg['image_size'], double_side=False)
fv, tex = benchmark_scene(B, subdivisions=cfg['subdiv'], texture=cfg['texture'], seed=0)
fv, tex = fv.cuda().requires_grad_(True), tex.cuda().requires_grad_(True)
g = torch.randn(B, 4, cfg['image_size'], cfg['image_size'], device='cuda')


def step():
    img = render(fv, tex, **opts)
    return torch.autograd.grad(img, (fv, tex), g)


def timeit(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


eager = timeit(step)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    out = step()
replay = timeit(graph.replay)
print('eager %.4f ms/step (%.0f frames/s)   graph replay %.4f ms/step (%.0f frames/s)' % (eager, B / eager * 1e3, replay, B / replay * 1e3))
