"""Eager step of the headline workload through the two host paths, in one process on one box (boxes differ by several percent):
    python tools/eagerbench.py [--config c2] [--batch N]
  cpp     gendr_amd.functional.render -> the C++ autograd node (csrc/gendr_torch.cpp)
  python  the same call with the node switched off -> GenDRFunction (ctypes)
  graph   the step captured once and replayed: what the GPU alone needs
and the host time of a step with the GPU out of the way (launch-only: the stream is not synchronised per step, the kernels of a
1-frame 32^2 workload are negligible)."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench as B
from gendr_amd.functional import renderer as R
from gendr_amd.synthetic import benchmark_scene


def timed(fn, n):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='c2'); ap.add_argument('--batch', type=int, default=None); ap.add_argument('--steps', type=int, default=50)
    a = ap.parse_args()
    cfg = B.CONFIGS[a.config]
    Bn = a.batch or cfg['batch']
    isz = cfg['image_size']
    opts = dict(cfg['opts']); opts.setdefault('double_side', False)
    fv, tex = benchmark_scene(Bn, subdivisions=cfg['subdiv'], texture=cfg['texture'], device='cuda')
    fv.requires_grad_(True); tex.requires_grad_(True)
    grad = torch.randn(Bn, 4, isz, isz, device='cuda')

    def step(f=fv, t=tex, g=grad, s=isz):
        f.grad = None; t.grad = None
        R.render(f, t, image_size=s, **opts).backward(g)

    fs, ts = benchmark_scene(1, subdivisions=1, texture=cfg['texture'], device='cuda')
    fs.requires_grad_(True); ts.requires_grad_(True)
    gs = torch.randn(1, 4, 32, 32, device='cuda')
    res = {}
    for name, on in (('cpp', True), ('python', False), ('cpp', True), ('python', False)):
        R._CPP_AUTOGRAD = on
        for _ in range(10):
            step()
        res.setdefault(name, []).append(min(timed(step, a.steps), timed(step, a.steps)))
        small = lambda: step(fs, ts, gs, 32)
        for _ in range(10):
            small()
        res.setdefault(name + '_host_only', []).append(min(timed(small, 200), timed(small, 200)))
    R._CPP_AUTOGRAD = True
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    fv.grad = None; tex.grad = None
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode='thread_local'):
        step()
    g.replay()
    res['graph'] = [min(timed(g.replay, a.steps), timed(g.replay, a.steps))]
    for k, v in res.items():
        print('%-18s %s ms per step' % (k, ' '.join('%.4f' % x for x in v)))
    print('eager cpp / graph = %.3f, eager python / graph = %.3f' % (min(res['cpp']) / res['graph'][0], min(res['python']) / res['graph'][0]))


if __name__ == '__main__':
    main()
