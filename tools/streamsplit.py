"""Experiment: the C2 batch as two halves on two HIP streams (fork / join by events) against one call.
    python tools/streamsplit.py [--config c2] [--iters 30] [--parts 2]
"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench as B
from gendr_amd.functional import renderer as R
from gendr_amd.synthetic import benchmark_scene
import parity


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='c2'); ap.add_argument('--iters', type=int, default=30); ap.add_argument('--parts', type=int, default=2)
    ap.add_argument('--batch', type=int, default=None); ap.add_argument('--graph', action='store_true', help='time the replay of a captured HIP graph (no host launch gaps)')
    args = ap.parse_args()
    cfg = B.CONFIGS[args.config]
    Bn = args.batch or cfg['batch']; isz = cfg['image_size']
    opts = dict(cfg['opts']); opts.setdefault('double_side', False)
    fv, tex = benchmark_scene(Bn, subdivisions=cfg['subdiv'], texture=cfg['texture'])
    dev = 'cuda:0'
    o, extra = parity.split_options(opts)
    p = parity.hip_params(isz, o, extra)
    faces = fv.reshape(Bn, -1, 9).to(dev).contiguous(); t = tex.to(dev).contiguous()
    grad = torch.randn(Bn, 4, isz, isz, device=dev)

    def whole():
        rgba, aux, rec = R.native_forward(faces, t, p)
        return R.native_backward(faces, t, rgba, aux, rec, grad, p)

    n = args.parts
    cuts = [Bn * i // n for i in range(n + 1)]
    streams = [torch.cuda.Stream() for _ in range(n - 1)]
    parts = [(faces[cuts[i]:cuts[i + 1]].contiguous(), t[cuts[i]:cuts[i + 1]].contiguous(), grad[cuts[i]:cuts[i + 1]].contiguous()) for i in range(n)]

    def split():
        main_s = torch.cuda.current_stream()
        fork = torch.cuda.Event(); fork.record(main_s)
        outs = []
        for i in range(n):
            s = main_s if i == 0 else streams[i - 1]
            if i:
                s.wait_event(fork)
            with torch.cuda.stream(s):
                f, tt, g = parts[i]
                rgba, aux, rec = R.native_forward(f, tt, p)
                outs.append(R.native_backward(f, tt, rgba, aux, rec, g, p))
                if i:
                    e = torch.cuda.Event(); e.record(s); main_s.wait_event(e)
        return outs

    def split_phased():
        # forward of all parts, join, backward of all parts (what an autograd Function with an internal split would do)
        main_s = torch.cuda.current_stream()
        fork = torch.cuda.Event(); fork.record(main_s)
        fw = []
        for i in range(n):
            s = main_s if i == 0 else streams[i - 1]
            if i:
                s.wait_event(fork)
            with torch.cuda.stream(s):
                f, tt, g = parts[i]
                fw.append(R.native_forward(f, tt, p))
                if i:
                    e = torch.cuda.Event(); e.record(s); main_s.wait_event(e)
        fork2 = torch.cuda.Event(); fork2.record(main_s)
        outs = []
        for i in range(n):
            s = main_s if i == 0 else streams[i - 1]
            if i:
                s.wait_event(fork2)
            with torch.cuda.stream(s):
                f, tt, g = parts[i]
                outs.append(R.native_backward(f, tt, fw[i][0], fw[i][1], fw[i][2], g, p))
                if i:
                    e = torch.cuda.Event(); e.record(s); main_s.wait_event(e)
        return outs

    for name, fn in (('one call', whole), ('%d streams' % n, split), ('%d streams, joined between passes' % n, split_phased), ('one call', whole)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        if args.graph:
            cs = torch.cuda.Stream()
            cs.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(cs):
                fn()
            torch.cuda.current_stream().wait_stream(cs)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                keep = fn()
            fn = g.replay
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(args.iters):
            fn()
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / args.iters
        print('%-40s %8.3f ms per step -> %.0f frames/s' % (name, ms, Bn / (ms * 1e-3)), flush=True)


if __name__ == '__main__':
    main()
