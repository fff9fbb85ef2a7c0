"""Kernel-level forward / backward time of one option set at one shape (native calls, HIP events):
    python tools/shapebench.py <image_size> <batch> key=value ...       e.g.  64 24 dist_func=logistic aggr_rgb_func=hard dist_eps=100"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import parity
from gendr_amd.synthetic import benchmark_scene
from tools.kbench import time_calls

isz, Bn = int(sys.argv[1]), int(sys.argv[2])
opts = dict(double_side=False)
for kv in sys.argv[3:]:
    k, v = kv.split('=')
    try:
        v = float(v) if ('.' in v or 'e' in v) else int(v)
    except ValueError:
        pass
    opts[k] = v
fv, tex = benchmark_scene(Bn)
o, extra = parity.split_options(opts)
p = parity.hip_params(isz, o, extra)
faces = fv.reshape(Bn, -1, 9).cuda().contiguous(); t = tex.cuda().contiguous()
grad = torch.randn(Bn, 4, isz, isz, device='cuda')
f, b = time_calls(faces, t, p, grad, 20)
print('%4d^2 x %3d %-70s fwd %8.3f ms  bwd %8.3f ms' % (isz, Bn, ' '.join(sys.argv[3:]), f, b), flush=True)
