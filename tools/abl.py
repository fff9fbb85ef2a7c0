import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench as B, parity
from tools.kbench import time_calls
from gendr_amd.synthetic import benchmark_scene
cfg = B.CONFIGS['c2']; Bn = 64; isz = 256
opts = dict(cfg['opts']); opts['double_side'] = False
fv, tex = benchmark_scene(Bn)
dev = 'cuda:0'
grad = torch.randn(Bn, 4, isz, isz, device=dev)
o, extra = parity.split_options(opts)
p = parity.hip_params(isz, o, extra)
for name, nf, shift in (('nf=1280', 1280, 0.0), ('nf=1280 offscreen', 1280, 10.0), ('nf=256 offscreen', 256, 10.0), ('nf=1 offscreen', 1, 10.0), ('nf=0', 0, 0.0)):
    f = fv[:, :nf].clone(); f[..., 0] += shift
    faces = f.reshape(Bn, nf, 9).to(dev).contiguous(); t = tex[:, :nf].to(dev).contiguous()
    fm, bm = time_calls(faces, t, p, grad, 10)
    print('%-22s fwd %8.3f ms  bwd %8.3f ms' % (name, fm, bm), flush=True)
