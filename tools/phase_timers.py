"""Wave-time per phase of the backward kernel from a -DGENDR_TIMERS=1 build of the library (diagnostic).
    cp gpurun_ablate_timers.so gendr_amd/libgendr_hip.so; python tools/phase_timers.py [--config c2] [--batch N]"""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import bench as B
import parity
from gendr_amd.functional import renderer as R
from gendr_amd.synthetic import benchmark_scene

ap = argparse.ArgumentParser()
ap.add_argument('--config', default='c2')
ap.add_argument('--batch', type=int, default=None)
args = ap.parse_args()
cfg = B.CONFIGS[args.config]
Bn = args.batch or min(cfg['batch'], 64)
isz = cfg['image_size']
opts = dict(cfg['opts']); opts.setdefault('double_side', False)
fv, tex = benchmark_scene(Bn, subdivisions=cfg['subdiv'], texture=cfg['texture'])
o, extra = parity.split_options(opts)
p = parity.hip_params(isz, o, extra)
faces = fv.reshape(Bn, -1, 9).cuda().contiguous(); t = tex.cuda().contiguous()
grad = torch.randn(Bn, 4, isz, isz, device='cuda')
NAMES = ['wave start-up', 'tile record + pixel inputs', 'entry list + emit', 'codes + first gather', 'pair math (+2nd gather)',
         'partials to LDS', 'segment sums + atomics', 'tile tail']
tot = [0] * 8
for it in range(5):
    rgba, aux, ws = R.native_forward(faces, t, p)
    off = ws.numel() - 24 * 1024 * 4 + (16 * 1024 + 64) * 4
    view = ws[off:off + 64].view(torch.int64)
    torch.cuda.synchronize()
    before = view.cpu().clone()
    R.native_backward(faces, t, rgba, aux, ws, grad, p)
    torch.cuda.synchronize()
    d = (view.cpu() - before).tolist()
    if it:                      # first iteration warms up
        tot = [a + b for a, b in zip(tot, d)]
s = float(sum(tot))
print('backward wave-time by phase (%s, batch %d): %.3g cycles summed over waves, 4 launches' % (args.config, Bn, s))
for n, v in zip(NAMES, tot):
    print('  %-30s %5.1f %%' % (n, 100.0 * v / s))
