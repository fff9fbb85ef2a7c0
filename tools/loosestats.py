"""Faces flagged with a loose cull box and the live-pixel boxes loose_faces_kernel left, read back from the workspace.
    python tools/loosestats.py [--config c2] [--batch 64]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
import bench as B
import parity
from gendr_amd.functional import renderer as R
from gendr_amd.synthetic import benchmark_scene

ap = argparse.ArgumentParser()
ap.add_argument('--config', default='c2'); ap.add_argument('--batch', type=int, default=64)
args = ap.parse_args()
cfg = B.CONFIGS[args.config]
Bn, isz = args.batch, cfg['image_size']
opts = dict(cfg['opts']); opts.setdefault('double_side', False)
fv, tex = benchmark_scene(Bn, subdivisions=cfg['subdiv'], texture=cfg['texture'])
o, extra = parity.split_options(opts)
p = parity.hip_params(isz, o, extra)
faces = fv.reshape(Bn, -1, 9).cuda().contiguous(); t = tex.cuda().contiguous()
nf = faces.shape[1]
rgba, aux, ws = R.native_forward(faces, t, p)
torch.cuda.synchronize()
w = ws.cpu().numpy()
a256 = lambda v: (v + 255) // 256 * 256
control_off = len(w) - 24 * 1024 * 4
off = control_off - a256(Bn * nf * 4) - a256(Bn * nf * 16) - a256(Bn * 4)
flag = w[off:off + Bn * nf * 4].view(np.int32).reshape(Bn, nf); off += a256(Bn * nf * 4)
box = w[off:off + Bn * nf * 16].view(np.int32).reshape(Bn, nf, 4)
print('%s batch %d: %d faces flagged in %d images' % (args.config, Bn, int((flag != 0).sum()), int(((flag != 0).sum(1) > 0).sum())))
for b in range(Bn):
    for f in np.nonzero(flag[b])[0]:
        print('  image %2d face %4d: columns %s rows %s' % (b, f, tuple(box[b, f, :2]), tuple(box[b, f, 2:])))
