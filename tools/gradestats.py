"""Queue lengths, grade counts of the graded split (walk_plan) and pair weights per listed tile, read back from the workspace.
    python tools/gradestats.py [--config c2] [--batch 8]"""
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
ap.add_argument('--config', default='c2'); ap.add_argument('--batch', type=int, default=8)
args = ap.parse_args()
cfg = B.CONFIGS[args.config]
Bn, isz = args.batch, cfg['image_size']
opts = dict(cfg['opts']); opts.setdefault('double_side', False)
fv, tex = benchmark_scene(Bn, subdivisions=cfg['subdiv'], texture=cfg['texture'])
o, extra = parity.split_options(opts)
p = parity.hip_params(isz, o, extra)
faces = fv.reshape(Bn, -1, 9).cuda().contiguous(); t = tex.cuda().contiguous()
nf, T = faces.shape[1], t.shape[2]
rgba, aux, ws = R.native_forward(faces, t, p)
torch.cuda.synchronize()
w = ws.cpu().numpy()
a256 = lambda v: (v + 255) // 256 * 256
tiles_x = (isz + 7) // 8
tiles = Bn * tiles_x * tiles_x
chunks = (nf + 63) // 64
rec = 60 if cfg['texture'] == 'vertex' else {1: 56}.get(T, 48)
off = a256(Bn * nf * 16 * 4) + a256(Bn * nf * rec * 4) + a256(tiles * chunks * 8) + a256(tiles * 4)
info = w[off:off + tiles * 16].view(np.int32).reshape(tiles, 4)
control = w[len(w) - 24 * 1024 * 4:].view(np.int32)
for x in range(8):
    n = int(control[x * 1024]); qb = x * tiles // 8
    pw = info[qb:qb + n, 3]
    print('queue %d: %4d listed, flag %d, grades >=512/256/128: %s | from the records: %d %d %d | pairs max %d, p50 %d'
          % (x, n, control[x * 1024 + 1], list(control[x * 1024 + 2:x * 1024 + 5]), int((pw >= 512).sum()), int((pw >= 256).sum()), int((pw >= 128).sum()),
             int(pw.max()) if n else 0, int(np.median(pw)) if n else 0))
