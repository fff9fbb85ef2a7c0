"""Queue lengths, grades of the graded sub-tile split (order_tiles_kernel, TileWalk) and pair weights per listed tile, read back
from the workspace.
    python tools/gradestats.py [--config c2] [--batch 8] [--size 64] [key=value ...]"""
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
ap.add_argument('--config', default='c2'); ap.add_argument('--batch', type=int, default=8); ap.add_argument('--size', type=int, default=0)
ap.add_argument('opts', nargs='*')
args = ap.parse_args()
cfg = B.CONFIGS[args.config]
Bn, isz = args.batch, args.size or cfg['image_size']
opts = dict(cfg['opts']); opts.setdefault('double_side', False)
for kv in args.opts:
    k, v = kv.split('=')
    try:
        v = float(v) if ('.' in v or 'e' in v) else int(v)
    except ValueError:
        pass
    opts[k] = v
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
off = a256(Bn * nf * 4 * 4) + a256(Bn * nf * rec * 4) + a256(tiles * chunks * 8) + a256(tiles * 4)
info = w[off:off + tiles * 16].view(np.int32).reshape(tiles, 4)
control = w[len(w) - 24 * 1024 * 4:].view(np.int32)
for x in range(8):
    n = int(control[x * 1024]); qb = x * tiles // 8
    pw = info[qb:qb + n, 3]
    live, g8, g4, g2 = (int(v) for v in control[x * 1024 + 4:x * 1024 + 8])
    print('queue %d: %4d listed, %4d live, hint flag %d, split 8/4/2-fold: %d %d %d -> %d work items | pairs per tile max %d, p90 %d, p50 %d, entries p50 %d'
          % (x, n, live, control[x * 1024 + 1], g8, g4, g2, live + 7 * g8 + 3 * g4 + g2,
             int(pw.max()) if n else 0, int(np.percentile(pw, 90)) if n else 0, int(np.median(pw)) if n else 0, int(np.median(info[qb:qb + n, 2])) if n else 0))

# region tags of the coverage entries (CoverEnt.npix bits 8..9) and how full the entries are
ent_off = off + a256(tiles * 16)
ents = w[ent_off:].view(np.int32)
tags = np.zeros(4, np.int64); px = np.zeros(4, np.int64); full = 0; n_ent = 0
for x in range(8):
    n = int(control[x * 1024]); qb = x * tiles // 8
    for tile, first, cnt, pairs in info[qb:qb + n]:
        if first < 0 or cnt <= 0:
            continue
        e = ents[first * 4:(first + cnt) * 4].reshape(cnt, 4)
        t = (e[:, 1] >> 8) & 3; p = e[:, 1] & 255
        for k in range(4):
            tags[k] += int((t == k).sum()); px[k] += int(p[t == k].sum())
        full += int((p == 64).sum()); n_ent += cnt
print('entries %d, with all 64 pixels %.1f %%, pixels per entry %.1f; region tag none / edge 0 / 1 / 2: %s of the entries, %s of the pairs'
      % (n_ent, 100.0 * full / max(n_ent, 1), px.sum() / max(n_ent, 1), np.round(100.0 * tags / max(tags.sum(), 1), 1), np.round(100.0 * px / max(px.sum(), 1), 1)))
