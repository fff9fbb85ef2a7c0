"""How many of the listed (pixel, face) pairs the backward kernel has to evaluate: the forward kernel's pair hints (PairHints: two bits per pair,
3 = no gradient) read back from the workspace -- live pairs per batch, batches a compaction of the live pairs would leave.
    python tools/hintstats.py [--config c2] [--batch 64]"""
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
p = parity.hip_params(isz, o, dict(extra, pair_hints=1))
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
info_off = a256(Bn * nf * 4 * 4) + a256(Bn * nf * rec * 4) + a256(tiles * chunks * 8) + a256(tiles * 4)
ent_off = info_off + a256(tiles * 16)
control_off = len(w) - 24 * 1024 * 4
sorted_off = control_off - a256(tiles * 16)
H = (sorted_off - ent_off) // 2
hints = w[ent_off + H:ent_off + 2 * H].view(np.uint64).reshape(-1, 2)
info = w[info_off:info_off + tiles * 16].view(np.int32).reshape(tiles, 4)
control = w[control_off:].view(np.int32)
tot = live = nbat = dead_bat = comp_bat = 0
hist = np.zeros(65, np.int64)
for x in range(8):
    n = int(control[x * 1024]); qb = x * tiles // 8
    for tile, first, cnt, pairs in info[qb:qb + n]:
        if first < 0 or cnt <= 0 or pairs <= 0:
            continue
        pairs = int(pairs); first = int(first); nb = (pairs + 63) // 64
        tl = 0
        for k in range(nb):
            lo, hi = int(hints[first + k, 0]), int(hints[first + k, 1])
            valid = int(min(64, int(pairs) - 64 * k))
            m = (1 << valid) - 1
            l = bin(~(lo & hi) & m).count('1')
            hist[l] += 1; tl += l
            dead_bat += l == 0
        tot += pairs; live += tl; nbat += nb; comp_bat += (tl + 63) // 64
print('%s batch %d: %d pairs in %d batches (%.1f per batch); %d live = %.1f %%; batches without a live pair %d (skipped today); '
      'a compaction of the live pairs per tile would leave %d batches (%.1f %% of the %d evaluated today)'
      % (args.config, Bn, tot, nbat, tot / max(nbat, 1), live, 100.0 * live / max(tot, 1), dead_bat, comp_bat, 100.0 * comp_bat / max(nbat - dead_bat, 1), nbat - dead_bat))
print('live pairs per batch, deciles:', [int(np.searchsorted(np.cumsum(hist), q * hist.sum() / 10)) for q in range(1, 10)])
