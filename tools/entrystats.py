"""Tile / entry / pair statistics of a config from the workspace the forward call leaves (queue records + entry pool).
    python tools/entrystats.py [--config c2] [--batch N]"""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
import bench as B
import parity
from gendr_amd import _native
from gendr_amd.functional import renderer as R
from gendr_amd.synthetic import benchmark_scene

ap = argparse.ArgumentParser()
ap.add_argument('--config', default='c2')
ap.add_argument('--batch', type=int, default=4)
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
rec = {1: 56, 3: 60}.get(T, 48) if cfg['texture'] != 'vertex' else 60
off = a256(Bn * nf * 4 * 4) + a256(Bn * nf * rec * 4)
masks = w[off:off + tiles * chunks * 8].view(np.uint64); off += a256(tiles * chunks * 8)
off += a256(tiles * 4)
info = w[off:off + tiles * 16].view(np.int32).reshape(tiles, 4); off += a256(tiles * 16)
control_off = len(w) - 24 * 1024 * 4                       # the entry pool fills the space up to the control block
cap = (control_off - off) // 16
ent = w[off:off + cap * 16].view(np.uint32).reshape(cap, 4)
control = w[control_off:].view(np.int32)
nlisted = [int(control[x * 1024]) for x in range(8)]
listings = 0                                              # mask rows of listed tiles only: the others are never written
list_per_tile = []
for x in range(8):
    qb = x * tiles // 8
    for s_ in range(nlisted[x]):
        row = masks[int(info[qb + s_][0]) * chunks:(int(info[qb + s_][0]) + 1) * chunks]
        n_ = int(sum(bin(int(m)).count('1') for m in row))
        listings += n_; list_per_tile.append(n_)
n_entries = n_pairs = n_batches = fallback = 0
rows_hist = np.zeros(9, np.int64)
per_tile = []
for x in range(8):
    qb = x * tiles // 8
    for s in range(nlisted[x]):
        tile, first, cnt, _ = info[qb + s]
        if first < 0:
            fallback += 1
            continue
        e = ent[first:first + cnt]
        pc = np.array([bin(int(lo)).count('1') + bin(int(hi)).count('1') for lo, hi in e[:, 2:4]], dtype=np.int64)
        rows_hist += np.bincount([sum(1 for r in range(8) if ((int(lo) | (int(hi) << 32)) >> (8 * r)) & 255) for lo, hi in e[:, 2:4]], minlength=9)[:9]
        n_entries += cnt
        n_pairs += int(pc.sum())
        n_batches += int(np.ceil(pc.sum() / 64.0))
        per_tile.append((int(pc.sum()), int(cnt), x, s))
P = isz * isz
print('%s batch %d: %d tiles, %d listed (%.1f %%), %s listings (bin), %d entries (cover) = %.1f per listed tile, %d pairs = %.1f per entry, '
      '%.2f pairs per pixel, >= %d batches (%.1f per listed tile), %d tiles without a pool slice'
      % (args.config, Bn, tiles, sum(nlisted), 100.0 * sum(nlisted) / tiles, listings, n_entries, n_entries / max(1, sum(nlisted)),
         n_pairs, n_pairs / max(1, n_entries), n_pairs / float(Bn * P), n_batches, n_batches / max(1, sum(nlisted)), fallback))
lp = np.array(list_per_tile)
print('faces listed by the binning kernel per listed tile, percentiles [0, 10, 25, 50, 75, 90, 99, 100]:', [int(v) for v in np.percentile(lp, [0, 10, 25, 50, 75, 90, 99, 100])], ' mean %.1f;  sixteen-face steps per tile: mean %.2f' % (lp.mean(), np.ceil(lp / 16.0).mean()))
print('pixel rows per entry (histogram 0..8):', rows_hist.tolist(), ' mean %.2f' % ((rows_hist * np.arange(9)).sum() / max(1, rows_hist.sum())))
pt = np.array([t[0] for t in per_tile]); en = np.array([t[1] for t in per_tile])
q = [0, 10, 25, 50, 75, 90, 99, 100]
print('pairs per listed tile, percentiles', q, ':', [int(v) for v in np.percentile(pt, q)], ' mean %.0f' % pt.mean())
print('entries per listed tile, percentiles', q, ':', [int(v) for v in np.percentile(en, q)], ' mean %.0f' % en.mean())
print('share of all pairs in the heaviest 1 %% / 10 %% of tiles: %.1f %% / %.1f %%'
      % (100.0 * np.sort(pt)[-max(1, len(pt) // 100):].sum() / pt.sum(), 100.0 * np.sort(pt)[-max(1, len(pt) // 10):].sum() / pt.sum()))
