"""Lane-utilisation statistics of the tile loop for one benchmark image (numpy, float64 geometry):
per (8x8 tile, face): lanes in the cull box, lanes passing the edge reject, lanes that truly contribute."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
from gendr_amd.synthetic import benchmark_scene

def main(isz=256, r=0.01, img=3):
    fv, _ = benchmark_scene(img + 1)
    f = fv[img].numpy().astype(np.float64)       # [nf,3,3]
    nf = f.shape[0]
    x, y = f[..., 0], f[..., 1]
    pc = (2 * np.arange(isz) + 1 - isz) / isz
    PX, PY = np.meshgrid(pc, pc[::-1])           # row 0 = top
    P = np.stack([PX.ravel(), PY.ravel()], -1)    # [P,2]
    # boxes (ignore E, tiny)
    xlo, xhi, ylo, yhi = x.min(-1) - r, x.max(-1) + r, y.min(-1) - r, y.max(-1) + r
    inbox = (P[:, None, 0] <= xhi) & (P[:, None, 0] >= xlo) & (P[:, None, 1] <= yhi) & (P[:, None, 1] >= ylo)   # [P,nf]
    # signed distance to triangle (negative inside)
    def seg_dist(p, a, b):
        ab = b - a; t = np.clip(((p[:, None, :] - a) * ab).sum(-1) / np.maximum((ab * ab).sum(-1), 1e-30), 0, 1)
        c = a + t[..., None] * ab
        return np.sqrt(((p[:, None, :] - c) ** 2).sum(-1))
    v = f[..., :2]
    d = np.minimum(np.minimum(seg_dist(P, v[:, 0], v[:, 1]), seg_dist(P, v[:, 1], v[:, 2])), seg_dist(P, v[:, 2], v[:, 0]))
    def edge(p, a, b): return (b[:, 0] - a[:, 0]) * (p[:, None, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (p[:, None, 0] - a[:, 0])
    e0, e1, e2 = edge(P, v[:, 0], v[:, 1]), edge(P, v[:, 1], v[:, 2]), edge(P, v[:, 2], v[:, 0])
    inside = ((e0 >= 0) & (e1 >= 0) & (e2 >= 0)) | ((e0 <= 0) & (e1 <= 0) & (e2 <= 0))
    contrib = inbox & (inside | (d < r))
    # edge-reject analogue: beyond some edge line by > r
    area2 = np.abs((v[:, 1, 0] - v[:, 0, 0]) * (v[:, 2, 1] - v[:, 0, 1]) - (v[:, 1, 1] - v[:, 0, 1]) * (v[:, 2, 0] - v[:, 0, 0]))
    orient = np.sign((v[:, 1, 0] - v[:, 0, 0]) * (v[:, 2, 1] - v[:, 0, 1]) - (v[:, 1, 1] - v[:, 0, 1]) * (v[:, 2, 0] - v[:, 0, 0]))
    def lined(e, a, b): return -orient * e / np.maximum(np.sqrt(((b - a) ** 2).sum(-1)), 1e-30)     # >0 outside
    beyond = (lined(e0, v[:, 0], v[:, 1]) > r) | (lined(e1, v[:, 1], v[:, 2]) > r) | (lined(e2, v[:, 2], v[:, 0]) > r)
    edgepass = inbox & ~beyond
    T = isz // 8
    def per_tile(m):   # [P,nf] -> [T,T,nf] lane counts
        return m.reshape(T, 8, T, 8, nf).sum((1, 3))
    nb, ne, nc, ni = per_tile(inbox), per_tile(edgepass), per_tile(contrib), per_tile(contrib & inside)
    tiles = T * T
    print('per tile: box wave-evals %.2f | edge-pass wave-evals %.2f | contributing wave-evals %.2f' % ((nb > 0).sum() / tiles, (ne > 0).sum() / tiles, (nc > 0).sum() / tiles))
    print('lanes per box wave-eval %.1f | per edge-pass wave-eval: pass %.1f contributing %.1f' % (nb[nb > 0].mean(), ne[ne > 0].mean(), nc[ne > 0].mean()))
    both = ((ni > 0) & ((nc - ni) > 0)).sum() / max(1, (nc > 0).sum())
    print('wave-evals with both inside and outside contributing lanes: %.2f ; with inside lanes: %.2f' % (both, (ni > 0).sum() / max(1, (nc > 0).sum())))
    print('pairs per pixel: box %.2f edge-pass %.2f contributing %.2f' % (inbox.sum() / P.shape[0], edgepass.sum() / P.shape[0], contrib.sum() / P.shape[0]))

main()
