"""CPU replica of face_setup_kernel's cull box (numpy) to study list lengths per 16x16 tile / 8x8 quadrant."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gendr_amd.synthetic import benchmark_scene

def boxes(fv, cull_r, sthr=10.0):
    f = fv.astype(np.float32)
    x = f[..., 0]; y = f[..., 1]
    X = x.astype(np.float64); Y = y.astype(np.float64)
    def geom(x, y, dt):
        x0,x1,x2 = x[...,0],x[...,1],x[...,2]; y0,y1,y2 = y[...,0],y[...,1],y[...,2]
        adj = np.stack([y1-y2, x2-x1, x1*y2-x2*y1, y2-y0, x0-x2, x2*y0-x0*y2, y0-y1, x1-x0, x0*y1-x1*y0], -1).astype(dt)
        det = (x2*(y0-y1) + x0*(y1-y2) + x1*(y2-y0)).astype(dt)
        return adj, det
    adj32, det32 = geom(x, y, np.float32)
    det32c = np.where(det32 > 0, np.maximum(det32.astype(np.float64), 1e-10), np.minimum(det32.astype(np.float64), -1e-10)).astype(np.float32)
    inv32 = (adj32 / det32c[..., None]).astype(np.float32)
    adj64, det64 = geom(X, Y, np.float64)
    inv64 = adj64 / det64[..., None]
    eps = 1.1920928955078125e-07
    vn = np.abs(X) + np.abs(Y)          # [...,3]
    dinv = np.abs(inv32.astype(np.float64) - inv64).reshape(inv32.shape[:-1] + (3, 3)).sum(-1)
    wk = np.abs(inv32.astype(np.float64)).reshape(inv32.shape[:-1] + (3, 3)).sum(-1)
    E = ((dinv + 4 * eps * wk) * vn).sum(-1)
    E = 2 * E + 8 * eps * (1 + wk.max(-1)) * vn.sum(-1)
    Rf = cull_r * (1 + 1 / 1024.) + E
    xmax, xmin, ymax, ymin = x.max(-1), x.min(-1), y.max(-1), y.min(-1)
    return xmin - Rf, xmax + Rf, ymin - Rf, ymax + Rf, E, det64

def main():
    B = 8; isz = int(sys.argv[1]) if len(sys.argv) > 1 else 256; r = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
    fv, _ = benchmark_scene(B)
    fv = fv.numpy()
    xlo, xhi, ylo, yhi, E, det = boxes(fv, r)
    print('E percentiles (50,90,99,99.9,max):', np.percentile(E, [50, 90, 99, 99.9, 100]))
    print('faces with E > tau:', (E > r).mean(), ' E > 0.1:', (E > 0.1).mean())
    pc = (2 * np.arange(isz) + 1 - isz) / isz
    for ts in (16, 8):
        nt = isz // ts
        lo = pc[0::ts]; hi = pc[ts - 1::ts]
        # tile x-range [lo_i, hi_i]; y similar (flip irrelevant for counting)
        hx = ~((lo[None, None, :] > xhi[..., None]) | (hi[None, None, :] < xlo[..., None]))   # [B,nf,nt]
        hy = ~((lo[None, None, :] > yhi[..., None]) | (hi[None, None, :] < ylo[..., None]))
        cnt = np.einsum('bfx,bfy->bxy', hx.astype(np.float64), hy.astype(np.float64))
        print('tile %2d: mean list %.1f  max %d  nonempty frac %.2f  mean over nonempty %.1f' % (ts, cnt.mean(), cnt.max(), (cnt > 0).mean(), cnt[cnt > 0].mean()))
    # per-pixel box hits
    hx = ~((pc[None, None, :] > xhi[..., None]) | (pc[None, None, :] < xlo[..., None]))
    hy = ~((pc[None, None, :] > yhi[..., None]) | (pc[None, None, :] < ylo[..., None]))
    cnt = np.einsum('bfx,bfy->bxy', hx.astype(np.float64), hy.astype(np.float64))
    print('per-pixel box hits: mean %.2f max %d' % (cnt.mean(), cnt.max()))

main()
