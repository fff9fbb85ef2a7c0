"""Time line of the backward kernel's waves from a -DGENDR_TRACE=1 build (diagnostic):
    bash tools/variant.sh gpurun_ablate_trace.so -DGENDR_TRACE=1
    cp gpurun_ablate_trace.so gendr_amd/libgendr_hip.so; python tools/wave_trace.py [--config c2] [--batch N]
Every wave leaves its start, the moment the queue lengths arrived, the moment its tile's pixel inputs were parked, the
entry of its first batch and its end (shader clock, s_waitcnt before every stamp), its batches and pairs."""
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
ap.add_argument('--batch', type=int, default=None)
ap.add_argument('--ghz', type=float, default=2.1)
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
L = _native.lib()
L.gendr_trace_read.restype = ctypes.c_int
L.gendr_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
tiles = Bn * ((isz + 7) // 8) ** 2
nw = min(1 << 17, max(16384 if tiles >= 16384 else tiles, (tiles + 3) // 4))
nw = (nw + 7) // 8 * 8
for it in range(3):
    rgba, aux, ws = R.native_forward(faces, t, p)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    R.native_backward(faces, t, rgba, aux, ws, grad, p)
    e1.record()
    torch.cuda.synchronize()
buf = np.zeros((nw, 8), np.uint64)
assert L.gendr_trace_read(buf.ctypes.data, nw) == 0
tr = buf.astype(np.float64)
us = lambda c: c / (args.ghz * 1e3)            # shader-clock cycles (differences inside one wave only: every CU has its own origin)
rt0 = tr[:, 6].min()
start = (tr[:, 6] - rt0) / 100.0               # us on the chip-wide 100 MHz clock
end = (tr[:, 7] - rt0) / 100.0
busy = tr[:, 2] > 0
print('%s batch %d: %d waves traced, %d with a tile; backward call %.1f us by events (zero fill included), first start -> last end %.1f us'
      % (args.config, Bn, nw, int(busy.sum()), e0.elapsed_time(e1) * 1e3, end.max()))
q = [0, 10, 50, 90, 99, 100]
pr = lambda name, v, f=us: print('  %-46s' % name, ' '.join('%8.1f' % f(x) for x in np.percentile(v, q)))
ident = lambda x: x
print('  %-46s' % 'us, percentiles', ' '.join('%8d' % x for x in q))
pr('wave start (dispatch), all waves', start, ident)
pr('wave start, waves with a tile', start[busy], ident)
pr('wave end, waves with a tile', end[busy], ident)
if (~busy).any():
    pr('idle wave: lifetime', (tr[:, 4] - tr[:, 0])[~busy])
b = tr[busy]
pr('busy wave: start -> queue lengths', b[:, 1] - b[:, 0])
pr('busy wave: -> pixel inputs parked', b[:, 2] - b[:, 1])
has = b[:, 3] > 0
pr('busy wave: -> first batch entered', (b[:, 3] - b[:, 2])[has])
pr('busy wave: first batch -> end', (b[:, 4] - b[:, 3])[has])
pr('busy wave: lifetime', b[:, 4] - b[:, 0])
nb = b[has, 5]
print('  batches per busy wave: mean %.2f max %d; (first batch -> end) / batches: median %.2f us'
      % (nb.mean(), int(nb.max()), us(np.median((b[has, 4] - b[has, 3]) / nb))))
for i in np.argsort(end)[-5:]:
    print('  a last wave: start %.1f, end %.1f us, %d batches' % (start[i], end[i], int(tr[i, 5])))
if os.environ.get('WAVE_TRACE_RANKS'):
    # queue 0's busy waves by rank (w = blockIdx * 4 + wave; rank = (blockIdx >> 3) * 4 + wave, queue = blockIdx & 7): start, end, batches
    w = np.arange(nw); blk = w >> 2; rank = (blk >> 3) * 4 + (w & 3)
    sel = np.where(((blk & 7) == 0) & busy)[0]
    sel = sel[np.argsort(rank[sel])]
    step = max(1, len(sel) // 48)
    print('  queue 0, busy waves by rank (every %d-th): rank start end batches' % step)
    print('   ', ' | '.join('%d %.1f %.1f %d' % (rank[i], start[i], end[i], int(tr[i, 5])) for i in sel[::step]))
edges = np.arange(0, end.max() + 5, 5.0)
print('  waves alive per 5-us bin (all / with a tile):',
      ' '.join('%d/%d' % (int(((start < e + 5) & (end > e)).sum()), int(((start < e + 5) & (end > e) & busy).sum())) for e in edges[:-1]))

# ---- start / end spans of the other kernels of the same step (coverage: per wave, binning: per workgroup, forward: per wave)
L.gendr_span_read.restype = ctypes.c_int
L.gendr_span_read.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
supers = Bn * ((isz + 63) // 64) ** 2
for k, name, n in ((1, 'bin_faces_kernel (workgroups)', min(supers, 1 << 16)), (0, 'cover_kernel (waves)', min(nw, 1 << 16)),
                   (2, 'render_forward_kernel (waves)', min(nw, 1 << 16))):
    sp = np.zeros((n, 2), np.uint64)
    assert L.gendr_span_read(sp.ctypes.data, k, n) == 0
    sp = sp.astype(np.float64)
    ok = sp[:, 1] > 0
    st = (sp[ok, 0] - sp[ok, 0].min()) / 100.0
    en = (sp[ok, 1] - sp[ok, 0].min()) / 100.0
    print('%s: %d traced, first start -> last end %.1f us' % (name, int(ok.sum()), en.max()))
    pr('  start', st, ident)
    pr('  lifetime', en - st, ident)
    edges = np.arange(0, en.max() + 5, 5.0)
    print('    alive per 5-us bin:', ' '.join(str(int(((st < e + 5) & (en > e)).sum())) for e in edges[:-1]))
