"""Reads the tile masks / queue counters of one forward call back from the workspace: how many tiles list a face, how
many faces a listed tile lists, how balanced the 8 queues are.  (Layout as in csrc/gendr_capi.hip workspace_layout.)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from gendr_amd.functional import renderer as R
from gendr_amd.synthetic import benchmark_scene

name = sys.argv[1] if len(sys.argv) > 1 else 'c2'
cfg = bench.CONFIGS[name]
B, isz = cfg['batch'], cfg['image_size']
o = dict(background_color=[0, 0, 0], dist_func='uniform', dist_scale=1e-2, dist_squared=False, dist_shape=None, dist_shift=None,
         dist_eps=1e4, aggr_alpha_func='probabilistic', aggr_alpha_t_conorm_p=None, aggr_rgb_func='softmax', aggr_rgb_eps=1e-3,
         aggr_rgb_gamma=1e-3, near=1, far=100, double_side=False, texture_type='surface')
o.update(cfg['opts'])
fv, tex = benchmark_scene(B, subdivisions=cfg['subdiv'], texture=cfg['texture'], seed=0)
fv, tex = fv.cuda().reshape(B, -1, 9).contiguous(), tex.cuda().contiguous()
nf, T = fv.shape[1], tex.shape[2]
params = R.make_params(isz, **o)
rgba, aux, ws = R.native_forward(fv, tex, params)
torch.cuda.synchronize()
ws = ws.cpu().numpy()
al = lambda v: (v + 255) // 256 * 256
rec = {True: 60, False: 56 if T == 1 else 48}[o['texture_type'] == 'vertex']
tiles_x, chunks = (isz + 7) // 8, (nf + 63) // 64
tpi = tiles_x * tiles_x
records_off = al(B * nf * 16)
masks_off = records_off + al(B * nf * rec * 4)
lists_off = masks_off + al(B * tpi * chunks * 8)
control_off = lists_off + al(B * tpi * 4)
masks = ws[masks_off:masks_off + B * tpi * chunks * 8].view(np.uint64).reshape(B * tpi, chunks)
pop = np.zeros(B * tpi, np.int64)
for c in range(chunks):
    v = masks[:, c].copy()
    pop += np.unpackbits(v.view(np.uint8).reshape(-1, 8), axis=1).sum(1).astype(np.int64)
ctl = ws[control_off:control_off + 16 * 1024 * 4].view(np.int32)[::1024]
print('%s: B=%d tiles=%d listed tiles=%d (%.1f%%)  listings=%d  mean faces per listed tile %.1f  max %d'
      % (name, B, B * tpi, (pop > 0).sum(), 100.0 * (pop > 0).mean(), pop.sum(), pop[pop > 0].mean(), pop.max()))
print('queue lengths', ctl[:8].tolist(), ' unlisted', ctl[8:].tolist())
hist = np.bincount(np.minimum(pop[pop > 0], 64))
print('faces-per-tile histogram (1..): ', hist[1:41].tolist())
per_image = (pop > 0).reshape(B, tpi).sum(1)
cost = pop.reshape(B, tpi).sum(1)                      # listings per image ~ phase A work
for label, q in (('contiguous', np.arange(B) * 8 // B), ('round-robin', np.arange(B) % 8)):
    t = np.bincount(q, weights=per_image, minlength=8)
    c = np.bincount(q, weights=cost, minlength=8)
    print('%-12s tiles max/mean %.3f   listings max/mean %.3f' % (label, t.max() / t.mean(), c.max() / c.mean()))
half = (pop > 0).reshape(B, 2, tpi // 2).sum(2).reshape(-1)
hc = pop.reshape(B, 2, tpi // 2).sum(2).reshape(-1)
q = np.arange(2 * B) % 8
print('half-images round-robin: tiles max/mean %.3f  listings max/mean %.3f' % (
    np.bincount(q, weights=half).max() / np.bincount(q, weights=half).mean(), np.bincount(q, weights=hc).max() / np.bincount(q, weights=hc).mean()))
print('per-image listed tiles', per_image.tolist())
