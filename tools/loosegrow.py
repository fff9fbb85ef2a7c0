import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import bench as B, parity
from gendr_amd.functional import renderer as R
from gendr_amd.synthetic import benchmark_scene
cfg = B.CONFIGS['c5']; Bn = 8; isz = cfg['image_size']
a256 = lambda v: (v + 255) // 256 * 256
for eps in (1e4, 300.0):
    opts = dict(cfg['opts'], dist_eps=eps); opts.setdefault('double_side', False)
    fv, tex = benchmark_scene(Bn, subdivisions=cfg['subdiv'], texture=cfg['texture'])
    o, extra = parity.split_options(opts)
    p = parity.hip_params(isz, o, extra)
    faces = fv.reshape(Bn, -1, 9).cuda().contiguous(); t = tex.cuda().contiguous()
    nf = faces.shape[1]
    keep = []
    for it in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rgba, aux, ws = R.native_forward(faces, t, p)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        w = ws.cpu().numpy()
        control_off = len(w) - 24 * 1024 * 4
        off = control_off - a256(Bn * 16 * 4)
        heads = w[off:off + Bn * 16 * 4].view(np.int32).reshape(Bn, 16)
        off2 = control_off - a256(Bn * nf * 4) - a256(Bn * nf * 16) - a256(Bn * 16 * 4)
        flag = w[off2:off2 + Bn * nf * 4].view(np.int32).reshape(Bn, nf)
        print('eps %g call %d: %.2f ms, flagged faces %d, heads (stamp, n): %s' % (eps, it, dt * 1e3, int((flag != 0).sum()), [(int(h) >> 4, int(h) & 15) for h in heads[:, 0]]), flush=True)
        if it % 2: keep.append(ws)
