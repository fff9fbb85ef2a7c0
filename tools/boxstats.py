"""Cull-box sizes (pixels) of the benchmark scene's faces, read back from the workspace: python tools/boxstats.py [c2|c5] [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import bench as B, parity
from gendr_amd.functional import renderer as R
from gendr_amd.synthetic import benchmark_scene
cfg = B.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'c2']
Bn = int(sys.argv[2]) if len(sys.argv) > 2 else 16
isz = cfg['image_size']
opts = dict(cfg['opts']); opts.setdefault('double_side', False)
fv, tex = benchmark_scene(Bn, subdivisions=cfg['subdiv'], texture=cfg['texture'])
o, extra = parity.split_options(opts)
p = parity.hip_params(isz, o, extra)
faces = fv.reshape(Bn, -1, 9).cuda().contiguous(); t = tex.cuda().contiguous()
rgba, aux, ws = R.native_forward(faces, t, p)
nf = faces.shape[1]
boxes = ws[:Bn * nf * 64].view(torch.float32).view(Bn, nf, 16)[:, :, :4].cpu().numpy()
W = (np.clip(boxes[..., 1], -1, 1) - np.clip(boxes[..., 0], -1, 1)) * isz / 2
H = (np.clip(boxes[..., 3], -1, 1) - np.clip(boxes[..., 2], -1, 1)) * isz / 2
A = np.maximum(W, 0) * np.maximum(H, 0)
print('W pct', np.percentile(W, [50, 90, 99, 99.9, 100]).round(1), 'H', np.percentile(H, [50, 90, 99, 99.9, 100]).round(1))
print('area pct', np.percentile(A, [50, 90, 99, 99.9, 100]).round(0), 'faces with area > 4096:', int((A > 4096).sum()), 'of', A.size, ' sum area / image', A.sum() / Bn / isz / isz)
big = np.argwhere(A > 4096)[:5]
for b, f in big:
    print('big', b, f, boxes[b, f], fv[b, f].numpy().round(4).tolist())
