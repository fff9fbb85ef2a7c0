import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import parity, scenes
from gendr_amd.functional import renderer as R
fv, tex = scenes.slivers()
isz = 64
opts = {}
a = parity.run_hip(fv, tex, isz, dict(opts, cull=1))
b = parity.run_hip(fv, tex, isz, dict(opts, cull=0))
d = np.argwhere(a['rgba'] != b['rgba'])
print('differing elements', len(d), 'of', a['rgba'].size)
pix = sorted(set((int(x[0]), int(x[2]), int(x[3])) for x in d))
print('pixels (image,row,col):', pix[:40])
o, extra = parity.split_options(opts)
p = parity.hip_params(isz, o, extra)
Bn, nf = fv.shape[:2]
faces = torch.from_numpy(fv).reshape(Bn, nf, 9).cuda().contiguous(); t = torch.from_numpy(tex).cuda().contiguous()
rgba, aux, ws = R.native_forward(faces, t, p)
torch.cuda.synchronize()
w = ws.cpu().numpy()
a256 = lambda v: (v + 255) // 256 * 256
control_off = len(w) - 24 * 1024 * 4
off = control_off - a256(Bn * nf * 4) - a256(Bn * nf * 16) - a256(Bn * 4)
flag = w[off:off + Bn * nf * 4].view(np.int32).reshape(Bn, nf); off += a256(Bn * nf * 4)
box = w[off:off + Bn * nf * 16].view(np.int32).reshape(Bn, nf, 4); off += a256(Bn * nf * 16)
img = w[off:off + Bn * 4].view(np.int32)
print('image stamps', img, 'flagged per image', (flag != 0).sum(1))
for bb in range(Bn):
    for f in np.nonzero(flag[bb])[0]:
        print('  image %d face %d box cols %s rows %s' % (bb, f, tuple(box[bb, f, :2]), tuple(box[bb, f, 2:])))
