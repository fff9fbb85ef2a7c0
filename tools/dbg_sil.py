import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch, scenes
from gendr_amd.functional import render, render_silhouette
fv, tex = scenes.soup()
fv, tex = torch.from_numpy(fv).cuda(), torch.from_numpy(tex).cuda()
full = render(fv, tex, image_size=48); torch.cuda.synchronize(); print('full ok', flush=True)
sil = render_silhouette(fv, image_size=48); torch.cuda.synchronize(); print('sil fwd ok', bool(torch.equal(sil, full[:, 3])), flush=True)
x = fv.clone().requires_grad_(True)
s = render_silhouette(x, image_size=48); s.sum().backward(); torch.cuda.synchronize(); print('sil bwd ok', float(x.grad.abs().sum()), flush=True)
