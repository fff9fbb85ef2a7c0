import sys
import numpy as np
sys.path.insert(0, 'tests')
sys.path.insert(0, '.')
import parity  # noqa: E402
import scenes
from oracle import ref_gpu

name = sys.argv[1] if len(sys.argv) > 1 else 'logistic_prob'
scene = sys.argv[2] if len(sys.argv) > 2 else 'sphere'
opts = dict(scenes.OPTION_MATRIX)[name]
kw = {}
if opts.get('texture_type') == 'vertex':
    kw['vertex_tex'] = True
fv, tex = (scenes.sphere(B=2, **kw) if scene == 'sphere' else scenes.soup(B=2, nf=24, **kw))
isz = 32
o, extra = parity.split_options(opts)
p = parity.hip_params(isz, o, extra)
grad = np.random.RandomState(5).randn(fv.shape[0], 4, isz, isz)
dt = np.float64
r = ref_gpu.render(fv, tex, isz, p, grad.astype(dt), dt)
c = parity.run_oracle(fv.astype(dt), tex.astype(dt), isz, opts, grad.astype(dt), dt)
h = parity.run_hip(fv, tex, isz, opts, grad.astype(np.float32))
d = np.abs(r['rgba'] - c['rgba'])
idx = np.argwhere(d > 1e-9)
print(name, scene, 'differing rgba elements', len(idx), 'of', d.size)
for i in idx[:12]:
    i = tuple(i)
    print(i, 'ref %.12g oracle %.12g hip %.9g | aggrs ref %s oracle %s' % (r['rgba'][i], c['rgba'][i], h['rgba'][i],
          r['aggrs_info'][i[0], :, i[2], i[3]], c['aggrs_info'][i[0], :, i[2], i[3]]))
    print('    rgba ref', r['rgba'][i[0], :, i[2], i[3]], 'oracle', c['rgba'][i[0], :, i[2], i[3]])
g = np.abs(r['grad_faces'].reshape(c['grad_faces'].shape) - c['grad_faces'])
gi = np.argwhere(g > 1e-9 * np.maximum(1e-30, c['abs_faces']))
print('differing grad_faces elements', len(gi), 'of', g.size)
order = np.argsort(-g[tuple(gi.T)]) if len(gi) else []
for i in gi[order][:8]:
    i = tuple(i)
    print(i, 'ref %.12g oracle %.12g abs %.3g' % (r['grad_faces'].reshape(c['grad_faces'].shape)[i], c['grad_faces'][i], c['abs_faces'][i]))
