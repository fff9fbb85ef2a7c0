"""Reference kernels (oracle/_ref) vs the C oracle vs the HIP product on the option matrix: exploratory report."""
import json
import sys

import numpy as np

sys.path.insert(0, 'tests')
sys.path.insert(0, '.')
import parity  # noqa: E402
import scenes
from oracle import ref_gpu


def inputs(opts, scene):
    kw = {}
    if opts.get('texture_type') == 'vertex':
        kw['vertex_tex'] = True
    if 'T' in opts:
        kw['T'] = opts['T']
    if scene == 'soup':
        return scenes.soup(B=2, nf=24, **kw)
    if scene == 'slivers':
        return scenes.slivers(B=1, nf=36, **kw)
    return scenes.sphere(B=2, **kw)


def main():
    isz = 32
    rows = []
    for name, opts in scenes.OPTION_MATRIX:
        for scene in ('soup', 'sphere', 'slivers'):
            fv, tex = inputs(opts, scene)
            o, extra = parity.split_options(opts)
            p = parity.hip_params(isz, o, extra)
            grad = np.random.RandomState(5).randn(fv.shape[0], 4, isz, isz)
            row = dict(case=name, scene=scene)
            for dt, tag in ((np.float32, 'f32'), (np.float64, 'f64')):
                r = ref_gpu.render(fv, tex, isz, p, grad.astype(dt), dt)
                c = parity.run_oracle(fv.astype(dt), tex.astype(dt), isz, opts, grad.astype(dt), dt)
                for k in ('faces_info', 'rgba', 'aggrs_info'):
                    s = parity.stats(r[k], c[k])
                    row['%s_%s' % (tag, k)] = (s['exact'], s['max_rel'], s['frac_gt_1e5'])
                for k, ak in (('grad_faces', 'abs_faces'), ('grad_textures', 'abs_textures')):
                    s = parity.stats(r[k].reshape(c[k].shape), c[k], scale=c[ak])
                    row['%s_%s' % (tag, k)] = (s['exact'], s['max_rel'], s['p99_rel'])
            rf = ref_gpu.render(fv, tex, isz, p, grad.astype(np.float32), np.float32, variant='gendr_ref_kernels_fma')
            r0 = ref_gpu.render(fv, tex, isz, p, grad.astype(np.float32), np.float32)
            row['fma_vs_nofma_rgba'] = parity.stats(rf['rgba'], r0['rgba'])['max_rel']
            row['fma_vs_nofma_gf'] = parity.stats(rf['grad_faces'], r0['grad_faces'], scale=c['abs_faces'].reshape(r0['grad_faces'].shape))['max_rel']
            rows.append(row)
            print(json.dumps(row), flush=True)
    json.dump(rows, open('gpurun_out/refpin_study.json', 'w'), indent=0)


main()
