"""Prints the HIP-vs-oracle parity table over the option matrix (run on the GPU box)."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import scenes, parity

def main():
    out = {}
    for scene_name, maker, isz in (("soup", scenes.soup, 48), ("sphere", scenes.sphere, 64)):
        for name, opts in scenes.OPTION_MATRIX:
            kw = {}
            if opts.get('texture_type') == 'vertex':
                kw['vertex_tex'] = True
            if 'T' in opts:
                kw['T'] = opts['T']
            fv, tex = maker(**kw)
            t = time.time()
            res, h, r = parity.compare(fv, tex, isz, opts)
            # culled vs all-pairs must be bit-identical
            o2 = dict(opts); o2['cull'] = 0
            rs = np.random.RandomState(1)
            grad = rs.randn(fv.shape[0], 4, isz, isz).astype(np.float32)
            h2 = parity.run_hip(fv, tex, isz, o2, grad)
            same = all(np.array_equal(h[k], h2[k], equal_nan=True) for k in ('rgba', 'aggrs_info'))
            gsame = max(float(np.abs(h[k] - h2[k]).max()) for k in ('grad_faces', 'grad_textures'))
            print(parity.fmt(scene_name + ':' + name, res), 'cull==allpairs:', same, 'grad cull diff %.2e' % gsame, flush=True)
            out[scene_name + ':' + name] = dict(res=res, cull_identical=bool(same), grad_cull_maxdiff=gsame)
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(out, open('gpurun_out/parity_report.json', 'w'), indent=1)

if __name__ == '__main__':
    main()
