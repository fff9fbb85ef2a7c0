"""HIP-vs-oracle parity table at HEAD (run on the GPU box):

    python tests/gpu_report.py [tag]        -> gpurun_out/parity_<tag>.json  (copy it to profiles/)

For every option set of the matrix on the three test scenes, and for BASELINE configs C2..C5 at their full image
size (one frame each), per tensor: max / p99 relative error, fraction above 1e-5, fraction bit-identical -- for rgba,
alpha, aggrs_info, the raw gradients (error / |reference element|) and the conditioned gradients (error / sum of
|contributions|) -- with the oracle's own fp32-vs-fp64 spread (the noise floor of tests/criteria.py) beside it, the
verdict of the acceptance rule, and whether the culled traversal is bit-identical to the all-pairs one."""
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np

import criteria
import parity
import scenes

FULL = {
    'C2': (256, 'surface', dict(dist_func='uniform', dist_scale=1e-2, aggr_alpha_func='probabilistic', aggr_rgb_func='softmax', double_side=False)),
    'C3': (256, 'surface', dict(dist_func='gaussian', dist_scale=1e-4, dist_squared=True, aggr_alpha_func='einstein', double_side=False)),
    'C4': (512, 'surface', dict(dist_func='logistic', dist_scale=1e-2, aggr_alpha_func='probabilistic', aggr_rgb_func='softmax', double_side=False)),
    'C5': (2048, 'vertex', dict(dist_func='gamma', dist_shape=2.0, dist_scale=1e-2, aggr_alpha_func='yager', aggr_alpha_t_conorm_p=2.0,
                                aggr_rgb_func='softmax', texture_type='vertex', double_side=False)),
}


def one_case(fv, tex, isz, opts, with_cull_check=True):
    res, h, r = parity.compare(fv, tex, isz, opts)
    grad = np.random.RandomState(1).randn(fv.shape[0], 4, isz, isz).astype(np.float32)
    noise = criteria.noise_floor(fv, tex, isz, opts, grad, oracle_f32=r)
    entry = dict(hip_vs_oracle=res, oracle_fp32_vs_fp64=noise, accepted=not criteria.check(res, noise),
                 failures=criteria.check(res, noise))
    if with_cull_check:
        h2 = parity.run_hip(fv, tex, isz, dict(opts, cull=0), grad)
        entry['cull_identical'] = bool(all(np.array_equal(h[k], h2[k], equal_nan=True) for k in ('rgba', 'aggrs_info')))
        entry['grad_cull_maxdiff_rel'] = max(
            float(np.nanmax(np.abs(h[k] - h2[k])) / max(1e-30, float(np.nanmax(np.abs(h2[k]))))) for k in ('grad_faces', 'grad_textures'))
    return entry, res


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'r02'
    try:
        head = subprocess.check_output(['git', 'rev-parse', '--short', 'HEAD'], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        head = None
    out = dict(tag=tag, head=head, date=time.strftime('%Y-%m-%d %H:%M:%S'),
               metric='rel = |hip - oracle| / max(|oracle|, 1e-6 max|oracle|); *_cond: denominator also >= sum of |contributions|',
               matrix={}, full_size={})
    for scene_name, maker, isz in (("soup", scenes.soup, 48), ("sphere", scenes.sphere, 64), ("slivers", scenes.slivers, 64)):
        for name, opts in scenes.OPTION_MATRIX:
            kw = {}
            if opts.get('texture_type') == 'vertex':
                kw['vertex_tex'] = True
            if 'T' in opts:
                kw['T'] = opts['T']
            fv, tex = maker(**kw)
            entry, res = one_case(fv, tex, isz, opts)
            print(parity.fmt(scene_name + ':' + name, res), 'ok' if entry['accepted'] else 'REJECTED', flush=True)
            out['matrix'][scene_name + ':' + name] = entry
    from gendr_amd.synthetic import benchmark_scene
    for name, (isz, texture, opts) in FULL.items():
        fv, tex = benchmark_scene(3, texture=texture)
        fv, tex = fv.numpy()[2:3], tex.numpy()[2:3]
        entry, res = one_case(fv, tex, isz, opts, with_cull_check=name in ('C2', 'C3', 'C4'))
        print(parity.fmt(name + '@%d' % isz, res), 'ok' if entry['accepted'] else 'REJECTED', flush=True)
        out['full_size'][name] = dict(entry, image_size=isz, faces=int(fv.shape[1]))
    keys = ('rgba', 'grad_faces', 'grad_textures', 'grad_faces_cond', 'grad_textures_cond')
    cases = list(out['matrix'].values()) + list(out['full_size'].values())
    out['summary'] = dict(
        cases=len(cases), accepted=sum(1 for c in cases if c['accepted']),
        over_1e5={k: sum(1 for c in cases if c['hip_vs_oracle'][k]['max_rel'] > 1e-5) for k in keys},
        over_1e5_p99={k: sum(1 for c in cases if c['hip_vs_oracle'][k]['p99_rel'] > 1e-5) for k in keys},
        cull_identical=sum(1 for c in cases if c.get('cull_identical', True)))
    os.makedirs('gpurun_out', exist_ok=True)
    path = 'gpurun_out/parity_%s.json' % tag
    json.dump(out, open(path, 'w'), indent=1)
    print('wrote', path, json.dumps(out['summary']))


if __name__ == '__main__':
    main()
