"""HIP-vs-oracle parity table at HEAD (run on the GPU box):

    python tests/gpu_report.py [tag]        -> gpurun_out/parity_<tag>.json  (copy it to profiles/)

For every option set of the matrix on the three test scenes, and for BASELINE configs C2..C5 at their full image
size (one frame each), per tensor: max / p99 relative error, fraction above 1e-5, fraction bit-identical -- for rgba,
aggrs_info and the gradients, for BOTH build variants (default: fp32-accurate gradient side; exact: the reference's
rounding, -DGENDR_EXACT_GRADIENT=1), the report of the element-wise acceptance rule (tests/criteria.py: violations, how
many elements are not held to 1e-5 and why), the conditioned gradient error (error / sum of |contributions|), default
vs exact directly, whether the culled traversal is bit-identical to the all-pairs one, and -- when oracle/_ref is built --
the same inputs through the REFERENCE's own kernels: the HIP product and the restatement against their float output
(element-wise rule), the restatement against their double output (maximum relative deviation)."""
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
import pin
import scenes

FULL = {
    'C2': (256, 'surface', dict(dist_func='uniform', dist_scale=1e-2, aggr_alpha_func='probabilistic', aggr_rgb_func='softmax', double_side=False)),
    'C3': (256, 'surface', dict(dist_func='gaussian', dist_scale=1e-4, dist_squared=True, aggr_alpha_func='einstein', double_side=False)),
    'C4': (512, 'surface', dict(dist_func='logistic', dist_scale=1e-2, aggr_alpha_func='probabilistic', aggr_rgb_func='softmax', double_side=False)),
    'C5': (2048, 'vertex', dict(dist_func='gamma', dist_shape=2.0, dist_scale=1e-2, aggr_alpha_func='yager', aggr_alpha_t_conorm_p=2.0,
                                aggr_rgb_func='softmax', texture_type='vertex', double_side=False)),
}


VARIANTS = ('default', 'exact')
FAST = os.path.exists(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gendr_amd', 'libgendr_hip_fast.so'))


def one_case(fv, tex, isz, opts, with_cull_check=True, n_jitter=len(criteria.JITTER_MODES)):
    """Both build variants against the oracle under the element-wise rule (tests/criteria.py)."""
    grad = np.random.RandomState(1).randn(fv.shape[0], 4, isz, isz).astype(np.float32)
    refs = criteria.references(fv, tex, isz, opts, grad, n_jitter=n_jitter)
    o32 = refs['o32']
    entry = dict(variants={})
    hips = {}
    for v in VARIANTS:
        h = hips[v] = parity.run_hip(fv, tex, isz, opts, grad, variant=v)
        rep = criteria.elementwise(h, refs)
        bad = criteria.failures(rep)
        cond = {k: parity.stats(h[k], o32[k], scale=o32[ak]) for k, ak in criteria.GRAD_KEYS}
        entry['variants'][v] = dict(elementwise=rep, accepted=not bad, failures=bad,
                                    grad_faces_cond=cond['grad_faces'], grad_textures_cond=cond['grad_textures'])
    # how much of the deviation is the relaxed gradient arithmetic: default vs exact, relative to the sum of |contributions|
    entry['default_vs_exact'] = {k: parity.stats(hips['default'][k], hips['exact'][k], scale=o32[ak].reshape(hips['exact'][k].shape))
                                 for k, ak in criteria.GRAD_KEYS}
    entry['rgba'] = parity.stats(hips['default']['rgba'], o32['rgba'])
    entry['oracle_fp32_vs_fp64'] = dict(rgba=parity.stats(o32['rgba'], refs['o64']['rgba']),
                                        grad_faces_cond=parity.stats(o32['grad_faces'], refs['o64']['grad_faces'], scale=refs['o64']['abs_faces']))
    entry['accepted'] = all(e['accepted'] for e in entry['variants'].values())
    if parity.reference_available() and parity.split_options(opts)[1]['texel_mode'] == 0:
        # the pin: the REFERENCE's own kernels (oracle/_ref) on the same inputs
        r32 = parity.run_reference(fv, tex, isz, opts, grad, np.float32)
        r64 = parity.run_reference(fv, tex, isz, opts, grad, np.float64)
        o64 = refs['o64']
        pinned = dict(refs, o32=dict(r32, abs_faces=o32['abs_faces'], abs_textures=o32['abs_textures'],
                                     grad_faces=r32['grad_faces'].reshape(o32['grad_faces'].shape)))
        rep = criteria.elementwise(hips['default'], pinned)
        rep_o = criteria.elementwise(o32, pinned)
        entry['reference_kernels'] = dict(
            hip_vs_reference_f32=dict(elementwise=rep, accepted=not criteria.failures(rep), failures=criteria.failures(rep),
                                      rgba=parity.stats(hips['default']['rgba'], r32['rgba']),
                                      grad_faces_cond=parity.stats(hips['default']['grad_faces'], r32['grad_faces'].reshape(o32['grad_faces'].shape), scale=o32['abs_faces'])),
            restatement_vs_reference_f32=dict(accepted=not criteria.failures(rep_o), failures=criteria.failures(rep_o),
                                              faces_info_identical=bool(np.array_equal(o32['faces_info'], r32['faces_info'], equal_nan=True)),
                                              rgba=parity.stats(o32['rgba'], r32['rgba'])),
            restatement_vs_reference_f64=dict(
                faces_info_identical=bool(np.array_equal(o64['faces_info'], r64['faces_info'], equal_nan=True)),
                rgba_max=float(parity.rel_error(r64['rgba'], o64['rgba']).max()),
                aggrs_info_max=float(parity.rel_error(r64['aggrs_info'], o64['aggrs_info']).max()),
                grad_faces_max=float(parity.rel_error(r64['grad_faces'], o64['grad_faces'], scale=o64['abs_faces'], floor=parity.GRAD_FLOOR).max()),
                grad_textures_max=float(parity.rel_error(r64['grad_textures'], o64['grad_textures'], scale=o64['abs_textures'], floor=parity.GRAD_FLOOR).max())))
        # the flat gate of tests/pin.py (1e-5 on every element against the reference's kernels) for the three build variants,
        # the spread of the reference's own two builds (contraction off / on), and the fast variant against that spread
        rf = parity.run_reference(fv, tex, isz, opts, grad, np.float32, variant='render_fma')
        spread = pin.measure(rf, r32, o32['abs_faces'], o32['abs_textures'])
        flat = {v: pin.measure(hips[v], r32, o32['abs_faces'], o32['abs_textures']) for v in hips}
        if FAST:
            hf = parity.run_hip(fv, tex, isz, opts, grad, variant='fast')
            flat['fast'] = pin.measure(hf, r32, o32['abs_faces'], o32['abs_textures'])
            br = pin.bracket(hf, r32, rf, o32['abs_faces'], o32['abs_textures'])
            entry['fast'] = dict(vs_reference=flat['fast'], outside_spread=pin.spread_failures('', flat['fast'], spread),
                                 elementwise_bracket={k: dict(violations=b['violations'], n=b['n'], worst_over_bound=b['worst_over_bound']) for k, b in br.items()})
        entry['reference_kernels']['flat_1e5'] = {v: dict(meets=not pin.exceptions_of(m), max={k: x['max'] for k, x in m.items()},
                                                          frac_gt_1e5={k: x['frac'] for k, x in m.items()}) for v, m in flat.items()}
        # the PLAIN relative error SURVEY 8(d) names -- |got - ref| / max(|ref|, 1e-6 max|ref|), no conditioning --
        # of both gradients against the reference kernels' float output, next to the conditioned one the gate uses (pin.measure:
        # relative to max(|ref|, sum of |contributions|)): the two differ only on elements that are the small difference of
        # large contributions, whose float value depends on the summation order (atomics) in the reference as well -- the
        # reference's own two builds are given the same treatment for scale
        def _plain(a):
            return {k: {q: parity.stats(a[k], np.asarray(r32[k]).reshape(np.asarray(a[k]).shape))[q] for q in ('max_rel', 'p99_rel', 'frac_gt_1e5')}
                    for k in ('grad_faces', 'grad_textures') if np.asarray(r32[k]).size}
        entry['reference_kernels']['plain_relative'] = dict({v: _plain(hips[v]) for v in hips}, reference_fma_build=_plain(rf))
        entry['reference_kernels']['conditioned_relative'] = {v: {k: {q: m[k][q] for q in ('max', 'p99', 'frac')} for k in ('grad_faces', 'grad_textures') if k in m}
                                                              for v, m in flat.items()}
        entry['reference_kernels']['reference_fma_vs_nofma'] = {k: {q: x[q] for q in ('max', 'p50', 'p99', 'p999', 'frac')} for k, x in spread.items()}
    if with_cull_check:
        h, h2 = hips['default'], parity.run_hip(fv, tex, isz, dict(opts, cull=0), grad)
        entry['cull_identical'] = bool(all(np.array_equal(h[k], h2[k], equal_nan=True) for k in ('rgba', 'aggrs_info')))
        entry['grad_cull_maxdiff_rel'] = max(
            float(np.nanmax(np.abs(h[k] - h2[k])) / max(1e-30, float(np.nanmax(np.abs(h2[k]))))) for k in ('grad_faces', 'grad_textures'))
    return entry


def short(name, entry):
    parts = [name]
    for v in VARIANTS:
        e = entry['variants'][v]
        parts.append('%s: %s gf max %.1e p99 %.1e >1e-5 %.1e' % (v, 'ok' if e['accepted'] else 'REJECTED', e['grad_faces_cond']['max_rel'],
                                                               e['grad_faces_cond']['p99_rel'], e['grad_faces_cond']['frac_gt_1e5']))
    parts.append('d-vs-e %.1e' % entry['default_vs_exact']['grad_faces']['max_rel'])
    parts.append('rgba max %.1e' % entry['rgba']['max_rel'])
    rk = entry.get('reference_kernels')
    if rk:
        f = rk['restatement_vs_reference_f64']
        parts.append('ref-kernels: hip %s, oracle f32 %s, f64 rgba %.1e gf %.1e' % (
            'ok' if rk['hip_vs_reference_f32']['accepted'] else 'REJECTED', 'ok' if rk['restatement_vs_reference_f32']['accepted'] else 'REJECTED',
            f['rgba_max'], f['grad_faces_max']))
    return ' | '.join(parts)


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'r06'
    try:
        head = subprocess.check_output(['git', 'rev-parse', '--short', 'HEAD'], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        head = None
    out = dict(tag=tag, head=head, date=time.strftime('%Y-%m-%d %H:%M:%S'),
               rule='element-wise: |hip - o32| <= max(1e-5 scale, %g libm-jitter noise, %g threshold flip), tests/criteria.py; '
                    '*_cond: |hip - o32| / max(|o32|, sum of |contributions|)' % (criteria.K_NOISE, criteria.K_FLIP),
               variants='default = shipped build (fp32-accurate gradient side); exact = -DGENDR_EXACT_GRADIENT=1 (reference rounding)',
               matrix={}, full_size={})
    for scene_name, maker, isz in (("soup", scenes.soup, 48), ("sphere", scenes.sphere, 64), ("slivers", scenes.slivers, 64)):
        for name, opts in scenes.OPTION_MATRIX:
            kw = {}
            if opts.get('texture_type') == 'vertex':
                kw['vertex_tex'] = True
            if 'T' in opts:
                kw['T'] = opts['T']
            fv, tex = maker(**kw)
            entry = one_case(fv, tex, isz, opts)
            print(short(scene_name + ':' + name, entry), flush=True)
            out['matrix'][scene_name + ':' + name] = entry
    from gendr_amd.synthetic import benchmark_scene
    only = [a for a in sys.argv[2:]]
    for name, (isz, texture, opts) in FULL.items():
        if only and name not in only:
            continue
        fv, tex = benchmark_scene(3, texture=texture)
        fv, tex = fv.numpy()[2:3], tex.numpy()[2:3]
        entry = one_case(fv, tex, isz, opts, with_cull_check=name in ('C2', 'C3', 'C4'), n_jitter=14 if isz <= 256 else (6 if isz <= 512 else 4))
        print(short(name + '@%d' % isz, entry), flush=True)
        out['full_size'][name] = dict(entry, image_size=isz, faces=int(fv.shape[1]))
    cases = list(out['matrix'].values()) + list(out['full_size'].values())
    summ = dict(cases=len(cases), cull_identical=sum(1 for c in cases if c.get('cull_identical', True)))
    for v in VARIANTS:
        es = [c['variants'][v] for c in cases]
        summ[v] = dict(accepted=sum(1 for e in es if e['accepted']),
                       grad_faces_cond_over_1e5_max=sum(1 for e in es if e['grad_faces_cond']['max_rel'] > 1e-5),
                       grad_faces_cond_over_1e5_p99=sum(1 for e in es if e['grad_faces_cond']['p99_rel'] > 1e-5))
    summ['default_vs_exact_grad_faces_max'] = max(c['default_vs_exact']['grad_faces']['max_rel'] for c in cases)
    pinned = [c['reference_kernels'] for c in cases if 'reference_kernels' in c]
    if pinned:
        non_cauchy = [c['reference_kernels'] for k, c in list(out['matrix'].items()) + list(out['full_size'].items())
                      if 'reference_kernels' in c and 'cauchy' not in k]
        summ['reference_kernels'] = dict(
            what="the reference's own kernels (oracle/_ref, oracle/build_ref.py) run on this GPU on the same inputs",
            cases=len(pinned),
            hip_accepted_f32=sum(1 for r in pinned if r['hip_vs_reference_f32']['accepted']),
            restatement_accepted_f32=sum(1 for r in pinned if r['restatement_vs_reference_f32']['accepted']),
            faces_info_identical=sum(1 for r in pinned if r['restatement_vs_reference_f32']['faces_info_identical'] and r['restatement_vs_reference_f64']['faces_info_identical']),
            restatement_f64_rgba_max_without_cauchy=max(r['restatement_vs_reference_f64']['rgba_max'] for r in non_cauchy),
            restatement_f64_grad_faces_max_without_cauchy=max(r['restatement_vs_reference_f64']['grad_faces_max'] for r in non_cauchy))
        summ['reference_kernels']['flat_1e5_met'] = {v: sum(1 for r in pinned if r['flat_1e5'].get(v, {}).get('meets')) for v in ('default', 'exact', 'fast')
                                                     if any(v in r['flat_1e5'] for r in pinned)}
        summ['reference_kernels']['gradients_at_baseline_configs'] = {
            name: dict(plain=c['reference_kernels']['plain_relative'], conditioned=c['reference_kernels']['conditioned_relative'])
            for name, c in out['full_size'].items() if 'reference_kernels' in c}
        fast = [c['fast'] for c in cases if 'fast' in c]
        if fast:
            summ['fast_variant'] = dict(
                cases=len(fast), inside_reference_spread=sum(1 for f in fast if not f['outside_spread']),
                with_elements_outside_elementwise_bracket=sum(1 for f in fast if any(b['violations'] for b in f['elementwise_bracket'].values())),
                rule='quantiles p50..p99.9 of the error against the reference kernels within %gx the same quantiles of the spread of the '
                     'reference\'s two builds (floor %g), tests/pin.py' % (pin.SPREAD_K, pin.SPREAD_FLOOR))
    out['summary'] = summ
    os.makedirs('gpurun_out', exist_ok=True)
    path = 'gpurun_out/parity_%s.json' % tag

    def compact(x):
        if isinstance(x, float):
            return float('%.4g' % x) if x == x and abs(x) != float('inf') else (None if x != x else (1e300 if x > 0 else -1e300))
        if isinstance(x, dict):
            return {k: compact(v) for k, v in x.items()}
        if isinstance(x, (list, tuple)):
            return [compact(v) for v in x]
        return x
    json.dump(compact(out), open(path, 'w'), separators=(',', ':'))
    print('ref_pin: ran %d cases against the reference kernels (oracle/_ref)' % len(pinned))
    print('wrote', path, json.dumps(out['summary']))


if __name__ == '__main__':
    main()
