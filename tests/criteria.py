"""Acceptance criterion of the GPU parity tests (DESIGN.md "parity policy").

Target: 1e-5 relative against the fp32 oracle.  The restated arithmetic is ill-conditioned in places by
construction of the reference's formulas (e.g. 1 - exp(-exp(u)) for gumbel_min, 1 - y for gamma_rev,
Frank/Aczel-Alsina near alpha -> 0, saturated alpha with the 1e-6 guards): there a one-ulp difference
between glibc and the device libm moves the result by far more than 1e-5, and the fp32 and fp64
instantiations of the SAME oracle differ by 1e-3 .. O(1).  For those tensors the HIP path has to be at least
as close to the fp32 oracle as fp32 arithmetic itself is to fp64 (noise floor measured per case)."""
import numpy as np

import parity

TOL = 1e-5


RAW_GRAD_KEYS = ('grad_faces', 'grad_textures')


def noise_floor(fv, tex, image_size, opts, grad, oracle_f32=None):
    """fp32-vs-fp64 spread of the oracle itself.  `oracle_f32`: an fp32 oracle run of the same inputs the caller
    already has (parity.compare returns it), so that large frames are not evaluated twice."""
    a = oracle_f32 if oracle_f32 is not None else parity.run_oracle(fv, tex, image_size, opts, grad, np.float32)
    b = parity.run_oracle(fv.astype(np.float64), tex.astype(np.float64), image_size, opts,
                          None if grad is None else grad.astype(np.float64), np.float64)
    out = dict(rgba=parity.stats(a['rgba'], b['rgba']), aggrs=parity.stats(a['aggrs_info'], b['aggrs_info']))
    if grad is not None:
        out['grad_faces_cond'] = parity.stats(a['grad_faces'], b['grad_faces'], scale=b['abs_faces'])
        out['grad_textures_cond'] = parity.stats(a['grad_textures'], b['grad_textures'], scale=b['abs_textures'])
        out['grad_faces'] = parity.stats(a['grad_faces'], b['grad_faces'])
        out['grad_textures'] = parity.stats(a['grad_textures'], b['grad_textures'])
    return out


def check(res, noise, keys=('rgba', 'aggrs', 'grad_faces_cond', 'grad_textures_cond', 'grad_faces', 'grad_textures'), strict=False):
    """Returns a list of failure strings (empty = pass).  The raw gradient tensors (error relative to |reference
    element|, not to the sum of |contributions|) are held to the same rule on p99 and on the fraction above 1e-5;
    their MAX is not asserted: one element whose contributions cancel to 1e-6 of their size moves by O(1) relative
    with the summation order alone (the conditioned metric carries the max)."""
    bad = []
    for k in keys:
        if k not in res:
            continue
        e = res[k]
        if e['max_rel'] <= TOL:
            continue
        if strict:
            bad.append('%s: max_rel %.2e > %.0e (strict case)' % (k, e['max_rel'], TOL))
            continue
        n = noise[k]
        if e['p99_rel'] > max(TOL, 2 * n['p99_rel']):
            bad.append('%s: p99 %.2e vs fp32-noise p99 %.2e' % (k, e['p99_rel'], n['p99_rel']))
        if k not in RAW_GRAD_KEYS and e['max_rel'] > max(TOL, 2 * n['max_rel']):
            bad.append('%s: max %.2e vs fp32-noise max %.2e' % (k, e['max_rel'], n['max_rel']))
        # raw gradients: p99 <= 1e-5 already allows 1 % of the elements above it (summation order of cancelling sums)
        if e['frac_gt_1e5'] > max(1e-2 if k in RAW_GRAD_KEYS else 1e-3, 2 * n['frac_gt_1e5']):
            bad.append('%s: fraction>1e-5 %.2e vs fp32-noise %.2e' % (k, e['frac_gt_1e5'], n['frac_gt_1e5']))
    return bad


# cases whose forward is purely algebraic (no libm call before alpha): alpha must be bit-exact
def alpha_is_algebraic(name):
    return name.startswith(('uniform', 'hard', 'cubic'))
