"""Acceptance criterion of the HIP product against the CPU restatement (DESIGN.md "parity policy").  ELEMENT-WISE since
round 3.  (Against the REFERENCE's own kernels, run on the GPU, the product is held to a FLAT 1e-5 with an enumerated
exception table: tests/pin.py.  The rule below is what a comparison with a CPU libm needs.)

Target (BASELINE.json north_star): 1e-5 relative against the fp32 oracle.  Every element e of every compared tensor
has to satisfy

    |hip_e - o32_e|  <=  max( 1e-5 * scale_e ,  K_NOISE * noise_e ,  K_FLIP * flip_e )        K_NOISE = 8, K_FLIP = 2

    scale_e = max(|o32_e|, 1e-6 * max|o32|)              images (rgba, aggrs_info)
            = max(|o32_e|, sum of |contributions|_e)     gradients: a gradient element is a sum over (pixel, face)
                                                          pairs in an order that differs by design (and from run to
                                                          run: float atomics), so the error is measured against the
                                                          sum of the magnitudes that were added
    noise_e = what a different but equally valid libm does to the reference's formula at e: the change of o32 when every
              single-precision libm result (expf, powf, logf, erfcf, asinf, coshf, atanf) is moved ONE ulp -- 14 modes
              (JITTER_MODES): all up, all down, and up or down by six different hash bits of (result, argument) and their
              complements, so that neighbouring pairs also move against each other (`oracle.libm_jitter`: the GPU's libm
              and glibc are different, equally valid libms).  1 - exp(-e^u), 1 - y, Frank / Aczel-Alsina near alpha -> 0 and
              the saturated 1e-6 guards amplify that one ulp by 1e3..1e7.  The largest of the 14, taken as the maximum
              over the element's neighbourhood (the 3x3 pixels around it in every channel; the 9 / 3T components of its
              face): one element's change is a sample of the noise, not a bound on it -- hence also K_NOISE = 8 rather
              than 1: fourteen samples of a sum of signed one-ulp moves underestimate its worst case (measured in round 3,
              tools/criteria_study.py: K = 4 left 3 of 97 cases with a handful of violating elements, K = 8 none).
              The oracle's fp32-vs-fp64 spread is NOT part of the noise (USE_F64_SPREAD = False: it made the gradient
              bounds 10-500x wider than the device needs -- the device repeats the oracle's float operations, only its
              libm differs).

so an ill-conditioned element can no longer excuse a well-conditioned one (the round-2 rule compared tensor-wide
maxima and percentiles).

The reference's own skip thresholds get a term of their own.  A pair contributes iff D > 1e-6 (kernel.cu:784) and
d^2 < dist_eps * tau (:769); a fragment that sits within a few ulps of a threshold flips with the last bit of expf /
erfcf (device libm vs glibc), and with softmax RGB one such fragment alone sets the pixel's colour.  The oracle is
therefore evaluated twice more in fp32 with the PROBABILITY threshold moved by -10 % / +10 % (gumbel_min's
1 - exp(-e^u) is quantised in steps of 6 % at D = 1e-6) and dist_eps by -0.1 % / +0.1 % (`parity.run_oracle`,
threshold_scale: a distance threshold flips by rounding only, a few ulps), and

    flip_e = max(|o32(0.9) - o32|, |o32(1.1) - o32|)

taken over the pixel's channels resp. the face's components: what the pairs inside the band can move the element by
(the factor 2: a subset of flips with mixed signs).  It is magnitude-aware -- a threshold fragment with D = 1e-6 moves a
gradient by 1e-6 of its neighbours' contributions, so gradients stay held to 1e-5 almost everywhere.

The share of elements whose bound is wider than 1e-5 * scale is reported per tensor (`loosened`, split by cause) and
CAPPED: tests/golden/loosened_table.json holds, per (case, tensor), the share the rule loosens today (a property of the
oracle alone -- computed on the CPU by tests/golden/make_loosened_table.py); `loosened_failures` rejects a report whose
share exceeds the tabulated one (+ 2 points), so the rule cannot be widened silently.
"""
import json
import os

import numpy as np

import parity

TOL = 1e-5
K_NOISE = 8.0
THRESHOLD_BAND = 0.1
K_FLIP = 2.0
USE_F64_SPREAD = False
# oracle.libm_jitter modes: all results one ulp up / down, then up or down by six different hash bits of (result, argument)
# and their complements -- two pairs of one pixel move the same way in every hashed mode with probability 1/64
JITTER_MODES = (+1, -1, +2, -2, +3, -3, +4, -4, +5, -5, +6, -6, +7, -7)

IMAGE_KEYS = ('rgba', 'aggrs_info')
GRAD_KEYS = (('grad_faces', 'abs_faces'), ('grad_textures', 'abs_textures'))


def _nbr_max_image(a):
    """a [B,C,H,W] -> per pixel the maximum over all channels and the 3x3 window, broadcast back to a's shape."""
    m = a.max(1, keepdims=True)
    p = np.pad(m, ((0, 0), (0, 0), (1, 1), (1, 1)), mode='edge')
    out = np.zeros_like(m)
    for dy in range(3):
        for dx in range(3):
            out = np.maximum(out, p[:, :, dy:dy + m.shape[2], dx:dx + m.shape[3]])
    return np.broadcast_to(out, a.shape)


def _per_face_max(a):
    """a [B,nf,...] -> maximum over the face's components, broadcast back."""
    if a.size == 0:
        return a
    m = a.reshape(a.shape[0], a.shape[1], -1).max(-1)
    return np.broadcast_to(m.reshape(m.shape + (1,) * (a.ndim - 2)), a.shape)


def _f64(x):
    return np.asarray(x, np.float64)


def _absdiff(a, b):
    d = np.abs(a - b)
    d = np.where(np.isnan(a) & np.isnan(b), 0.0, d)
    d = np.where((a == b), 0.0, d)                       # equal infinities
    return np.where(np.isnan(d), np.inf, d)              # NaN against a number


def references(fv, tex, image_size, opts, grad, oracle_f32=None, n_jitter=len(JITTER_MODES)):
    """The four oracle evaluations the rule needs: fp32 (nominal), fp64, and fp32 with the skip thresholds at
    (1 -+ THRESHOLD_BAND)."""
    o32 = oracle_f32 if oracle_f32 is not None else parity.run_oracle(fv, tex, image_size, opts, grad, np.float32)
    o64 = parity.run_oracle(fv.astype(np.float64), tex.astype(np.float64), image_size, opts,
                            None if grad is None else grad.astype(np.float64), np.float64)
    lo = parity.run_oracle(fv, tex, image_size, opts, grad, np.float32, threshold_scale=1.0 - THRESHOLD_BAND)
    hi = parity.run_oracle(fv, tex, image_size, opts, grad, np.float32, threshold_scale=1.0 + THRESHOLD_BAND)
    import oracle
    jit = []
    for j in JITTER_MODES[:n_jitter]:
        with oracle.libm_jitter(j):
            jit.append(parity.run_oracle(fv, tex, image_size, opts, grad, np.float32))
    return dict(o32=o32, o64=o64, lo=lo, hi=hi, jit=jit)


def _one(hip, o32, o64, lo, hi, jit, abs_sum, image):
    hip, o32, o64, lo, hi = _f64(hip).reshape(o32.shape), _f64(o32), _f64(o64), _f64(lo), _f64(hi)
    n = int(o32.size)
    if n == 0:
        return dict(n=0, violations=0, loosened=0.0, loosened_by_noise=0.0, loosened_by_threshold=0.0, bound_rel_p50=0.0, bound_rel_p99=0.0, max_rel=0.0, p99_rel=0.0,
                    frac_gt_1e5=0.0, max_rel_tight=0.0, exact=1.0, worst=None)
    finite = np.abs(o32[np.isfinite(o32)])
    floor = 1e-6 * (finite.max() if finite.size else 1.0)
    scale = np.maximum(np.abs(np.where(np.isfinite(o32), o32, 0.0)), floor)
    if abs_sum is not None:
        scale = np.maximum(scale, _f64(abs_sum).reshape(o32.shape))
    scale = np.maximum(scale, 1e-300)
    nbr = _nbr_max_image if image else _per_face_max
    err = _absdiff(hip, o32)
    spread = _absdiff(o32, o64) if USE_F64_SPREAD else np.zeros_like(o32)
    for j in jit:
        spread = np.maximum(spread, _absdiff(_f64(j), o32))
    noise = nbr(spread)
    scale_n = nbr(scale)
    flip = np.maximum(_absdiff(lo, o32), _absdiff(hi, o32))
    flip = np.broadcast_to(flip.max(1, keepdims=True), flip.shape) if image else _per_face_max(flip)
    bound = np.maximum(TOL * scale, np.maximum(K_NOISE * noise, K_FLIP * flip))
    viol = err > bound
    rel = err / scale
    worst = None
    if viol.any():
        i = int(np.argmax(np.where(viol, err / bound, 0.0)))
        idx = tuple(int(v) for v in np.unravel_index(i, o32.shape))
        worst = dict(index=idx, hip=float(hip.flat[i]), o32=float(o32.flat[i]), o64=float(o64.flat[i]),
                     err=float(err.flat[i]), bound=float(bound.flat[i]))
    tight = rel[bound <= TOL * scale] if (bound <= TOL * scale).any() else np.zeros(1)
    return dict(n=n, violations=int(viol.sum()),
                # elements not held to 1e-5 of their neighbourhood's magnitude ...
                loosened=float((bound > TOL * scale_n).mean()),
                loosened_by_noise=float((K_NOISE * noise > TOL * scale_n).mean()),    # ... because fp32 itself is that noisy there
                loosened_by_threshold=float((K_FLIP * flip > TOL * scale_n).mean()),  # ... because a skip threshold decides them
                bound_rel_p50=float(np.percentile(bound / scale_n, 50)), bound_rel_p99=float(np.percentile(bound / scale_n, 99)),
                max_rel=float(rel.max()), p99_rel=float(np.percentile(rel, 99)), frac_gt_1e5=float((rel > TOL).mean()),
                max_rel_tight=float(tight.max()), exact=float((hip == o32).mean()), worst=worst)


def elementwise(hip, refs):
    """hip: dict of numpy arrays as parity.run_hip returns; refs: references().  Returns {tensor: report}."""
    out = {}
    o32, o64, lo, hi, jit = (refs[k] for k in ('o32', 'o64', 'lo', 'hi', 'jit'))
    for k in IMAGE_KEYS:
        if k in hip:
            out[k] = _one(hip[k], o32[k], o64[k], lo[k], hi[k], [j[k] for j in jit], None, True)
    for k, ak in GRAD_KEYS:
        if k in hip and k in o32:
            out[k] = _one(hip[k], o32[k], o64[k], lo[k], hi[k], [j[k] for j in jit], o32[ak], False)
    return out


def failures(report):
    bad = []
    for k, r in report.items():
        if r['violations']:
            w = r['worst']
            bad.append('%s: %d of %d elements violate |hip - o32| <= max(1e-5 scale, %g noise, %g flip); worst at %s: hip %.9g '
                       'o32 %.9g o64 %.9g, error %.3g > bound %.3g' % (k, r['violations'], r['n'], K_NOISE, K_FLIP, w['index'], w['hip'],
                                                                       w['o32'], w['o64'], w['err'], w['bound']))
    return bad


LOOSENED_TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'loosened_table.json')
LOOSENED_SLACK = 0.02
_loosened = None


def loosened_table():
    global _loosened
    if _loosened is None:
        try:
            with open(LOOSENED_TABLE) as f:
                _loosened = json.load(f)['cases']
        except (OSError, ValueError, KeyError):
            _loosened = {}
    return _loosened


def loosened_failures(key, report):
    """The share of elements NOT held to 1e-5 must not exceed what tests/golden/loosened_table.json records for the case
    (absent: 0) by more than LOOSENED_SLACK."""
    row = loosened_table().get(key, {})
    return ['%s %s: %.1f %% of the elements are not held to 1e-5, the table allows %.1f %% (+ %.0f points)'
            % (key, k, 100 * r['loosened'], 100 * row.get(k, 0.0), 100 * LOOSENED_SLACK)
            for k, r in report.items() if r['loosened'] > row.get(k, 0.0) + LOOSENED_SLACK]


def check_case(fv, tex, image_size, opts, hip, grad, oracle_f32=None, n_jitter=len(JITTER_MODES), key=None):
    """-> (failure strings, report, references).  `key`: the case's name in the loosened table (None: no ceiling check)."""
    refs = references(fv, tex, image_size, opts, grad, oracle_f32, n_jitter)
    rep = elementwise(hip, refs)
    bad = failures(rep)
    if key is not None:
        bad += loosened_failures(key, rep)
    return bad, rep, refs


# cases whose forward is purely algebraic (no libm call before alpha): alpha must be bit-exact
def alpha_is_algebraic(name):
    return name.startswith(('uniform', 'hard', 'cubic'))
