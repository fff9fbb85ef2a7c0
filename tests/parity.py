"""HIP path (through the C ABI) vs the CPU oracle on the same inputs: shared by the -m gpu tests,
__graft_entry__.smoke() and the parity report.  Test infrastructure."""
import numpy as np
import torch

import oracle
from gendr_amd.functional import renderer as R

RENDER_KEYS = ('dist_func', 'dist_scale', 'dist_squared', 'dist_shape', 'dist_shift', 'dist_eps',
               'aggr_alpha_func', 'aggr_alpha_t_conorm_p', 'aggr_rgb_func', 'aggr_rgb_eps', 'aggr_rgb_gamma',
               'near', 'far', 'double_side', 'texture_type')
DEFAULTS = dict(dist_func='uniform', dist_scale=1e-2, dist_squared=False, dist_shape=None, dist_shift=None,
                dist_eps=1e4, aggr_alpha_func='probabilistic', aggr_alpha_t_conorm_p=None, aggr_rgb_func='softmax',
                aggr_rgb_eps=1e-3, aggr_rgb_gamma=1e-3, near=1, far=100, double_side=True, texture_type='surface')


def split_options(opts):
    o = dict(DEFAULTS)
    extra = dict(background=(0., 0., 0.), texel_mode=0, T=1, cull=1, deterministic=0, pool_entries_max=0, skip_unlisted_aux=0, pair_hints=0, loose_faces=0, team=0)
    for k, v in opts.items():
        if k in extra:
            extra[k] = v
        else:
            o[k] = v
    return o, extra


def hip_params(image_size, o, extra):
    p = R.make_params(image_size, list(extra['background']), *[o[k] for k in RENDER_KEYS])
    p.texel_mode = extra['texel_mode']
    p.cull = extra['cull']
    p.deterministic = extra['deterministic']
    p.pool_entries_max = extra['pool_entries_max']
    p.skip_unlisted_aux = extra['skip_unlisted_aux']
    p.pair_hints = extra['pair_hints']
    p.loose_faces = extra['loose_faces']
    p.team = extra['team']
    return p


def run_hip(fv, tex, image_size, opts, grad=None, device='cuda:0', variant=None):
    """fv [B,nf,3,3], tex [B,nf,T,3] numpy.  Returns dict of numpy arrays.  `variant`: build variant of the library
    ('default' / 'exact', gendr_amd/build.py); None = the process's active one."""
    if variant is not None:
        from gendr_amd import _native
        with _native.use_variant(variant):
            return run_hip(fv, tex, image_size, opts, grad, device)
    o, extra = split_options(opts)
    p = hip_params(image_size, o, extra)
    B, nf = fv.shape[:2]
    faces = torch.from_numpy(np.ascontiguousarray(fv, np.float32)).reshape(B, nf, 9).to(device)
    textures = torch.from_numpy(np.ascontiguousarray(tex, np.float32)).to(device)
    rgba, aux, rec = R.native_forward(faces, textures, p)
    out = dict(rgba=rgba.cpu().numpy(), aggrs_info=aux.cpu().numpy())
    if grad is not None:
        g = torch.from_numpy(np.ascontiguousarray(grad, np.float32)).to(device)
        gf, gt = R.native_backward(faces, textures, rgba, aux, rec, g, p)
        out['grad_faces'] = gf.cpu().numpy()
        out['grad_textures'] = gt.cpu().numpy()
    torch.cuda.synchronize()
    return out


def run_oracle(fv, tex, image_size, opts, grad=None, dtype=np.float32, threads=0, threshold_scale=1.0):
    """`threshold_scale`: sensitivity analysis (tests/criteria.py) -- the reference's two contribution thresholds
    (D <= 1e-6, kernel.cu:784; d^2 >= dist_eps * tau, :769) moved by that factor."""
    o, extra = split_options(opts)
    if threshold_scale != 1.0:
        o = dict(o, dist_eps=max(1.0, o['dist_eps'] * (1.0 + (threshold_scale - 1.0) * 1e-2)))
    oo = oracle.make_opts(image_size=image_size, texel_mode=extra['texel_mode'], num_threads=threads,
                          prob_threshold_scale=threshold_scale, **o)
    fwd = oracle.forward(fv, tex, oo, background=extra['background'], dtype=dtype)
    out = dict(rgba=fwd['rgba'], aggrs_info=fwd['aggrs_info'], faces_info=fwd['faces_info'])
    if grad is not None:
        gf, gt, af, at = oracle.backward(fwd, grad, oo, dtype=dtype)
        out.update(grad_faces=gf, grad_textures=gt, abs_faces=af, abs_textures=at)
    return out


def reference_available(name='render'):
    from oracle import ref_gpu
    return ref_gpu.available(name)


def require_reference(*names):
    """For fixtures of the reference-pin tests: oracle/_ref travels to the GPU box prebuilt, /root/reference does not exist there.
    Where the reference's sources ARE present (the build container) a missing oracle/_ref means oracle/build_ref.py failed --
    that must be a red result, not a silent skip of the only gate against the reference's own kernels."""
    import os
    import pytest
    missing = [n for n in (names or ('render',)) if not reference_available(n)]
    if not missing:
        return
    msg = 'oracle/_ref is not built (%s; python -m oracle.build_ref needs /root/reference)' % ', '.join(missing)
    if os.path.isdir('/root/reference') and os.environ.get('GENDR_ALLOW_MISSING_REF') != '1':
        pytest.fail(msg + ' -- the reference IS present here, so the build failed: fix it (GENDR_ALLOW_MISSING_REF=1 skips instead)')
    pytest.skip(msg)


def run_reference(fv, tex, image_size, opts, grad=None, dtype=np.float32, variant='render'):
    """The REFERENCE's own kernels on the GPU (oracle/_ref, built by oracle/build_ref.py from the reference's .cu file;
    launched by oracle/ref_gpu.py).  Same inputs / outputs as run_oracle.  texel_mode has no meaning here (the reference
    has one behaviour, the one texel_mode = 0 restates)."""
    from oracle import ref_gpu
    global REF_PIN_CALLS
    REF_PIN_CALLS += 1
    o, extra = split_options(opts)
    assert extra['texel_mode'] == 0, 'the reference has no clamped texel mode'
    p = hip_params(image_size, o, extra)
    g = None if grad is None else np.asarray(grad, dtype)
    out = ref_gpu.render(np.asarray(fv, dtype), np.asarray(tex, dtype), image_size, p, g, dtype, variant=variant,
                         background=extra['background'] if dtype == np.float64 else None)
    if int(p.dist_func) == 0 and out.get('grad_faces') is not None:
        # Heaviside backward: the reference multiplies UNINITIALISED locals by D' = 0 (kernel.cu:926-933, :1034-1051; DESIGN
        # quirk i) -- exactly 0 whenever the stale registers hold finite values, NaN when they do not, which depends on what
        # ran on the CU before (seen: the same case passes alone and fails inside the whole suite).  The defined value is 0.
        gf = out['grad_faces'].reshape(-1, 3, 3)
        xy = gf[:, :, :2]
        xy[~np.isfinite(xy)] = 0
    return out


GRAD_FLOOR = 1e-6
REF_PIN_CALLS = 0      # launches of the reference's own kernels in this process: tests/conftest.py prints it, so a silently skipped pin shows


def rel_error(got, ref, scale=None, floor=1e-10):
    """|got - ref| relative to max(|ref|, scale, 1e-6 of the tensor's largest magnitude, floor).  The absolute floor of
    the gradients (GRAD_FLOOR; inputs and upstream gradients are O(1), ordinary gradient elements 1 .. 40): where a pixel
    is covered to 1 - 1e-15, d alpha / d D = (1 - alpha) / (1 - D) is a difference of two doubles that agree in all but
    their last few bits -- a handful of ulps whose count follows the last bit of exp().  Gradient elements of 1e-15
    made of such terms alone (logistic tails under hard RGB, single-sided) differ by factors between two libms; they
    are held to 1e-9 * 1e-6 absolute instead."""
    got, ref = np.asarray(got, np.float64).reshape(ref.shape), np.asarray(ref, np.float64)
    d = np.abs(got - ref)
    d = np.where((got == ref) | (np.isnan(got) & np.isnan(ref)), 0.0, d)
    mag = np.abs(np.where(np.isfinite(ref), ref, 0.0))
    if scale is not None:
        mag = np.maximum(mag, np.asarray(scale, np.float64).reshape(ref.shape))
    den = np.maximum(mag, max(1e-6 * (float(mag.max()) if mag.size else 0.0), floor))
    return np.where(np.isnan(d), np.inf, d) / den


def stats(got, ref, scale=None):
    """max abs error, max / p99 relative error with the denominator floored at 1e-6 * max|ref|
    (or at `scale`, e.g. the sum of |contributions| of a gradient element), fraction above 1e-5."""
    got = np.asarray(got, np.float64).ravel()
    ref = np.asarray(ref, np.float64).ravel()
    if ref.size == 0:
        return dict(max_abs=0.0, max_rel=0.0, p99_rel=0.0, frac_gt_1e5=0.0, exact=1.0, n=0)
    d = np.abs(got - ref)
    d = np.where(np.isnan(got) & np.isnan(ref), 0.0, d)
    floor = 1e-6 * (np.nanmax(np.abs(ref)) if np.isfinite(np.nanmax(np.abs(ref))) else 1.0)
    den = np.maximum(np.abs(ref), floor)
    if scale is not None:
        den = np.maximum(den, np.asarray(scale, np.float64).ravel())
    rel = d / np.maximum(den, 1e-300)
    rel = np.where(np.isnan(rel), np.inf, rel)
    return dict(max_abs=float(np.nanmax(d)), max_rel=float(rel.max()), p99_rel=float(np.percentile(rel, 99)),
                frac_gt_1e5=float((rel > 1e-5).mean()), exact=float((got == ref).mean()), n=int(ref.size))


def compare(fv, tex, image_size, opts, seed=1, with_grad=True):
    rs = np.random.RandomState(seed)
    B = fv.shape[0]
    grad = rs.randn(B, 4, image_size, image_size).astype(np.float32) if with_grad else None
    h = run_hip(fv, tex, image_size, opts, grad)
    r = run_oracle(fv, tex, image_size, opts, grad)
    res = dict(rgba=stats(h['rgba'], r['rgba']), alpha=stats(h['rgba'][:, 3], r['rgba'][:, 3]),
               aggrs=stats(h['aggrs_info'], r['aggrs_info']))
    if with_grad:
        res['grad_faces'] = stats(h['grad_faces'], r['grad_faces'])
        res['grad_textures'] = stats(h['grad_textures'], r['grad_textures'])
        # conditioning-aware: error relative to the sum of |contributions| of each element
        res['grad_faces_cond'] = stats(h['grad_faces'], r['grad_faces'], scale=r['abs_faces'])
        res['grad_textures_cond'] = stats(h['grad_textures'], r['grad_textures'], scale=r['abs_textures'])
    return res, h, r


def fmt(name, res):
    parts = []
    for k, s in res.items():
        parts.append('%s[max_rel=%.2e p99=%.1e >1e-5:%.2e exact=%.3f]' % (k, s['max_rel'], s['p99_rel'], s['frac_gt_1e5'], s['exact']))
    return '%-26s ' % name + ' '.join(parts)
