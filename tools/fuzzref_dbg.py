"""Replays seeds of tests/test_gpu_fuzz.py::test_builds_agree_with_the_reference_kernels_wherever_its_two_builds_agree and prints, for every
violating element class, how the HIP product (default / exact / cull=0), the reference's two builds and the CPU restatement relate.
    python tools/fuzzref_dbg.py 5 6 13 ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import parity
import test_gpu_fuzz as F

for seed in [int(v) for v in sys.argv[1:]]:
    rs = np.random.RandomState(9000 + seed)
    name, opts, fv, tex, isz = F._draw(rs)
    opts['dist_scale'] = float(opts.get('dist_scale', 1e-2)) * float(rs.choice([1.0, 1.0, 4.0, 10.0]))
    if seed & 1:
        opts['dist_eps'] = float(rs.choice(F.EPS_REGIMES))
    if parity.split_options(opts)[1]['texel_mode'] != 0:
        opts['texel_mode'] = 0
    grad = np.random.RandomState(1).randn(fv.shape[0], 4, isz, isz).astype(np.float32)
    r1 = parity.run_reference(fv, tex, isz, opts, grad, np.float32)
    r2 = parity.run_reference(fv, tex, isz, opts, grad, np.float32, variant='render_fma')
    o = parity.run_oracle(fv, tex, isz, opts, grad, np.float32)
    print('seed', seed, name, opts, fv.shape, isz)
    runs = {'default': parity.run_hip(fv, tex, isz, opts, grad, variant='default'), 'exact': parity.run_hip(fv, tex, isz, opts, grad, variant='exact'),
            'cull0': parity.run_hip(fv, tex, isz, dict(opts, cull=0), grad, variant='default'), 'oracle': o}
    for k in ('rgba', 'aggrs_info'):
        a = np.asarray(r1[k], np.float64); b = np.asarray(r2[k], np.float64).reshape(a.shape)
        ok = np.isfinite(a) & np.isfinite(b) & (np.abs(a - b) <= 1e-6)
        for lab, h in runs.items():
            g = np.asarray(h[k], np.float64).reshape(a.shape)
            d = np.abs(g - a)
            viol = ok & ~(d <= 1e-5 * np.maximum(1.0, np.abs(a)))
            line = '  %-10s %-8s viol %5d of %d agreeing (%d elements); bit-identical to ref %.4f' % (k, lab, int(viol.sum()), int(ok.sum()), a.size, float((g == a).mean()))
            if viol.any():
                i = tuple(int(v) for v in np.argwhere(viol)[0])
                line += '  first %s: got %.9g ref %.9g ref_fma %.9g oracle %.9g' % (i, g[i], a[i], b[i], np.asarray(o[k]).reshape(a.shape)[i])
                ch = np.bincount(np.argwhere(viol)[:, 1], minlength=a.shape[1])
                line += ' per channel %s' % ch.tolist()
            print(line)
    for k, ak in (('grad_faces', 'abs_faces'), ('grad_textures', 'abs_textures')):
        sc = np.maximum(np.asarray(o[ak], np.float64), parity.GRAD_FLOOR)
        a = np.asarray(r1[k], np.float64).reshape(sc.shape); b = np.asarray(r2[k], np.float64).reshape(sc.shape)
        ok = np.isfinite(a) & np.isfinite(b) & (np.abs(a - b) <= 1e-6 * np.maximum(sc, np.abs(a)))
        for lab, h in runs.items():
            g = np.asarray(h[k], np.float64).reshape(sc.shape)
            rel = np.abs(g - a) / np.maximum(sc, np.abs(a))
            viol = ok & ~(rel <= 1e-5)
            line = '  %-13s %-8s viol %5d of %d agreeing (%d elements); max rel on agreeing %.3g' % (k, lab, int(viol.sum()), int(ok.sum()), a.size, float(rel[ok].max()) if ok.any() else 0.0)
            if viol.any():
                i = tuple(int(v) for v in np.argwhere(viol)[0])
                line += '  first %s: got %.9g ref %.9g ref_fma %.9g oracle %.9g scale %.3g' % (i, g[i], a[i], b[i], np.asarray(o[k]).reshape(sc.shape)[i], sc[i])
            print(line)
