"""Replays fuzz cases (tools/fuzz_parity.py draw) and compares the gradients of several settings with the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import parity, scenes
small_eps = 'smalleps' in sys.argv            # the draw of `fuzz_parity.py n seed smalleps`
small_rs = np.random.RandomState(77)
n_cases, seed, want = int(sys.argv[1]), int(sys.argv[2]), [int(v) for v in sys.argv[3:] if v != 'smalleps']
rs = np.random.RandomState(seed)
names = [n for n, _ in scenes.OPTION_MATRIX]
for case in range(n_cases):
    name, opts = scenes.OPTION_MATRIX[rs.randint(len(names))]
    opts = scenes.independent_options(rs, opts)              # (the draw of tools/fuzz_parity.py since round 6)
    B = int(rs.choice([1, 2, 3, 5, 9])); nf = int(rs.choice([1, 2, 17, 63, 64, 65, 127, 130, 200])); isz = int(rs.choice([8, 13, 31, 64, 72, 100, 128, 136, 192, 200]))
    vertex = opts.get('texture_type') == 'vertex'
    T = 1 if vertex else int(rs.choice([1, 1, 4, 9]))
    scale = float(rs.choice([0.25, 0.5, 1.0]))
    fv, tex = scenes.soup(B=B, nf=max(nf, 9), seed=int(rs.randint(1 << 30)), T=T, vertex_tex=vertex)
    fv, tex = fv[:, :nf].copy(), tex[:, :nf].copy()
    fv[..., :2] *= scale
    opts.pop('T', None); opts['T'] = T
    opts['dist_scale'] = float(opts.get('dist_scale', 1e-2)) * float(rs.choice([1.0, 1.0, 4.0, 10.0]))
    if small_eps:
        opts['dist_eps'] = float(small_rs.choice([1.0, 1.5, 3.0, 10.0, 30.0]))
    if case not in want:
        continue
    grad = np.random.RandomState(1).randn(B, 4, isz, isz).astype(np.float32)
    o = parity.run_oracle(fv, tex, isz, opts, grad)
    print(case, name, B, nf, isz, T, opts)
    for label, extra in (('default', {}), ('no hints', dict(pair_hints=-1)), ('hints on', dict(pair_hints=1)), ('cull 0', dict(cull=0)), ('loose off', dict(loose_faces=-1))):
        h = parity.run_hip(fv, tex, isz, dict(opts, **extra), grad)
        for k in ('grad_faces', 'grad_textures'):
            d = np.abs(h[k].reshape(o[k].shape) - o[k]); i = np.unravel_index(int(d.argmax()), d.shape)
            print('   %-10s %-14s max |diff| %.3g at %s: hip %.6g oracle %.6g' % (label, k, d.max(), i, h[k].reshape(o[k].shape)[i], o[k][i]))
    # the forward result against everything that can arbitrate: the restatement in float and double, the reference's own kernels in
    # both builds (oracle/_ref, when present) -- an O(1) difference on a few pixels that the reference's two builds also show between
    # each other is the closest-point formula's ill-conditioning, not a defect of either
    h = parity.run_hip(fv, tex, isz, opts, grad)
    others = [('oracle f32', parity.run_oracle(fv, tex, isz, opts, grad, np.float32)), ('oracle f64', parity.run_oracle(fv, tex, isz, opts, grad, np.float64))]
    if parity.reference_available():
        others += [('ref kernels', parity.run_reference(fv, tex, isz, opts, grad, np.float32)), ('ref fma build', parity.run_reference(fv, tex, isz, opts, grad, np.float32, variant='render_fma'))]
    for label, r in others:
        for k in ('grad_faces', 'grad_textures'):
            if k in r:
                d = np.abs(h[k].astype(np.float64).reshape(r[k].shape) - r[k]); i = np.unravel_index(int(d.argmax()), d.shape)
                print('   %s hip vs %-14s max |diff| %.3g at %s (hip %.6g, other %.6g)' % (k, label, d.max(), i, h[k].reshape(r[k].shape)[i], r[k][i]))
        d = np.abs(h['rgba'].astype(np.float64) - r['rgba'].reshape(h['rgba'].shape)); i = np.unravel_index(int(d.argmax()), d.shape)
        print('   rgba hip vs %-14s max |diff| %.3g at %s (hip %.6g, other %.6g); elements > 1e-5: %d' % (label, d.max(), i, h['rgba'][i], r['rgba'].reshape(h['rgba'].shape)[i], int((d > 1e-5).sum())))
    for (la, ra), (lb, rb) in ((others[0], others[1]),) + (((others[2], others[3]), (others[2], others[0])) if len(others) > 2 else ()):
        d = np.abs(ra['rgba'].astype(np.float64) - rb['rgba'].astype(np.float64).reshape(ra['rgba'].shape))
        print('   rgba %-14s vs %-14s max |diff| %.3g; elements > 1e-5: %d' % (la, lb, d.max(), int((d > 1e-5).sum())))
