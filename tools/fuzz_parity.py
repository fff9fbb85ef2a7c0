"""Randomised HIP-vs-oracle check over shapes the fixed suites do not enumerate (run on the GPU box):
    python tools/fuzz_parity.py [n_cases] [seed] [smalleps] [ref]
Random batch size, face count (around the 64-face chunk boundaries), image size (odd sizes, sizes with empty 64x64
super-tiles), texture layout and option set; the acceptance rule of tests/criteria.py; also culled == all-pairs.
`ref` (needs oracle/_ref): additionally the `exact` build variant against the reference's OWN kernels wherever the reference's
two builds (contraction off / on) agree with each other to 1e-6 -- there nothing about the scene is ill-conditioned, the
variant calls the reference's libm functions, and a difference above 1e-5 is a structural defect (culling, coverage), not
noise: the detector that would have flagged round 4's case 255 without a second look."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import ctypes
import criteria, parity, scenes
from gendr_amd import _native

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
small_eps = 'smalleps' in sys.argv[3:]
with_ref = 'ref' in sys.argv[3:] and parity.reference_available() and parity.reference_available('render_fma')
small_rs = np.random.RandomState(77)           # (its own stream: the draw of everything else stays the plain campaign's)
names = [n for n, _ in scenes.OPTION_MATRIX]
bad = 0
n_team = 0
n_tcover = 0
for case in range(n_cases):
    name, opts = scenes.OPTION_MATRIX[rs.randint(len(names))]
    opts = scenes.independent_options(rs, opts)              # dist_func and aggr_alpha_func picked independently
    name = ('%s>%s/%s' % (name, opts.get('dist_func', 'uniform'), opts.get('aggr_alpha_func', 'probabilistic')))[:40]
    B = int(rs.choice([1, 2, 3, 5, 9]))
    nf = int(rs.choice([1, 2, 17, 63, 64, 65, 127, 130, 200]))
    isz = int(rs.choice([8, 13, 31, 64, 72, 100, 128, 136, 192, 200]))
    vertex = opts.get('texture_type') == 'vertex'
    T = 1 if vertex else int(rs.choice([1, 1, 4, 9]))
    scale = float(rs.choice([0.25, 0.5, 1.0]))             # small scenes leave empty super-tiles
    fv, tex = scenes.soup(B=B, nf=max(nf, 9), seed=int(rs.randint(1 << 30)), T=T, vertex_tex=vertex)
    fv, tex = fv[:, :nf].copy(), tex[:, :nf].copy()
    fv[..., :2] *= scale
    if 'T' in opts:
        opts.pop('T')
    opts['T'] = T
    # longer tails now and then: entries that cover most of a tile (dense entries, pixel mode, region tags: round 4)
    opts['dist_scale'] = float(opts.get('dist_scale', 1e-2)) * float(rs.choice([1.0, 1.0, 4.0, 10.0]))
    # `smalleps`: every case with a small dist_eps -- the reference's border test (kernel.cu:747), not the distribution's tail,
    # ends a face's reach there (the regime of the coverage kernel's box test: round 4's case 255)
    if small_eps:
        opts['dist_eps'] = float(small_rs.choice([1.0, 1.5, 3.0, 10.0, 30.0]))
    res, h, r = parity.compare(fv, tex, isz, opts)
    grad = np.random.RandomState(1).randn(B, 4, isz, isz).astype(np.float32)
    fails, _, _ = criteria.check_case(fv, tex, isz, opts, h, grad, oracle_f32=r)
    h2 = parity.run_hip(fv, tex, isz, dict(opts, cull=0), grad)
    same = all(np.array_equal(h[k], h2[k], equal_nan=True) for k in ('rgba', 'aggrs_info'))
    if not same:
        fails.append('culled != all-pairs')
    # team kernels (round 5; chosen automatically for calls of few tiles where the option set has one): the forward results must be
    # bit for bit those of the one-wave kernels (team = -1 switches them off), the gradients differ by the order of the atomics only
    h3 = parity.run_hip(fv, tex, isz, dict(opts, team=-1), grad)
    o_, extra_ = parity.split_options(opts)
    on_team = bool(_native.lib().gendr_uses_team(B, nf, T, ctypes.byref(parity.hip_params(isz, o_, extra_)), 0))
    n_team += on_team
    on_tcover = bool(_native.lib().gendr_uses_team_cover(B, nf, T, ctypes.byref(parity.hip_params(isz, o_, extra_))))
    n_tcover += on_tcover
    if not all(np.array_equal(h[k], h3[k], equal_nan=True) for k in ('rgba', 'aggrs_info')):
        fails.append('team != one-wave kernels (forward)')
    for k in ('grad_faces', 'grad_textures'):
        den = max(float(np.nanmax(np.abs(h3[k]))), 1e-30)
        if not float(np.nanmax(np.abs(h[k].astype(np.float64) - h3[k]))) <= 2e-5 * den:
            fails.append('team != one-wave kernels (%s, %.1e of the largest element)' % (k, float(np.nanmax(np.abs(h[k].astype(np.float64) - h3[k]))) / den))
    if with_ref and parity.split_options(opts)[1]['texel_mode'] == 0:        # (the reference has no clamped texel mode)
        r1 = parity.run_reference(fv, tex, isz, opts, grad, np.float32)
        r2 = parity.run_reference(fv, tex, isz, opts, grad, np.float32, variant='render_fma')
        he = parity.run_hip(fv, tex, isz, opts, grad, variant='exact')
        a1 = r1['rgba'].astype(np.float64).reshape(he['rgba'].shape)
        agree = np.abs(a1 - r2['rgba'].reshape(he['rgba'].shape)) <= 1e-6
        viol = agree & (np.abs(he['rgba'] - a1) > 1e-5)
        if viol.any():
            fails.append('STRUCTURAL: exact variant differs from the reference kernels on %d rgba elements where the reference\'s two builds agree (of %d), first %s'
                         % (int(viol.sum()), int(agree.sum()), tuple(int(v) for v in np.argwhere(viol)[0])))
    status = 'ok' if not fails else 'FAIL ' + '; '.join(fails)
    bad += bool(fails)
    print('%3d %-40s B=%d nf=%3d is=%3d T=%d scale=%.2f %s rgba max %.1e  %s' % (case, name, B, nf, isz, T, scale, ('team' if on_team else '    ') + ('+tc' if on_tcover else '   '), res['rgba']['max_rel'], status), flush=True)
print('%d / %d cases failed (%d draws rendered by the team kernels, %d with the coverage kernel in its team form)' % (bad, n_cases, n_team, n_tcover))
sys.exit(1 if bad else 0)
