import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, parity, scenes
fv, tex = scenes.sphere()
opts = dict(dist_func='wigner_semicircle', dist_scale=5e-2, aggr_alpha_func='hamacher', aggr_alpha_t_conorm_p=0.5)
isz, b, row, xi = 64, 0, 41, 15
for sel in ([32], [70], [32, 70], None):
    f = fv[b:b+1] if sel is None else fv[b:b+1, sel]
    t = tex[b:b+1] if sel is None else tex[b:b+1, sel]
    h = parity.run_hip(f, t, isz, dict(opts, texel_mode=1))
    o = parity.run_oracle(f, t, isz, dict(opts, texel_mode=1))
    print(sel, 'hip rgba', ['%.9g' % v for v in h['rgba'][0, :, row, xi]], 'aux', ['%.9g' % v for v in h['aggrs_info'][0, :, row, xi]])
    print(sel, 'o32 rgba', ['%.9g' % v for v in o['rgba'][0, :, row, xi]], 'aux', ['%.9g' % v for v in o['aggrs_info'][0, :, row, xi]])
