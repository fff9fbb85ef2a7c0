"""Where does a build variant differ from the default one?  (diagnostic)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import parity, pin
variant = sys.argv[1] if len(sys.argv) > 1 else 'fast'
for name, opts, isz in pin.FULL[:int(sys.argv[2]) if len(sys.argv) > 2 else 2]:
    fv, tex = pin.full_inputs(name)
    g = pin.full_grad(isz)
    for cull in (1, 0):
        o = dict(opts, cull=cull)
        d = parity.run_hip(fv, tex, isz, o, g, variant='default')
        f = parity.run_hip(fv, tex, isz, o, g, variant=variant)
        err = np.abs(d['rgba'] - f['rgba'])
        err = np.where(np.isnan(err), 9.0, err)
        bad = np.argwhere(err > 1e-3)
        print(name, 'cull', cull, 'bad elements', len(bad), 'nan', int(np.isnan(f['rgba']).sum()), 'max', float(err.max()))
        for b in bad[:12]:
            b = tuple(b)
            print('   ', b, 'default', d['rgba'][b], variant, f['rgba'][b], 'aux', d['aggrs_info'][b[0], :, b[2], b[3]], f['aggrs_info'][b[0], :, b[2], b[3]])
        ys, xs = bad[:, 2], bad[:, 3]
        if len(bad):
            print('    channels', np.bincount(bad[:, 1], minlength=4), 'x%8', np.bincount(xs % 8, minlength=8), 'y%8', np.bincount(ys % 8, minlength=8))
