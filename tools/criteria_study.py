"""Offline study of the element-wise acceptance rule (tests/criteria.py) on dumped HIP outputs (tools/dump_hip.py)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import parity, criteria
from dump_hip import cases

def main():
    only = sys.argv[1:]
    tot = 0
    for name, fv, tex, isz, opts in cases(full=()):
        if only and not any(o in name for o in only): continue
        z = np.load(os.path.join(ROOT, 'gpurun_out', 'hipdump', name + '.npz'))
        grad = np.random.RandomState(1).randn(fv.shape[0], 4, isz, isz).astype(np.float32)
        refs = criteria.references(fv, tex, isz, opts, grad)
        line = [name]
        for v in ('default', 'exact'):
            hip = dict(rgba=z['default__rgba'], aggrs_info=z['default__aggrs_info'], grad_faces=z[v+'__grad_faces'], grad_textures=z[v+'__grad_textures'])
            rep = criteria.elementwise(hip, refs)
            bad = criteria.failures(rep)
            tot += len(bad)
            line.append(v + ': ' + ' '.join('%s[v%d n%.2f t%.2f b50 %.0e b99 %.0e e99 %.0e]' % (k[:6], r['violations'], r['loosened_by_noise'], r['loosened_by_threshold'], r['bound_rel_p50'], r['bound_rel_p99'], r['p99_rel']) for k, r in rep.items() if (v == 'default' or k.startswith('grad'))))
            for b in bad: line.append('\n    !! ' + b)
        print(' | '.join(line), flush=True)
    print('failures', tot)
main()
