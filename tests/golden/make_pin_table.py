"""Produces tests/golden/reference/pin_table.json ON THE GPU BOX (needs oracle/_ref and the built libraries):

    python tests/golden/make_pin_table.py            -> gpurun_out/pin_table.json (+ gpurun_out/pin_report.json, every case)
    cp gpurun_out/pin_table.json tests/golden/reference/pin_table.json

For every case of tests/pin.py (the option matrix x three scenes at 32^2 and BASELINE configurations C2..C5 at their image
size, one frame): the reference's own kernels (pin build and the contracted build), the HIP product (default and fast build
variants) and the CPU restatement (for the sums of |contributions| the gradient errors are measured against).  The table
holds, per build variant, the (case, tensor) pairs whose deviation from the reference kernels exceeds a flat 1e-5, with the
measured maximum / 99th percentile / share above 1e-5; and for the fast variant the pairs with elements outside the bracket
of the reference's two builds.  Data, not code: the pin tests read it (tests/pin.py: flat_failures, bracket_failures)."""
import json
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np

import parity
import pin


def one(key, fv, tex, isz, opts, grad, variants, report, table):
    r = parity.run_reference(fv, tex, isz, opts, grad, np.float32)
    rf = parity.run_reference(fv, tex, isz, opts, grad, np.float32, variant='render_fma')
    o = parity.run_oracle(fv, tex, isz, opts, grad, np.float32)
    entry = {}
    entry['reference_fma_vs_pin'] = pin.measure(rf, r, o['abs_faces'], o['abs_textures'])
    entry['restatement'] = pin.measure(o, r, o['abs_faces'], o['abs_textures'])
    for v in variants:
        h = parity.run_hip(fv, tex, isz, opts, grad, variant=v)
        m = pin.measure(h, r, o['abs_faces'], o['abs_textures'])
        entry[v] = m
        exc = pin.exceptions_of(m)
        if exc:
            table.setdefault(v, {})[key] = exc
        if v == 'fast':
            sp = pin.spread_failures(key, m, entry['reference_fma_vs_pin'])
            if sp:
                table.setdefault('fast_outside_spread', {})[key] = sp
            b = pin.bracket(h, r, rf, o['abs_faces'], o['abs_textures'])
            entry['fast_bracket'] = b
            out = {k: dict(violations=x['violations'], n=x['n'], worst_over_bound=x['worst_over_bound']) for k, x in b.items() if x['violations']}
            if out:
                table.setdefault('fast_bracket', {})[key] = out
    report[key] = entry
    line = [key]
    for v in variants:
        line.append('%s: ' % v + ' '.join('%s %.1e' % (k[:2] + k[-2:], m['max']) for k, m in entry[v].items()))
    if 'fast_bracket' in entry:
        line.append('bracket viol ' + ' '.join('%d' % x['violations'] for x in entry['fast_bracket'].values()))
    print(' | '.join(line), flush=True)


def main():
    from gendr_amd import build, _native
    # (PIN_VARIANTS=default,exact,<scratch name>: A/B libraries copied to gendr_amd/libgendr_hip_<name>.so)
    names = os.environ.get('PIN_VARIANTS', 'default,exact,fast').split(',')
    variants = [v for v in names if os.path.exists(_native.variant_path(v))]
    try:
        head = subprocess.check_output(['git', 'rev-parse', '--short', 'HEAD'], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        head = None
    table = dict(meta=dict(what='HIP product vs the reference\'s own kernels (oracle/_ref pin build) on an MI355X: (case, tensor) pairs above a '
                                'flat 1e-5, tests/pin.py', kernel_sha=build.source_sha(), head=head, date=time.strftime('%Y-%m-%d %H:%M:%S'),
                           variants=variants))
    report = {}
    only = sys.argv[1:]
    for scene in pin.SCENES:
        for name, opts in pin.MATRIX:
            key = pin.case_key(scene, name)
            if only and not any(s in key for s in only):
                continue
            fv, tex = pin.matrix_inputs(opts, scene)
            one(key, fv, tex, pin.MATRIX_SIZE, opts, pin.matrix_grad(fv, pin.MATRIX_SIZE), variants, report, table)
    for name, opts, isz in pin.FULL:
        if only and not any(s in name for s in only):
            continue
        fv, tex = pin.full_inputs(name)
        one(name, fv, tex, isz, opts, pin.full_grad(isz), variants, report, table)
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(table, open('gpurun_out/pin_table.json', 'w'), indent=1, sort_keys=True)
    json.dump(dict(meta=table['meta'], cases=report), open('gpurun_out/pin_report.json', 'w'))
    for v in variants:
        print(v, 'cases above a flat 1e-5:', len(table.get(v, {})), 'of', len(report))
    print('fast variant, cases with elements outside the element-wise bracket:', len(table.get('fast_bracket', {})),
          '; cases whose error quantiles leave the spread of the reference\'s two builds:', len(table.get('fast_outside_spread', {})))
    for k, v in table.get('fast_outside_spread', {}).items():
        print('   ', k, v)


if __name__ == '__main__':
    main()
