"""Phase table of the forward render kernel from the ablated builds (tools/fwd_phases.sh): vector instructions, active lane-slots and
the useful lane fraction SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU) per phase, by subtraction.   python tools/fwd_phases.py <dir>"""
import re, sys, os
d = sys.argv[1]
def ctr(n):
    out = {}
    for l in open(os.path.join(d, 'ctr_%d.txt' % n)):
        m = re.match(r'\s+(\S+)\s+mean (\S+)', l)
        if m: out[m.group(1)] = float(m.group(2))
    t = re.search(r'fwd\s+(\S+) ms', open(os.path.join(d, 'time_%d.txt' % n)).read())
    out['fwd_ms'] = float(t.group(1)) if t else float('nan')
    return out
c = {n: ctr(n) for n in (0, 1, 2, 5, 6, 7)}
def row(name, hi, lo):
    a, b = c[hi], (c[lo] if lo is not None else {})
    g = lambda k: a.get(k, 0) - b.get(k, 0)
    iv, tc, ai = g('SQ_INSTS_VALU'), g('SQ_THREAD_CYCLES_VALU'), g('SQ_ACTIVE_INST_VALU')
    print('%-58s %8.2f M valu  %7.2f M lds  %7.2f M salu  lane fraction %.3f   forward phase %+.1f us' % (
        name, iv / 1e6, g('SQ_INSTS_LDS') / 1e6, g('SQ_INSTS_SALU') / 1e6, tc / (64 * ai) if ai else float('nan'), 1e3 * g('fwd_ms')))
print('forward render kernel, BASELINE config 2 (batch 64), per launch; counters of one rocprofv3 --pmc pass per ablated build, by subtraction;')
print('"forward phase" = setup + binning + coverage + ordering + render launch (HIP events, tools/kbench.py), difference between the two builds')
row('whole kernel', 0, None)
row('fill of the unlisted tiles                  (0 - 5)', 0, 5)
row('per tile outside the batches               (1)', 1, None)
row('phase B: gathers + barycentrics            (6 - 1)', 6, 1)
row('phase B: distance + CDF + skip tests       (7 - 6)', 7, 6)
row('phase B: clip, depth, colour, hints        (2 - 7)', 2, 7)
row('phase C: per-pixel fold                    (0 - 2)', 0, 2)
for n in sorted(c): print('  build %d: %s' % (n, {k: (round(v / 1e6, 3) if k != 'fwd_ms' else v) for k, v in sorted(c[n].items())}))
