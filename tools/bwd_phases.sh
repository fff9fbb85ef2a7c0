#!/bin/bash
# The backward render kernel's vector instructions and lane fraction per phase, like tools/fwd_phases.sh:  build (local) / run (GPU box).
# GENDR_ABLATE: 0 the kernel; 3 no batch work (tile start-up, pixel inputs, entry walk, code list); 4 + pair math and partials (no sums); 8 + per-face sums (no atomics)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = build ]; then
  for n in 0 3 4 8; do GENDR_DEV_MIN=2 bash $ROOT/tools/devbuild.sh gpurun_ablate_b$n.so -DGENDR_ABLATE=$n & done; wait; ls -la $ROOT/gpurun_ablate_b*.so; exit 0
fi
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/bwd_phases; mkdir -p $OUT
cp gendr_amd/libgendr_hip.so /tmp/base.so
for n in 0 3 4 8; do
  cp gpurun_ablate_b$n.so gendr_amd/libgendr_hip.so
  python tools/kbench.py --iters 30 --modes normal > $OUT/time_$n.txt 2>&1
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES \
      --kernel-trace --output-format csv -d $OUT/pmc_$n -o pmc -- python $GRAFT_REPO_ROOT/tools/kbench.py --iters 3 --modes normal > $OUT/pmc_$n.log 2>&1)
  python tools/pmc_summary.py $OUT/pmc_$n render_backward > $OUT/ctr_$n.txt 2>&1
done
cp /tmp/base.so gendr_amd/libgendr_hip.so
rm -rf $OUT/pmc_*/*/*.db 2>/dev/null
python - <<'PY'
import re, os
d = 'gpurun_out/bwd_phases'
def ctr(n):
    out = {}
    for l in open(os.path.join(d, 'ctr_%d.txt' % n)):
        m = re.match(r'\s+(\S+)\s+mean (\S+)', l)
        if m: out[m.group(1)] = float(m.group(2))
    t = re.search(r'bwd\s+(\S+) ms', open(os.path.join(d, 'time_%d.txt' % n)).read())
    out['bwd_ms'] = float(t.group(1)) if t else float('nan')
    return out
c = {n: ctr(n) for n in (0, 3, 4, 8)}
def row(name, hi, lo):
    a, b = c[hi], (c[lo] if lo is not None else {})
    g = lambda k: a.get(k, 0) - b.get(k, 0)
    ai = g('SQ_ACTIVE_INST_VALU')
    print('%-52s %8.2f M valu %7.2f M lds %7.2f M salu  lane fraction %.3f  backward call %+.1f us' % (name, g('SQ_INSTS_VALU') / 1e6, g('SQ_INSTS_LDS') / 1e6, g('SQ_INSTS_SALU') / 1e6, g('SQ_THREAD_CYCLES_VALU') / (64 * ai) if ai else float('nan'), 1e3 * g('bwd_ms')))
print('backward render kernel, BASELINE config 2 (batch 64), per launch (tools/bwd_phases.sh)')
row('whole kernel', 0, None)
row('per tile outside the batches              (3)', 3, None)
row('pair math + partials                  (4 - 3)', 4, 3)
row('per-face sums                         (8 - 4)', 8, 4)
row('atomics                               (0 - 8)', 0, 8)
PY
