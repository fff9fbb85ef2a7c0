#!/bin/bash
# per-kernel durations (rocprofv3 kernel trace) of the full build and of the ablation builds gpurun_ablate_N.so:
#   1 = forward without phases B+C, 2 = forward without C, 3 = backward without phase B and the segment sums,
#   4 = backward without the segment sums / atomics, 5 = forward without the store loop of the unlisted tiles
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/ablate_${1:-x}
mkdir -p $OUT
cp gendr_amd/libgendr_hip.so /tmp/full.so
export TMPDIR=/tmp
for n in 0 1 2 3 4 5; do
  if [ $n -gt 0 ]; then [ -f gpurun_ablate_$n.so ] || continue; cp gpurun_ablate_$n.so gendr_amd/libgendr_hip.so; else cp /tmp/full.so gendr_amd/libgendr_hip.so; fi
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/v$n -o k -- python $GRAFT_REPO_ROOT/tools/kbench.py --iters 10 ${@:2} > $OUT/v$n.log 2>&1)
  echo "== ABLATE=$n"; grep normal $OUT/v$n.log
  python - <<PY
import csv,glob
f=glob.glob('$OUT/v$n/**/k_kernel_stats.csv', recursive=True)
for r in csv.DictReader(open(f[0])):
    if 'gendr' in r['Name']: print('   %-60s %9.1f us x %s' % (r['Name'][:60], float(r['AverageNs'])/1e3, r['Calls']))
PY
done
cp /tmp/full.so gendr_amd/libgendr_hip.so
