#!/bin/bash
# round 3, call 1: dump HIP outputs (default + exact variants) and kernel-level timing of both variants
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/dump_hip.py > gpurun_out/dump.log 2>&1
echo "dump rc $?" >> gpurun_out/dump.log
for cfg in c2 c3 c4 c5; do
  extra=""; [ $cfg = c4 ] && extra="--batch 32"
  for v in default exact; do
    echo "== $cfg $v" >> gpurun_out/variants.log
    GENDR_VARIANT=$v python tools/kbench.py --config $cfg --iters 20 --modes normal $extra >> gpurun_out/variants.log 2>&1
  done
done
python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench_head.log 2>&1
tail -3 gpurun_out/dump.log; cat gpurun_out/variants.log; cat gpurun_out/bench_head.log | cut -c1-600
