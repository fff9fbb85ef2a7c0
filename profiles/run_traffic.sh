#!/bin/bash
# HBM traffic of the bench kernels from the TCC counters, one counter per pass (FETCH_SIZE costs 3 TCC slots,
# WRITE_SIZE 2: they do not fit one pass).   gpurun --timeout 900 -- 'bash profiles/run_traffic.sh c2'
set -e
CFG=${1:-c2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/traffic_${CFG}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -o t -- \
      python $GRAFT_REPO_ROOT/bench.py --config $CFG --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $OUT/$C.log 2>&1
done
python $GRAFT_REPO_ROOT/tools/traffic_summary.py $OUT $CFG
