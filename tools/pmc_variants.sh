#!/bin/bash
# PMC counters for several library builds on one config: bash tools/pmc_variants.sh c2 "<counters>" a.so b.so ...
cd $GRAFT_REPO_ROOT
CFG=$1; CTRS=$2; shift 2
cp gendr_amd/libgendr_hip.so /tmp/base.so
export TMPDIR=/tmp
for f in "$@"; do
  cp $f gendr_amd/libgendr_hip.so
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcv_$(basename $f .so)
  rm -rf $OUT; mkdir -p $OUT
  ( cd /tmp && rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT -o pmc -- python $GRAFT_REPO_ROOT/tools/kbench.py --config $CFG --modes normal --iters 3 > $OUT/log.txt 2>&1 )
  echo "=== $f"; python tools/pmc_summary.py $OUT render_
done
cp /tmp/base.so gendr_amd/libgendr_hip.so
