#!/bin/bash
# PMC counters of one kbench invocation with the in-tree library: bash tools/pmc_one.sh "<counters>" "<kbench args>" [kernel substring]
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_one; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $OUT -o pmc -- python $GRAFT_REPO_ROOT/tools/kbench.py $2 > $OUT/log.txt 2>&1 )
python tools/pmc_summary.py $OUT $3
