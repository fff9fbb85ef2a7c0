#!/bin/bash
# PMC pass (counters only, own run): gpurun --timeout 900 -- 'bash profiles/run_pmc.sh <tag> "<counters>" [args to kbench]'
set -e
TAG=${1:-pmc}
CTRS=${2:-"SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"}
shift 2 || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT -o pmc -- python $GRAFT_REPO_ROOT/tools/kbench.py --iters 3 "$@" > $OUT/log.txt 2>&1 || (tail -20 $OUT/log.txt; exit 1)
ls $OUT
