#!/bin/bash
# Kernel-trace profile of the headline bench (run on the GPU box through gpurun):
#   gpurun --timeout 900 -- 'bash profiles/run_profile.sh r01 c2'
# Writes gpurun_out/prof_<tag>/ ; the *_kernel_stats.csv summary is then copied by hand into profiles/.
set -e
TAG=${1:-r01}
CFG=${2:-c2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_${CFG}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $CFG -- \
    python $GRAFT_REPO_ROOT/bench.py --config $CFG --steps 20 --warmup 3 --no-cpu-baseline --no-extra > $OUT/bench.log 2>&1
grep '^{' $OUT/bench.log > $OUT/bench.json || true
ls $OUT
