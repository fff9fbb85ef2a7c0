#!/bin/bash
# Evidence for the team kernels at opt_shape.py's shape -- 24 views of 64^2, logistic sigma 1e-2, hard RGB, dist_eps 100
# (/root/reference/experiments/opt_shape.py:134-145): kernel trace with the team kernels (automatic rule) and with the one-wave kernels
# (team=-1), and one SQ counter pass each.   gpurun --timeout 900 -- 'bash tools/team_profile.sh r05'   -> gpurun_out/<tag>_optshape_*
TAG=${1:-r05}
cd $GRAFT_REPO_ROOT
SHAPE="64 24 dist_func=logistic aggr_rgb_func=hard dist_eps=100 dist_scale=0.01"
OUT=gpurun_out/${TAG}_optshape_team.txt
: > $OUT
for mode in team onewave; do
  EX=""; [ $mode = onewave ] && EX="team=-1"
  echo "== kernel trace, $mode kernels ($SHAPE $EX)" >> $OUT
  bash tools/shapetrace.sh ${TAG}_$mode $SHAPE $EX > /dev/null 2>&1
  cat gpurun_out/shapetrace_${TAG}_$mode.txt >> $OUT
  grep "fwd" gpurun_out/shapetrace_${TAG}_$mode.log >> $OUT
  P=$GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_shape_$mode
  rm -rf $P; mkdir -p $P
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $P -o pmc -- python $GRAFT_REPO_ROOT/tools/shapebench.py $SHAPE $EX > $P/log.txt 2>&1)
  echo "== SQ counters, $mode kernels (useful lane fraction = SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU))" >> $OUT
  python tools/pmc_summary.py $P render_ >> $OUT
done
cat $OUT
