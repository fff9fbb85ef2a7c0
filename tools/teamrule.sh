#!/bin/bash
# Where do the team kernels pay?  One-wave kernels (team=-1) against team kernels (team=1) over shapes around the automatic rule
# (gendr_capi.hip pick_team: at most 4096 tiles, cull radius >= 2 pixels):   bash tools/teamrule.sh > gpurun_out/teamrule.txt
cd $GRAFT_REPO_ROOT
O="dist_func=logistic aggr_rgb_func=hard dist_eps=100"
for shape in "64 8 0.01" "64 24 0.01" "64 64 0.01" "64 128 0.01" "64 24 0.005" "64 24 0.003" "64 24 0.03" "128 4 0.01" "128 16 0.01" "128 32 0.01" "256 1 0.003" "256 4 0.003" "256 8 0.01" "32 24 0.02"; do
  set -- $shape
  for team in -1 1; do
    python tools/shapebench.py $1 $2 $O dist_scale=$3 team=$team 2>&1 | tail -1
  done
done
