#!/bin/bash
# The runtime-dispatch team kernel (light distributions x light aggregators) against the one-wave runtime-dispatch kernels at opt_shape.py's shape
cd $GRAFT_REPO_ROOT
for o in "dist_func=gaussian dist_scale=0.01" "dist_func=uniform dist_scale=0.03" "dist_func=laplace dist_scale=0.01 aggr_alpha_func=einstein" "dist_func=cubic_hermite dist_scale=0.05 aggr_alpha_func=max" "dist_func=gaussian dist_scale=0.0003 dist_squared=1"; do
  for team in -1 1; do
    python tools/shapebench.py 64 24 aggr_rgb_func=hard dist_eps=100 $o team=$team 2>&1 | tail -1
  done
done
