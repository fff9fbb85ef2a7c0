#!/bin/bash
# Team kernels against one-wave kernels at BASELINE config 4's option set (512^2, logistic, sigma 1e-2, softmax rgb) over batch sizes
cd $GRAFT_REPO_ROOT
O="dist_func=logistic aggr_rgb_func=softmax dist_scale=0.01"
for shape in "512 1" "512 4" "512 8" "512 32" "512 64" "256 32" "256 64"; do
  set -- $shape
  for team in -1 1; do
    python tools/shapebench.py $1 $2 $O team=$team 2>&1 | tail -1
  done
done
