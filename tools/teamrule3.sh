#!/bin/bash
# Team kernels at BASELINE config 2's option set (uniform, tau 1e-2, softmax rgb, 256^2) over the batch sizes of a strong-scaling run
cd $GRAFT_REPO_ROOT
for shape in "256 1" "256 2" "256 4" "256 8" "256 16" "256 32" "64 24" "64 256"; do
  set -- $shape
  for team in -1 1; do
    python tools/shapebench.py $1 $2 team=$team 2>&1 | tail -1
  done
done
