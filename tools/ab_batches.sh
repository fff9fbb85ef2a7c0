#!/bin/bash
# A/B of library builds on the C2 kernels at several batch sizes (the strong-scaling shares): bash tools/ab_batches.sh "1 8 16 64" a.so b.so ...
cd $GRAFT_REPO_ROOT
BATCHES=$1; shift
cp gendr_amd/libgendr_hip.so /tmp/base.so
for rep in 1 2; do
for f in "$@"; do
  cp $f gendr_amd/libgendr_hip.so
  for b in $BATCHES; do echo "== $f batch $b: $(python tools/kbench.py --iters 40 --batch $b --modes normal 2>&1 | grep normal)"; done
done; done
cp /tmp/base.so gendr_amd/libgendr_hip.so
