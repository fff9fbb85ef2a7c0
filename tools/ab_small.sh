#!/bin/bash
# A/B of library builds at small batches (strong-scaling shares of C2): bash tools/ab_small.sh a.so b.so ...
cd $GRAFT_REPO_ROOT
cp gendr_amd/libgendr_hip.so /tmp/base.so
for rep in 1 2; do
for f in "$@"; do
  cp $f gendr_amd/libgendr_hip.so
  echo "== $f"
  for b in 8 16 32 64; do python tools/kbench.py --config c2 --batch $b --iters 40 2>&1 | grep "normal" | sed "s/^/b$b /"; done
done; done
cp /tmp/base.so gendr_amd/libgendr_hip.so
