#!/bin/bash
# A/B of library builds on the C2 bench kernels: bash tools/ab.sh name1.so name2.so ...   (files at the repo root)
cd $GRAFT_REPO_ROOT
cp gendr_amd/libgendr_hip.so /tmp/base.so
for rep in 1 2; do
for f in base "$@"; do
  if [ $f = base ]; then cp /tmp/base.so gendr_amd/libgendr_hip.so; else cp $f gendr_amd/libgendr_hip.so; fi
  echo "== $f"; python tools/kbench.py --iters 30 2>&1 | grep -E "normal"
done; done
cp /tmp/base.so gendr_amd/libgendr_hip.so
