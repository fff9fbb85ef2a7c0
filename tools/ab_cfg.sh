#!/bin/bash
# A/B of library builds on one config's kernels: bash tools/ab_cfg.sh "<kbench args>" a.so b.so ...   (files at the repo root)
cd $GRAFT_REPO_ROOT
ARGS=$1; shift
cp gendr_amd/libgendr_hip.so /tmp/base.so
for rep in 1 2; do
for f in "$@"; do
  cp $f gendr_amd/libgendr_hip.so
  echo "== $f"; python tools/kbench.py $ARGS 2>&1 | grep -E "normal"
done; done
cp /tmp/base.so gendr_amd/libgendr_hip.so
