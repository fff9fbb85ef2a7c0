#!/bin/bash
# A/B of library builds over several configs: bash tools/ab_multi.sh "c2 c3" a.so b.so ...   (files at the repo root)
cd $GRAFT_REPO_ROOT
CFGS=$1; shift
cp gendr_amd/libgendr_hip.so /tmp/base.so
for cfg in $CFGS; do
  extra=""; [ $cfg = c4 ] && extra="--batch 32"
  for rep in 1 2; do
    for f in "$@"; do
      cp $f gendr_amd/libgendr_hip.so
      echo -n "$cfg $f: "; python tools/kbench.py --config $cfg --modes normal --iters 30 $extra 2>&1 | grep -E "normal"
    done
  done
done
cp /tmp/base.so gendr_amd/libgendr_hip.so
