#!/bin/bash
# A/B of library builds over batch sizes of one config: bash tools/ab_batches.sh c2 "2 8 16 32 64" a.so b.so ...
cd $GRAFT_REPO_ROOT
CFG=$1; BATCHES=$2; shift 2
cp gendr_amd/libgendr_hip.so /tmp/base.so
for b in $BATCHES; do
  for rep in 1 2; do
    for f in "$@"; do
      cp $f gendr_amd/libgendr_hip.so
      echo -n "$CFG b=$b $f: "; python tools/kbench.py --config $CFG --modes normal --iters 40 --batch $b 2>&1 | grep -E "normal"
    done
  done
done
cp /tmp/base.so gendr_amd/libgendr_hip.so
