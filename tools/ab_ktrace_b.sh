#!/bin/bash
# tools/ab_ktrace.sh at a given batch size: bash tools/ab_ktrace_b.sh <batch> a.so b.so ...
cd $GRAFT_REPO_ROOT
B=$1; shift
cp gendr_amd/libgendr_hip.so /tmp/base.so
for f in "$@"; do
  cp $f gendr_amd/libgendr_hip.so
  echo "== $f (batch $B)"; bash tools/ktrace.sh ab_$(basename $f .so) --batch $B --iters 30 --modes normal | cut -c1-60,96-
done
cp /tmp/base.so gendr_amd/libgendr_hip.so
