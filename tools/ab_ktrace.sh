#!/bin/bash
# Per-kernel durations (rocprofv3 kernel trace, tools/ktrace.sh) of several library builds on BASELINE config 2:
#   bash tools/ab_ktrace.sh a.so b.so ...      (files at the repo root; the geometry kernels' lines are printed)
cd $GRAFT_REPO_ROOT
cp gendr_amd/libgendr_hip.so /tmp/base.so
for f in "$@"; do
  cp $f gendr_amd/libgendr_hip.so
  echo "== $f"; bash tools/ktrace.sh ab_$(basename $f .so) --iters 30 --modes normal | cut -c1-60,96-
done
cp /tmp/base.so gendr_amd/libgendr_hip.so
