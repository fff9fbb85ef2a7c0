#!/bin/bash
# A/B of library builds at a given batch: bash tools/ab2.sh "<kbench args>" name1.so ...
cd $GRAFT_REPO_ROOT
cp gendr_amd/libgendr_hip.so /tmp/base.so
ARGS=$1; shift
for f in base "$@"; do
  if [ $f = base ]; then cp /tmp/base.so gendr_amd/libgendr_hip.so; else cp $f gendr_amd/libgendr_hip.so; fi
  echo "== $f"; python tools/kbench.py --iters 30 $ARGS 2>&1 | grep -E "normal"
done
cp /tmp/base.so gendr_amd/libgendr_hip.so
