#!/bin/bash
# batch sweep (tools/batch_sweep.py) of several library builds: bash tools/sweep_ab.sh a.so b.so
cd $GRAFT_REPO_ROOT
cp gendr_amd/libgendr_hip.so /tmp/full.so
for f in "$@"; do cp $f gendr_amd/libgendr_hip.so; echo "===== $f"; python tools/batch_sweep.py c2 tmp 2>&1 | grep -E "batch|^\{"; done
cp /tmp/full.so gendr_amd/libgendr_hip.so
