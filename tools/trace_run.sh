#!/bin/bash
# wave time lines of the given -DGENDR_TRACE=1 builds: bash tools/trace_run.sh "<wave_trace args>" a.so b.so
cd $GRAFT_REPO_ROOT
ARGS=$1; shift
cp gendr_amd/libgendr_hip.so /tmp/full.so
for f in "$@"; do cp $f gendr_amd/libgendr_hip.so; echo "===== $f"; python tools/wave_trace.py $ARGS 2>&1 | grep -v amdgpu.ids; done
cp /tmp/full.so gendr_amd/libgendr_hip.so
