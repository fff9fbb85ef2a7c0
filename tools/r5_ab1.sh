#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r5_ab1.log
bash tools/ab.sh nohoist.so > $L 2>&1
cp gendr_amd/libgendr_hip.so /tmp/keep.so
cp gpurun_ablate_timers.so gendr_amd/libgendr_hip.so
python tools/phase_timers.py --config c2 >> $L 2>&1
python tools/phase_timers.py --config c3 >> $L 2>&1
cp /tmp/keep.so gendr_amd/libgendr_hip.so
grep -v amdgpu $L
