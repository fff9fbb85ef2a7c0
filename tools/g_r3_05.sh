#!/bin/bash
cd $GRAFT_REPO_ROOT
cp gendr_amd/libgendr_hip.so /tmp/keep.so
cp ab_fused.so gendr_amd/libgendr_hip.so
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_silhouette.py tests/test_gpu_fill.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -8
cp /tmp/keep.so gendr_amd/libgendr_hip.so
bash tools/ab_multi.sh "c2 c3 c4 c5" ab_split3.so ab_fused.so 2>&1 | tee gpurun_out/ab_fused.log
bash tools/ab_batches.sh c2 "2 8" ab_split3.so ab_fused.so 2>&1 | tee -a gpurun_out/ab_fused.log
