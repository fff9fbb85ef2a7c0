#!/bin/bash
cd $GRAFT_REPO_ROOT
cp gendr_amd/libgendr_hip.so /tmp/keep.so
cp ab_sep.so gendr_amd/libgendr_hip.so
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fill.py tests/test_gpu_fuzz.py tests/test_gpu_silhouette.py tests/test_gpu_api.py -x -q 2>&1 | tail -3
cp /tmp/keep.so gendr_amd/libgendr_hip.so
bash tools/ktrace.sh "--config c2 --modes normal --iters 20" ab_sep.so 2>&1 | grep -E "==|gendr"
bash tools/ab_multi.sh "c2 c3 c4 c5" ab_head.so ab_sep.so 2>&1 | tee gpurun_out/ab_sep.log
bash tools/ab_batches.sh c2 "2 8 16" ab_head.so ab_sep.so 2>&1 | tee -a gpurun_out/ab_sep.log
