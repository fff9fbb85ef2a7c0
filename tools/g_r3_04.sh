#!/bin/bash
cd $GRAFT_REPO_ROOT
cp ab_split.so gendr_amd/libgendr_hip.so
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_silhouette.py -x -q 2>&1 | tail -3
bash tools/ab_batches.sh c2 "1 2 4 8 16 32 64" ab_new.so ab_split.so 2>&1 | tee gpurun_out/ab_split.log
