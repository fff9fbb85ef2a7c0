#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_silhouette.py tests/test_gpu_api.py -x -q 2>&1 | tail -5
bash tools/ab_multi.sh "c2 c3 c4 c5" ab_head.so ab_new.so 2>&1 | tee gpurun_out/ab_planA.log
