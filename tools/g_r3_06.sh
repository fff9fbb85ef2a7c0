#!/bin/bash
cd $GRAFT_REPO_ROOT
cp gendr_amd/libgendr_hip.so /tmp/keep.so
cp ab_fused2.so gendr_amd/libgendr_hip.so
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fill.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -3
cp /tmp/keep.so gendr_amd/libgendr_hip.so
bash tools/ktrace.sh "--config c2 --modes normal --iters 20" ab_fused2.so ab_fused2w6.so 2>&1 | grep -E "==|tile_cover|render_f"
bash tools/ktrace.sh "--config c2 --modes normal --iters 20 --batch 8" ab_fused2.so ab_fused2w6.so 2>&1 | grep -E "==|tile_cover"
bash tools/ktrace.sh "--config c5 --modes normal --iters 10" ab_fused2.so ab_fused2w6.so 2>&1 | grep -E "==|tile_cover"
bash tools/ktrace.sh "--config c4 --modes normal --iters 10 --batch 32" ab_split3.so ab_fused2.so ab_fused2w6.so 2>&1 | grep -E "==|tile_cover|cover_k|bin_fa"
