#!/bin/bash
# round 6, first call: the new tests (cross product, option cache, double backward, independent fuzz picks), smoke, the headline bench
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=gpurun_out/r6_first.log
python __graft_entry__.py smoke > $L 2>&1
python -m pytest tests/test_gpu_cross_product.py tests/test_gpu_api.py -q -m gpu 2>&1 | tail -40 >> $L
python -m pytest tests/test_gpu_fuzz.py -q -m gpu 2>&1 | tail -40 >> $L
python bench.py > gpurun_out/r6_first_bench.json 2>> $L
tail -c 1500 gpurun_out/r6_first_bench.json >> $L
grep -v amdgpu.ids $L | tail -120
