#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r5_check3.log
: > $L
python -m pytest tests/test_gpu_api.py tests/test_gpu_graph.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_c5.py -x -q 2>&1 | tail -8 >> $L
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep "^{" > gpurun_out/r5_bench_c2_b.json
python -c "
import json; d=json.load(open('gpurun_out/r5_bench_c2_b.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['kernel_ms']); print(d['extra'].get('caller_shapes')); print(d['extra'].get('eager'))" >> $L
cat $L
