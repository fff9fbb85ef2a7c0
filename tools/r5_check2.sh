#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r5_check2.log
: > $L
python -m pytest tests/test_gpu_exact_math.py -q -s -k normal 2>&1 | tail -6 >> $L
python -m pytest tests/test_gpu_fuzz.py -q -k "reference_kernels" 2>&1 | tail -60 >> $L
python -m pytest tests/test_gpu_reference_pin.py -q -x -k "product" 2>&1 | tail -5 >> $L
tail -100 $L
