#!/bin/bash
# round 5, verification of the conformant default build: selftest of the polynomial normal CDF, pin table (default / exact),
# cost at C2 / C3 / C5, the new -m gpu tests (callers, reference-arbitrated fuzz, forced loose-face list path)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r5_check1.log
: > $L
python -m pytest tests/test_gpu_exact_math.py -q -s -k normal 2>&1 | tail -6 >> $L
PIN_VARIANTS=default,exact python tests/golden/make_pin_table.py $PIN_ONLY > gpurun_out/r5_pin_full.log 2>&1
tail -4 gpurun_out/r5_pin_full.log >> $L
cp gpurun_out/pin_table.json gpurun_out/r5_pin_table.json
for cfg in c2 c3 c5; do
  for v in default exact; do
    echo "== $cfg $v" >> $L
    GENDR_VARIANT=$v python tools/kbench.py --config $cfg --iters 20 --modes normal 2>&1 | grep normal >> $L
  done
done
python -m pytest tests/test_gpu_callers.py tests/test_gpu_round3.py -x -q -k "script or silhouette or loose" 2>&1 | tail -15 >> $L
python -m pytest tests/test_gpu_fuzz.py -x -q -k "reference_kernels" 2>&1 | tail -60 >> $L
tail -100 $L
