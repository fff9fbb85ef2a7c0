#!/bin/bash
# Regenerates the pin table (tests/golden/reference/pin_table.json) and runs the pin / variant tests against it.
#   gpurun --timeout 2400 -- 'bash tools/r4_pin.sh'
cd $GRAFT_REPO_ROOT
timeout 1500 python tests/golden/make_pin_table.py > gpurun_out/r4_pin_table.log 2>&1
grep -E "cases above|fast variant" gpurun_out/r4_pin_table.log
cp gpurun_out/pin_table.json tests/golden/reference/pin_table.json
timeout 1800 python -m pytest tests/test_gpu_reference_pin.py tests/test_gpu_fast_variant.py tests/test_gpu_exact_gradient.py -q -x 2>&1 | tail -15
