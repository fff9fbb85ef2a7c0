#!/bin/bash
# Round 4 validation on one box: pin table at the current kernels, the whole -m gpu suite, the parity report.
#   gpurun --timeout 3600 -- 'bash tools/r4_validate.sh'
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python tests/golden/make_pin_table.py > gpurun_out/r4_pin_table.log 2>&1
grep -E "cases above|fast variant" gpurun_out/r4_pin_table.log
cp gpurun_out/pin_table.json tests/golden/reference/pin_table.json
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r4_suite.txt
cat gpurun_out/r4_suite.txt
timeout 1500 python tests/gpu_report.py r04 > gpurun_out/r4_report.log 2>&1
tail -3 gpurun_out/r4_report.log
