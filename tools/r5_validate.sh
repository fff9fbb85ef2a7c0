#!/bin/bash
# Round 5 validation on ONE box (VERDICT r4 hygiene 9: the pin table and the parity report come from the same gpurun call):
# pin table at the current kernels (default / exact: PIN_VARIANTS), the whole -m gpu suite, the parity report.
#   gpurun --timeout 3600 -- 'bash tools/r5_validate.sh'
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PIN_VARIANTS=default,exact timeout 1500 python tests/golden/make_pin_table.py > gpurun_out/r5_pin_table.log 2>&1
grep -E "cases above" gpurun_out/r5_pin_table.log
cp gpurun_out/pin_table.json gpurun_out/r5_pin_table.json
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r5_suite.txt
cat gpurun_out/r5_suite.txt
timeout 1500 python tests/gpu_report.py r05 > gpurun_out/r5_report.log 2>&1
tail -3 gpurun_out/r5_report.log
