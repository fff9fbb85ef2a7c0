#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time python -m pytest tests -m gpu -x -q ) > gpurun_out/pytest_gpu.log 2>&1
tail -15 gpurun_out/pytest_gpu.log
( time python tests/gpu_report.py r03 ) > gpurun_out/report.log 2>&1
tail -8 gpurun_out/report.log | cut -c1-600
