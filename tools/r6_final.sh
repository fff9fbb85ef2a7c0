#!/bin/bash
# round 6, closing call at the final kernels: the whole -m gpu suite, the parity report (gpurun_out/parity_r06.json), a second reference-arbitrated
# fuzz campaign (500 draws, seed 7: the coverage kernel -- which decides which pairs exist -- changed after the first one), the profiles of every
# config (profiles/run_all.sh) and the render kernels' phase tables.   `nofuzz`: without the campaign (a re-collection after a comment-only change).   Local first: bash tools/fwd_phases.sh build; bash tools/bwd_phases.sh build
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/ -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r06_gpu_suite.txt; cat gpurun_out/r06_gpu_suite.txt
python tests/gpu_report.py r06 > gpurun_out/parity_r06.log 2>&1; tail -2 gpurun_out/parity_r06.log | cut -c1-600
if [ "$1" != nofuzz ]; then python tools/fuzz_parity.py 500 7 ref > gpurun_out/r06_fuzz500_seed7_ref.log 2>&1; tail -3 gpurun_out/r06_fuzz500_seed7_ref.log; grep -c FAIL gpurun_out/r06_fuzz500_seed7_ref.log; fi
make -C tools/micro bin/valucal >/dev/null 2>&1
bash profiles/run_all.sh r06 > gpurun_out/run_all_r06.log 2>&1; tail -2 gpurun_out/run_all_r06.log
bash tools/fwd_phases.sh run > gpurun_out/fwd_phases.log 2>&1; python tools/fwd_phases.py gpurun_out/fwd_phases > gpurun_out/r06_c2_fwd_phases.txt; head -9 gpurun_out/r06_c2_fwd_phases.txt
bash tools/bwd_phases.sh run > gpurun_out/r06_c2_bwd_phases.txt 2>&1; tail -6 gpurun_out/r06_c2_bwd_phases.txt
