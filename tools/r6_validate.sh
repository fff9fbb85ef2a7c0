#!/bin/bash
# round 6: the parity report (tests/gpu_report.py -> gpurun_out/parity_r06.json, with the plain-relative gradient statistics of VERDICT r5 item 7)
# and the round's ONE reference-arbitrated fuzz campaign (500 draws, dist_func and aggr_alpha_func picked independently) in one gpurun call
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tests/gpu_report.py r06 > gpurun_out/parity_r06.log 2>&1
tail -3 gpurun_out/parity_r06.log | cut -c1-1500
python tools/fuzz_parity.py 500 6 ref > gpurun_out/r06_fuzz500_seed6_ref.log 2>&1
tail -3 gpurun_out/r06_fuzz500_seed6_ref.log; grep -c FAIL gpurun_out/r06_fuzz500_seed6_ref.log
