#!/bin/bash
# kbench with alternative register budgets of the occupancy-capped forward kernel swapped in (diagnostic)
cp gendr_amd/libgendr_hip.so /tmp/full.so
for n in 5; do cp gpurun_ablate_f$n.so gendr_amd/libgendr_hip.so; echo "fwd waves=$n"; python tools/kbench.py 2>&1 | grep -E "normal|offscreen"; done
cp /tmp/full.so gendr_amd/libgendr_hip.so; echo "6 (default)"; python tools/kbench.py | grep -E "normal|offscreen"
