#!/bin/bash
# kbench with alternative register budgets of the occupancy-capped kernels swapped in (diagnostic)
cp gendr_amd/libgendr_hip.so /tmp/full.so
for n in 75 66 76; do cp gpurun_ablate_w$n.so gendr_amd/libgendr_hip.so; echo "fwd/bwd waves=$n"; python tools/kbench.py 2>&1 | grep normal; done
cp /tmp/full.so gendr_amd/libgendr_hip.so; echo "65 (default)"; python tools/kbench.py | grep normal; python tools/kbench.py | grep normal
