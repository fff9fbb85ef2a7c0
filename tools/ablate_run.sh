#!/bin/bash
# runs kbench with the ablation builds swapped in (diagnostic): 1 = forward without phases B+C, 2 = forward without C,
# 3 = backward without phase B and the segment sums, 4 = backward without the segment sums / atomics
cp gendr_amd/libgendr_hip.so /tmp/full.so
for n in 1 2 3 4; do cp gpurun_ablate_$n.so gendr_amd/libgendr_hip.so; echo "ABLATE=$n"; python tools/kbench.py 2>&1 | grep normal; done
cp /tmp/full.so gendr_amd/libgendr_hip.so; echo FULL; python tools/kbench.py | grep normal
