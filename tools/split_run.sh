#!/bin/bash
# kbench with alternative batch-split thresholds swapped in (diagnostic); 65 = never split
cp gendr_amd/libgendr_hip.so /tmp/full.so
for n in 1 4 16 65; do cp gpurun_ablate_s$n.so gendr_amd/libgendr_hip.so; echo "split_min=$n"; python tools/kbench.py 2>&1 | grep normal; done
cp /tmp/full.so gendr_amd/libgendr_hip.so; echo "8 (default)"; python tools/kbench.py | grep normal
