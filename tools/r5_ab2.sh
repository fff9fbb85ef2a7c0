#!/bin/bash
# round 5: increments on the geometry chain and the pair math, A/B inside one call (files at the repo root, built by tools/variant.sh):
#   nofilter.so  round-4 cover kernel                      cov_f.so   + tile-level edge filter
#   nobf.so      + lane-parallel mask-row unpack           bf.so      + branch-free region logic in point_to_face()
#   base         + five-operation double division in the uniform CDF (the library in gendr_amd/)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r5_ab2.log
bash tools/ab.sh nofilter.so cov_f.so nobf.so bf.so > $L 2>&1
echo "---- c3" >> $L
bash tools/ab_cfg.sh "--config c3 --iters 20 --modes normal" nofilter.so bf.so /tmp/base.so >> $L 2>&1
echo "---- c5" >> $L
bash tools/ab_cfg.sh "--config c5 --iters 10 --modes normal" nofilter.so /tmp/base.so >> $L 2>&1
echo "---- c4 batch 32" >> $L
bash tools/ab_cfg.sh "--config c4 --batch 32 --iters 10 --modes normal" nofilter.so /tmp/base.so >> $L 2>&1
python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_fuzz.py tests/test_gpu_callers.py tests/test_gpu_silhouette.py -x -q 2>&1 | tail -4 >> $L
python -m pytest tests/test_gpu_reference_pin.py -x -q -k product 2>&1 | tail -3 >> $L
grep -v amdgpu $L
