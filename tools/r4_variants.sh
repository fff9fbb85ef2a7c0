#!/bin/bash
# Round 4, build variants side by side on one box: kernel-level timing of default / fast at C2..C5, the fast variant's
# culled == all-pairs check, and the pin table (default and fast against the reference's own kernels).
#   gpurun --timeout 2400 -- 'bash tools/r4_variants.sh'
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for rep in 1 2; do
for v in default fast; do
  for c in c2 c3; do echo "== $v $c"; GENDR_VARIANT=$v python tools/kbench.py --config $c --modes normal --iters 30 2>&1 | grep normal; done
  echo "== $v c4 batch 32"; GENDR_VARIANT=$v python tools/kbench.py --config c4 --batch 32 --modes normal --iters 5 2>&1 | grep normal
  echo "== $v c5 batch 8"; GENDR_VARIANT=$v python tools/kbench.py --config c5 --batch 8 --modes normal --iters 8 2>&1 | grep normal
done; done
} > gpurun_out/r4_variants_kbench.txt 2>&1
cat gpurun_out/r4_variants_kbench.txt
GENDR_VARIANT=fast timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -x -k "culling" 2>&1 | tail -5 > gpurun_out/r4_fast_cull.txt
cat gpurun_out/r4_fast_cull.txt
timeout 1500 python tests/golden/make_pin_table.py > gpurun_out/r4_pin_table.log 2>&1
tail -8 gpurun_out/r4_pin_table.log
