#!/bin/bash
# where the forward render kernel's vector instructions and its idle lanes are, phase by phase.
# Local (build container):   bash tools/fwd_phases.sh build      -> gpurun_ablate_{0,1,2,5,6,7}.so at the repo root (C2's kernels only, 15 s each)
# GPU box:                   bash tools/fwd_phases.sh run        -> gpurun_out/fwd_phases/  (one rocprofv3 --pmc pass + one plain timing per build)
# then                       python tools/fwd_phases.py gpurun_out/fwd_phases > profiles/r06_c2_fwd_phases.txt
# Ablations of render_forward_body (GENDR_ABLATE): 0 the kernel; 5 without the fill of unlisted tiles; 1 no batch body (tile start-up, list
# building, epilogue); 6 + gathers and barycentrics; 7 + distance and CDF; 2 all of phase B, no fold.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = build ]; then
  for n in 0 1 2 5 6 7; do GENDR_DEV_MIN=2 bash $ROOT/tools/devbuild.sh gpurun_ablate_$n.so -DGENDR_ABLATE=$n & done; wait; ls -la $ROOT/gpurun_ablate_*.so; exit 0
fi
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/fwd_phases; mkdir -p $OUT
cp gendr_amd/libgendr_hip.so /tmp/base.so
for n in 0 1 2 5 6 7; do
  cp gpurun_ablate_$n.so gendr_amd/libgendr_hip.so
  python tools/kbench.py --iters 30 --modes normal > $OUT/time_$n.txt 2>&1
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES \
      --kernel-trace --output-format csv -d $OUT/pmc_$n -o pmc -- python $GRAFT_REPO_ROOT/tools/kbench.py --iters 3 --modes normal > $OUT/pmc_$n.log 2>&1)
  python tools/pmc_summary.py $OUT/pmc_$n render_forward > $OUT/ctr_$n.txt 2>&1
done
cp /tmp/base.so gendr_amd/libgendr_hip.so
rm -rf $OUT/pmc_*/*/*.db 2>/dev/null
cat $OUT/time_*.txt; cat $OUT/ctr_0.txt
