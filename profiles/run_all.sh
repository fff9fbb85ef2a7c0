#!/bin/bash
# Everything a round's numbers come from, in one gpurun call:
#   gpurun --timeout 2400 -- 'bash profiles/run_all.sh r04'
# For EVERY BASELINE config (c2 .. c5): kernel trace of bench.py (rocprofv3 --kernel-trace --stats), HBM traffic
# counters (FETCH_SIZE and WRITE_SIZE in separate passes), two SQ counter passes on the native calls, and the plain bench
# line.  Results land in gpurun_out/all_<tag>/ ; copy what is to be kept into profiles/.
TAG=${1:-r05}
CFGS=${2:-"c2 c3 c4 c5"}
OUT=$GRAFT_REPO_ROOT/gpurun_out/all_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
# calibration of the VALU-occupation figure: a kernel of known occupation under the counters of pass b (tools/micro/valucal.hip)
CAL=$GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_cal
mkdir -p $CAL
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $CAL -o pmc -- $GRAFT_REPO_ROOT/tools/micro/bin/valucal > $CAL/log.txt 2>&1)
for c in $CFGS; do
  case $c in c4) EX="--batch 32"; SQB=32;; c5) EX="--batch 8"; SQB=8;; *) EX=""; SQB=64;; esac      # SQ passes (tools/kbench.py) at the single-GPU share of C4 / a quarter of C5; the traffic passes run bench.py at the config's batch -- both batches are recorded in pmc_<config>.json
  bash profiles/run_profile.sh $TAG $c > $OUT/profile_$c.log 2>&1
  cp gpurun_out/prof_${TAG}_$c/${c}_kernel_stats.csv $OUT/${TAG}_${c}_kernel_stats.csv
  cp gpurun_out/prof_${TAG}_$c/bench.json $OUT/${TAG}_${c}_bench_under_rocprof.json
  bash profiles/run_traffic.sh $c > $OUT/traffic_$c.log 2>&1
  cp gpurun_out/traffic_$c/pmc_$c.json $OUT/pmc_$c.json
  bash profiles/run_pmc.sh ${TAG}_${c}_a "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY" --config $c --modes normal $EX > $OUT/pmc_a_$c.log 2>&1
  bash profiles/run_pmc.sh ${TAG}_${c}_b "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_CVT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" --config $c --modes normal $EX > $OUT/pmc_b_$c.log 2>&1
  for P in a b; do python tools/pmc_summary.py gpurun_out/pmc_${TAG}_${c}_$P gendr; done > $OUT/${TAG}_${c}_sq_counters.txt 2>&1
  # kernel-source hash + VALU occupation into the traffic summary (bench.py quotes it only while the hash matches)
  python tools/pmc_finalize.py $OUT/pmc_$c.json gpurun_out/pmc_${TAG}_${c}_b $OUT/${TAG}_${c}_kernel_stats.csv $SQB $CAL >> $OUT/traffic_$c.log 2>&1
  cp $OUT/pmc_$c.json profiles/pmc_$c.json
done
python tools/batch_sweep.py c2 $TAG > $OUT/batch_sweep.log 2>&1
cp gpurun_out/${TAG}_c2_batch_sweep.json $OUT/ 2>/dev/null
cp gpurun_out/${TAG}_c2_batch_sweep.json profiles/ 2>/dev/null
timeout 900 python bench.py --config c2 --steps 30 --warmup 5 2>/dev/null | grep '^{' > $OUT/${TAG}_c2_bench.json
for c in c3 c4 c5; do [[ " $CFGS " == *" $c "* ]] && timeout 600 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' > $OUT/${TAG}_${c}_bench.json; done
ls -la $OUT
