#!/bin/bash
# Everything the round's numbers come from, in one gpurun call:
#   gpurun --timeout 1700 -- 'bash profiles/run_all.sh r01_v5'
# kernel trace of the headline bench, HBM traffic counters (two passes), SQ counters (two passes), plain bench lines.
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/all_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
bash profiles/run_profile.sh $TAG c2 > $OUT/profile.log 2>&1
cp gpurun_out/prof_${TAG}_c2/c2_kernel_stats.csv $OUT/${TAG}_c2_kernel_stats.csv
cp gpurun_out/prof_${TAG}_c2/bench.json $OUT/${TAG}_c2_bench_under_rocprof.json
bash profiles/run_traffic.sh c2 > $OUT/traffic.log 2>&1
cp gpurun_out/traffic_c2/pmc_c2.json $OUT/pmc_c2.json
cp $OUT/pmc_c2.json profiles/pmc_c2.json                      # so that the bench lines below carry the traffic figure
bash profiles/run_pmc.sh ${TAG}_a "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" > $OUT/pmc_a.log 2>&1
bash profiles/run_pmc.sh ${TAG}_b "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" > $OUT/pmc_b.log 2>&1
for P in a b; do python tools/pmc_summary.py $(ls gpurun_out/pmc_${TAG}_$P/*counter_collection.csv | head -1); done > $OUT/${TAG}_c2_sq_counters.txt 2>&1
timeout 600 python bench.py --config c2 --steps 30 --warmup 5 2>/dev/null | grep '^{' > $OUT/${TAG}_c2_bench.json
for c in c3 c4 c5; do timeout 300 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' > $OUT/${TAG}_${c}_bench.json; done
ls -la $OUT
