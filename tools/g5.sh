cd $GRAFT_REPO_ROOT
bash profiles/run_pmc.sh r02d_a "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" > /dev/null 2>&1
bash profiles/run_pmc.sh r02d_b "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" > /dev/null 2>&1
for P in a b; do python tools/pmc_summary.py $(ls gpurun_out/pmc_r02d_$P/*counter_collection.csv | head -1); done
