cd $GRAFT_REPO_ROOT
bash profiles/run_pmc.sh r02e_c "SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_CVT SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F32" > /dev/null 2>&1
bash profiles/run_pmc.sh r02e_d "SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_VALU_ADD_F64 SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES" > /dev/null 2>&1
bash profiles/run_pmc.sh r02e_e "TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES" > /dev/null 2>&1
bash profiles/run_pmc.sh r02e_f "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32" > /dev/null 2>&1
for P in c d e f; do python tools/pmc_summary.py $(ls gpurun_out/pmc_r02e_$P/*counter_collection.csv | head -1) | grep -A12 "render_\|cover"; done
