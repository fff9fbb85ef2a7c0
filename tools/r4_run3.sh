cd $GRAFT_REPO_ROOT
bash profiles/run_pmc.sh r04_c4_a "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY" --config c4 --modes normal --batch 32 > /dev/null 2>&1
bash profiles/run_pmc.sh r04_c4_b "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_CVT SQ_INSTS_VALU_MUL_F64 SQ_LDS_BANK_CONFLICT" --config c4 --modes normal --batch 32 > /dev/null 2>&1
for P in a b; do python tools/pmc_summary.py gpurun_out/pmc_r04_c4_$P render; done > gpurun_out/r04_c4_sq_now.txt
cat gpurun_out/r04_c4_sq_now.txt
rm -rf gpurun_out/pmc_r04_c4_a gpurun_out/pmc_r04_c4_b
