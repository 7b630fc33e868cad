#!/bin/bash
# SQ counters of the shipped decode kernel (both instantiations together), separate rocprofv3 --pmc passes, kernel-trace only beside them
mkdir -p gpurun_out/r3
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
G2="SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"
G3="SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"
echo "== launch path (gemv_f16_kernel / gemv_f16_xp2_kernel, 128 launches per step, mean per dispatch)" > gpurun_out/r3/pmc_sq.txt
bash tools/gpu_pmc_kernel.sh "tools/prof_decode.py launch" gemv_f16 "$G1" "$G2" "$G3" >> gpurun_out/r3/pmc_sq.txt 2>&1
cat gpurun_out/r3/pmc_sq.txt
