#!/bin/bash
# SQ counters of the dense GEMM at 8192 x 4096 x 4096: separate --pmc passes, kernel-trace only
mkdir -p gpurun_out/r4
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
G2="SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"
G3="SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"
echo "== dense GEMM 8192 x 4096 x 4096 (in-tree)" > gpurun_out/r4/pmc_dense.txt
bash tools/gpu_pmc_kernel.sh "tools/prof_gemm_dense.py 8192 4096 4096" dense_gemm_kernel "$G1" "$G2" "$G3" >> gpurun_out/r4/pmc_dense.txt 2>&1
cat gpurun_out/r4/pmc_dense.txt
