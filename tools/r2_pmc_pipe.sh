#!/bin/bash
# SQ counters of the pipelined GEMM at M = 8192 (256-token tile) and M = 128 (4-wave tile): separate --pmc passes, kernel-trace only
mkdir -p gpurun_out/r2
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
G2="SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"
G3="SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"
echo "== 4096 x 4096, M = 8192 (8 waves, 256-token tile)" > gpurun_out/r2/pmc_pipe.txt
bash tools/gpu_pmc_kernel.sh "tools/prof_gemm_pipe.py 4096 4096 8192 0" gemm_pipe_f16_kernel "$G1" "$G2" "$G3" >> gpurun_out/r2/pmc_pipe.txt 2>&1
echo "== 4096 x 4096, M = 128 (4 waves, 128-token tile, KS = 4)" >> gpurun_out/r2/pmc_pipe.txt
bash tools/gpu_pmc_kernel.sh "tools/prof_gemm_pipe.py 4096 4096 128 0" gemm_pipe_f16_kernel "$G1" "$G2" "$G3" >> gpurun_out/r2/pmc_pipe.txt 2>&1
cat gpurun_out/r2/pmc_pipe.txt
