#!/bin/bash
# SQ counters of the skinny GEMM at 32 rows: the wide tile (11008 x 4096) and the narrow one (4096 x 4096); separate rocprofv3 --pmc passes
mkdir -p gpurun_out/r3
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
G2="SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"
G3="SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"
echo "== wide tile: 11008 x 4096 int4 at 32 rows (172 workgroups x 16 chunks), mean per dispatch" > gpurun_out/r3/pmc_sq_skinny.txt
bash tools/gpu_pmc_kernel.sh "tools/prof_skinny.py 11008 4096 32" skinny_f16 "$G1" "$G2" "$G3" >> gpurun_out/r3/pmc_sq_skinny.txt 2>&1
echo "== narrow tile: 4096 x 4096 int4 at 32 rows (64 panels x 4 K splits), mean per dispatch" >> gpurun_out/r3/pmc_sq_skinny.txt
bash tools/gpu_pmc_kernel.sh "tools/prof_skinny.py 4096 4096 32" skinny_f16 "$G1" "$G2" "$G3" >> gpurun_out/r3/pmc_sq_skinny.txt 2>&1
cat gpurun_out/r3/pmc_sq_skinny.txt
