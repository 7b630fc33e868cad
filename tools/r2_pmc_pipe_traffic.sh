#!/bin/bash
# HBM traffic of the pipelined GEMM (FETCH_SIZE / WRITE_SIZE in separate --pmc passes, kernel-trace only): 4096 x 4096 at M = 128 and M = 8192
mkdir -p gpurun_out/r2
for cfg in "4096 4096 128 0" "4096 4096 8192 0"; do
  echo "== N K M KS(0 = plan): $cfg"
  bash tools/gpu_pmc_kernel.sh "tools/prof_gemm_pipe.py $cfg" gemm_pipe "FETCH_SIZE" "WRITE_SIZE"
done > gpurun_out/r2/pmc_pipe_traffic.txt 2>&1
cat gpurun_out/r2/pmc_pipe_traffic.txt
