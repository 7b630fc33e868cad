#!/bin/bash
# next step for the decode kernel: the kernel-argument-preload build of gemv.hip against the shipped one.
#   here (no GPU needed):  bash tools/build_variant.sh preload gemv.hip "-DGV_LAB_PRELOAD -mllvm -amdgpu-kernarg-preload-count=16"
#   on the GPU box:        bash tools/r3_preload_check.sh
# 1. the decode parity tests with the variant library loaded (HQQ_AMD_LIB), 2. the 7B-stack A/B on this box (tools/ab.sh).
# If both are good: make it the default (Makefile: FLAGS_gemv += -DGV_LAB_PRELOAD -mllvm -amdgpu-kernarg-preload-count=16).
mkdir -p gpurun_out/r3
export HQQ_AMD_LIB=$PWD/tools/libhqq_hip_preload.so
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_round2_gpu.py tests/test_layer_gpu.py -q -m gpu -n 4 --tb=short -x -k "not engine and not solver and not quant and not pipe" > gpurun_out/r3/pytest_preload.txt 2>&1
tail -n 6 gpurun_out/r3/pytest_preload.txt
unset HQQ_AMD_LIB
bash tools/ab.sh preload 2>&1 | tee gpurun_out/r3/ab_preload.txt
