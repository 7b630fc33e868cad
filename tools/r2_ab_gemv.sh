#!/bin/bash
# gemv.hip rework (two units really in flight, buffer loads, lean row end): decode parity tests, then A/B against the previous library
mkdir -p gpurun_out/r2
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_round2_gpu.py -q -m gpu -n 4 --tb=short -x -k "not engine and not solver and not quant and not pipe" > gpurun_out/r2/pytest_gemv.txt 2>&1
tail -n 15 gpurun_out/r2/pytest_gemv.txt
bash tools/ab.sh old 2>&1 | tee gpurun_out/r2/ab_gemv.txt
