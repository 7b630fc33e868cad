#!/bin/bash
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_round2_gpu.py tests/test_layer_gpu.py tests/test_round3_gpu.py -q -m gpu -n 4 --tb=short -x -k "not engine and not solver and not quant and not pipe" > gpurun_out/r3/pytest_x2.txt 2>&1
tail -n 3 gpurun_out/r3/pytest_x2.txt
bash tools/r3_ab_generic.sh prev
