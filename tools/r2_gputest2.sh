#!/bin/bash
mkdir -p gpurun_out/r2
for i in 1 2; do
timeout 900 python -m pytest tests/test_round2_gpu.py -q -m gpu -n 4 2>&1 | grep -v "^$" | grep -B30 "Error\|passed\|failed" | tail -n 80 > gpurun_out/r2/pytest_r2_$i.txt
cat gpurun_out/r2/pytest_r2_$i.txt
done
