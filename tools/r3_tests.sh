#!/bin/bash
mkdir -p gpurun_out/r3
python tools/factored_measure.py > gpurun_out/r3/factored_measure.txt 2>&1; cat gpurun_out/r3/factored_measure.txt | grep -v amdgpu.ids
timeout 1200 python -m pytest tests/test_round3_gpu.py tests/test_chain_gpu.py -q -m gpu --tb=short -n 2 > gpurun_out/r3/pytest_r3.txt 2>&1
grep -v "^$" gpurun_out/r3/pytest_r3.txt | tail -n 60
