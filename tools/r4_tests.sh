#!/bin/bash
# round 4: the whole GPU suite (parity first)
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests -q -m gpu --tb=short -n 4 > gpurun_out/r4/pytest_gpu.txt 2>&1
tail -n 30 gpurun_out/r4/pytest_gpu.txt
