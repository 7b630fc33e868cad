#!/bin/bash
mkdir -p gpurun_out/r2
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2/smoke.txt 2>&1; echo "smoke rc=$?" >> gpurun_out/r2/smoke.txt
tail -n 5 gpurun_out/r2/smoke.txt
timeout 1500 python -m pytest tests -q -m gpu -x -n 4 2>&1 | tail -n 40 > gpurun_out/r2/pytest_gpu.txt
cat gpurun_out/r2/pytest_gpu.txt
