#!/bin/bash
mkdir -p gpurun_out/r3
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3/smoke.txt 2>&1; echo "smoke rc=$?" >> gpurun_out/r3/smoke.txt
tail -n 2 gpurun_out/r3/smoke.txt
timeout 1700 python -m pytest tests -q -m gpu -n 4 --tb=short ${PYTEST_ARGS} > gpurun_out/r3/pytest_gpu_full.txt 2>&1
grep -v "^$" gpurun_out/r3/pytest_gpu_full.txt | grep -A25 "^____\|passed\|failed" | tail -n 60
