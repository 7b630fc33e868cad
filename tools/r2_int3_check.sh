#!/bin/bash
# 3-bit decode kernels after the loop / buffer-load rework: parity tests, the forward fuzzer, bench --nbits 3
mkdir -p gpurun_out/r2
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_round2_gpu.py -q -m gpu -n 4 --tb=short -x -k "3bit or three_op or meta_check or nan or tiny" > gpurun_out/r2/pytest_int3.txt 2>&1
tail -n 6 gpurun_out/r2/pytest_int3.txt
timeout 300 python tools/fuzz_forward.py 150 2>&1 | tail -n 3
for i in 1 2; do python bench.py --nbits 3 --no-legs --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('int3', d['ms_per_step'], d['roofline']['frac'])"; done
