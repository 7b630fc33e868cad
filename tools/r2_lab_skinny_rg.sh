#!/bin/bash
# skinny GEMM panel shape: 64 packed rows x 8 waves (default) vs 32 rows x 4 waves (rg2) vs 32 rows x 2 waves (rg2s1); bs=32 7B stack
mkdir -p gpurun_out/r2
for rep in 1 2; do
  for v in default rg2 rg2s1; do
    if [ "$v" = default ]; then unset HQQ_AMD_LIB; else export HQQ_AMD_LIB=$PWD/tools/libhqq_hip_$v.so; fi
    python bench.py --bs 32 --no-legs --no-cpu-baseline --random-codes --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['roofline']['frac'])"
  done
done 2>&1 | tee gpurun_out/r2/lab_skinny_rg.txt
export HQQ_AMD_LIB=$PWD/tools/libhqq_hip_rg2.so
timeout 300 python -m pytest tests/test_hip_parity.py -q -m gpu -n 4 -x --tb=line -k "skinny or batch" 2>&1 | tail -3
