#!/bin/bash
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_round2_gpu.py tests/test_layer_gpu.py -q -m gpu -n 4 --tb=short -x -k "not engine and not solver and not quant and not pipe" > gpurun_out/r3/pytest_w8.txt 2>&1
tail -n 4 gpurun_out/r3/pytest_w8.txt
for rep in 1 2; do
  HQQ_BENCH_E2E=0 timeout 300 python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read())
print('stack', d['ms_per_step'], d['roofline']['frac'], '| single4096', [ (l['ms_per_step'], l['roofline_frac']) for l in d['legs'] if l['name'].startswith('4096x4096 bs=1 (one')], '| int2', [l['ms_per_step'] for l in d['legs'] if 'int2' in l['name']])"
done 2>&1 | tee gpurun_out/r3/ab_w8_landed.txt
