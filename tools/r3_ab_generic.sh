#!/bin/bash
# A/B on one box: default library vs variants ($@); 7B stack bs=1 headline + single-layer + int2 legs, two repetitions, interleaved
mkdir -p gpurun_out/r3
for rep in 1 2; do
for v in default "$@"; do
  if [ "$v" = default ]; then unset HQQ_AMD_LIB; else export HQQ_AMD_LIB=$PWD/tools/libhqq_hip_$v.so; fi
  HQQ_BENCH_E2E=0 timeout 300 python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read())
print('$v', 'stack', d['ms_per_step'], d['roofline']['frac'], '| single4096', [ (l['ms_per_step'], l['roofline_frac']) for l in d['legs'] if l['name'].startswith('4096x4096 bs=1 (one')], '| int2', [l['ms_per_step'] for l in d['legs'] if 'int2' in l['name']])"
done; done 2>&1 | tee gpurun_out/r3/ab_$1.txt
