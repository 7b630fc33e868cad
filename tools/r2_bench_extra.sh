#!/bin/bash
# round-2 extra bench lines: prefill (one 7B block, fused vs composition at 8192 and 512 tokens), the 70B stack on one GPU
O=gpurun_out/r2; mkdir -p $O
for M in 8192 512; do
  python bench.py --workload prefill --prefill-tokens $M --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_prefill_${M}_fused.json 2>/dev/null
  python bench.py --workload prefill --prefill-tokens $M --steps 10 --warmup 3 --no-cpu-baseline --library-gemm > $O/bench_prefill_${M}_composition.json 2>/dev/null
done
python bench.py --workload decode70b --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_decode70b_1gpu.json 2>/dev/null
for f in $O/bench_prefill_*.json $O/bench_decode70b_1gpu.json; do echo "== $f"; python -c "
import json,sys
d=json.load(open('$f'))
print(d['ms_per_step'], d['value'], d['unit'], d.get('tflops'), d['roofline']['frac'], d['roofline']['kernel'][:60])"; done
