#!/bin/bash
# round 3: chained launches (gemv_chain.hip) — parity tests, then A/B against the stream-ordered launches on this box
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_chain_gpu.py -q -m gpu -x --tb=short > gpurun_out/r3/pytest_chain.txt 2>&1
tail -n 15 gpurun_out/r3/pytest_chain.txt
for rep in 1 2 3; do
  for v in default chain; do
    extra=""; [ "$v" = chain ] && extra="--chain"
    timeout 300 python bench.py $extra --no-legs --no-cpu-baseline --steps 30 --warmup 5 2>gpurun_out/r3/bench_$v.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['roofline']['frac'])"
  done
done 2>&1 | tee gpurun_out/r3/ab_chain.txt
tail -n 3 gpurun_out/r3/bench_chain.err
