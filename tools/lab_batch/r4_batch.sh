#!/bin/bash
# round 4: the no-K-split batched-decode kernel — parity, then bs = 32 / 8 / 64 stacks against the split-K kernel on the same box
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_batch_gpu.py -x -q -m gpu --tb=short -n 4 > gpurun_out/r4/pytest_batch.txt 2>&1
tail -n 25 gpurun_out/r4/pytest_batch.txt
for bs in 32 8 64; do
  for o in 0 4096; do
    HQQ_BENCH_OPTS=$o timeout 300 python bench.py --bs $bs --random-codes --no-cpu-baseline --no-legs --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bs', $bs, 'opts', $o, 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'])"
  done
done
