#!/bin/bash
# A/B on one box: default library vs variants; 7B stack bs=1 headline, three repetitions each, interleaved
for rep in 1 2 3; do
  for v in default "$@"; do
    if [ "$v" = default ]; then unset HQQ_AMD_LIB; else export HQQ_AMD_LIB=$PWD/tools/libhqq_hip_$v.so; fi
    python bench.py --no-legs --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['roofline']['frac'])"
  done
done
