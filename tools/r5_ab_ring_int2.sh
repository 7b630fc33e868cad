#!/bin/bash
cd $GRAFT_REPO_ROOT
for nb in 2 4; do
for rep in 1 2; do
  for v in default ring2 ring3 ring4; do
    if [ "$v" = default ]; then unset HQQ_AMD_LIB; else export HQQ_AMD_LIB=$PWD/tools/libhqq_hip_$v.so; fi
    python bench.py --nbits $nb --no-legs --no-cpu-baseline --steps 30 --warmup 5 --random-codes 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('int$nb', '$v', d['ms_per_step'], d['roofline']['frac'])"
  done
done
done
