#!/bin/bash
# same-box A/B of library variants on the 7B stack (bs=1): tools/r6/ab.sh OUTFILE "<bench args>" name1 name2 ...   (name "cur" = hqq_amd/lib, else tools/r6/libhqq_hip_<name>.so)
OUT=$1; shift; ARGS=$1; shift
for rep in $(seq 1 ${REPS:-3}); do
  for v in "$@"; do
    if [ "$v" = cur ]; then unset HQQ_AMD_LIB; else export HQQ_AMD_LIB=$PWD/tools/r6/libhqq_hip_$v.so; fi
    python bench.py --no-legs --no-cpu-baseline --steps 30 --warmup 5 $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', '$ARGS', d['ms_per_step'], d['roofline']['frac'])"
  done
done >> $OUT
unset HQQ_AMD_LIB
