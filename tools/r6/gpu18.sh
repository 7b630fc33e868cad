#!/bin/bash
# same-box: per-launch histograms of the round-5 decode kernels vs the current ones, and the headline A/B
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r6q; mkdir -p $OUT; export TMPDIR=/tmp
for v in r5gemv cur r5gemv cur; do
  if [ "$v" = cur ]; then unset HQQ_AMD_LIB; else export HQQ_AMD_LIB=$R/tools/r6/libhqq_hip_$v.so; fi
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_$v -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs > $OUT/b_$v.json 2>/dev/null )
  echo "######## $v  $(python -c "import json; d=json.load(open('$OUT/b_$v.json')); print(d['ms_per_step'], d['roofline']['frac'])")" >> $OUT/hist.txt
  python $R/tools/r6/per_launch.py $OUT/prof_$v | head -8 | cut -c1-420 >> $OUT/hist.txt
  rm -rf $OUT/prof_$v
done
unset HQQ_AMD_LIB
cd $R; bash tools/r6/ab.sh $OUT/ab.txt "" r5gemv cur
cat $OUT/hist.txt $OUT/ab.txt
