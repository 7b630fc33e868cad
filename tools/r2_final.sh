#!/bin/bash
# end-of-round validation on one box: smoke + the whole GPU suite, the default bench line, rocprofv3 kernel stats of the same
# command, the PMC traffic passes, the 3-bit profile
bash tools/r2_gputest.sh
bash tools/r2_bench_default.sh
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2prof
mkdir -p $OUT
export TMPDIR=/tmp
run() {   # tag, bench args...
  tag=$1; shift
  ( cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs "$@" > $OUT/${tag}_bench_under_rocprof.json 2> $OUT/$tag.err )
  f=$(find $OUT/$tag -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${tag}_kernel_stats.csv
  find $OUT/$tag -name "*.csv" -size +2M -delete
  echo "== $tag"; head -4 $OUT/${tag}_kernel_stats.csv | cut -c1-220; cut -c1-260 $OUT/${tag}_bench_under_rocprof.json
}
run decode_int4
bash tools/gpu_pmc.sh r2prof/pmc > $OUT/pmc_stdout.txt 2>&1
cat gpurun_out/r2prof/pmc/pmc_summary.json
run decode_int3 --nbits 3
