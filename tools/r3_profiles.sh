#!/bin/bash
# round-3 profiles (GPU box): rocprofv3 --kernel-trace --stats of the bench command (headline, chained, bs=32, int2, int3), the default
# bench line, then the PMC traffic passes (separate runs).  Summaries land in gpurun_out/r3prof/; copy what is judged into profiles/.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3prof
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run() {   # tag, bench args...
  tag=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs "$@" > $OUT/${tag}_bench_under_rocprof.json 2> $OUT/$tag.err
  f=$(find $OUT/$tag -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${tag}_kernel_stats.csv
  find $OUT/$tag -name "*.csv" -size +2M -delete
}
run decode_int4
run decode_int4_chain --chain
run decode_int4_bs32 --bs 32
run decode_int2 --nbits 2
run decode_int3 --nbits 3
cd $GRAFT_REPO_ROOT
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
bash tools/gpu_pmc.sh r3prof/pmc > $OUT/pmc_stdout.txt 2>&1
for t in decode_int4 decode_int4_chain decode_int4_bs32 decode_int2 decode_int3; do echo "== $t"; head -5 $OUT/${t}_kernel_stats.csv | cut -c1-220; cat $OUT/${t}_bench_under_rocprof.json | cut -c1-200; echo; done
cat $OUT/bench_default.json | cut -c1-600
cat gpurun_out/r3prof/pmc/pmc_summary.json
