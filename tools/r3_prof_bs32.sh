#!/bin/bash
# rocprofv3 kernel stats of the bs = 32 decode stack after the skinny GEMM's narrow tile went in
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3prof
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
tag=decode_int4_bs32_narrow
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs --bs 32 > $OUT/${tag}_bench_under_rocprof.json 2> $OUT/$tag.err
f=$(find $OUT/$tag -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $OUT/${tag}_kernel_stats.csv
find $OUT/$tag -name "*.csv" -size +2M -delete
head -6 $OUT/${tag}_kernel_stats.csv | cut -c1-260; cut -c1-300 $OUT/${tag}_bench_under_rocprof.json
