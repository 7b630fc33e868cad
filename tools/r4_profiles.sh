#!/bin/bash
# round-4 profiles (GPU box): rocprofv3 --kernel-trace --stats of the bench command (headline, bs=32, int2, int3, prefill at 8192 and 65,536
# tokens), the default bench line, the PMC traffic passes of the decode kernel and the SQ counters / clock of the dense GEMM (separate runs).
# Summaries land in gpurun_out/r4prof/; copy what is judged into profiles/.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4prof
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
run decode_int4_bs32 --bs 32
run decode_int2 --nbits 2
run decode_int3 --nbits 3
run prefill_int4_8192 --workload prefill --prefill-tokens 8192 --steps 10 --warmup 3
run prefill_int4_65536 --workload prefill --prefill-tokens 65536 --steps 3 --warmup 1
cd $GRAFT_REPO_ROOT
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --workload prefill --prefill-tokens 65536 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_prefill_65536.json 2> $OUT/bench_prefill_65536.err
python bench.py --workload prefill --prefill-tokens 8192 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_prefill_8192.json 2> $OUT/bench_prefill_8192.err
bash tools/gpu_pmc.sh r4prof/pmc > $OUT/pmc_stdout.txt 2>&1
bash tools/r4_dense_pmc.sh > /dev/null 2>&1; cp gpurun_out/r4/pmc_dense.txt $OUT/dense_sq_counters.txt
bash tools/r4_dense_clock.sh > $OUT/dense_clock.txt 2>&1
for t in decode_int4 decode_int4_bs32 decode_int2 decode_int3 prefill_int4_8192 prefill_int4_65536; do echo "== $t"; head -6 $OUT/${t}_kernel_stats.csv | cut -c1-220; cat $OUT/${t}_bench_under_rocprof.json | cut -c1-300; echo; done
cat $OUT/bench_default.json | cut -c1-600
cat $OUT/bench_prefill_65536.json | cut -c1-900
cat gpurun_out/r4prof/pmc/pmc_summary.json
cat $OUT/dense_clock.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; tail -n 2 $OUT/smoke.txt
python bench.py --workload decode70b --no-cpu-baseline --steps 10 --warmup 3 > $OUT/bench_decode70b_1gpu.json 2> $OUT/bench_decode70b_1gpu.err; cut -c1-500 $OUT/bench_decode70b_1gpu.json
python bench.py --workload prefill --shapes 70b --prefill-tokens 8192 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_prefill70b_8192.json 2> $OUT/bench_prefill70b_8192.err; cut -c1-700 $OUT/bench_prefill70b_8192.json
