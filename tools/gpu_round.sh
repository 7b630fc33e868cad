#!/bin/bash
# One GPU-box session: parity tests, smoke, micro-benchmarks, the contract bench, and a rocprofv3 kernel trace of it.
# usage (via gpurun): bash tools/gpu_round.sh [tag]
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $OUT/pytest.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 ) > $OUT/smoke.log
( timeout 600 python tools/microbench.py gemv gemm quant pack 2>&1 ) > $OUT/microbench.log
( timeout 600 python bench.py 2>$OUT/bench.err ) > $OUT/bench.json
( timeout 600 python bench.py --workload prefill --steps 5 --warmup 2 --no-cpu-baseline 2>>$OUT/bench.err ) > $OUT/bench_prefill.json
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --random-codes > $GRAFT_REPO_ROOT/$OUT/prof_bench.json 2>$GRAFT_REPO_ROOT/$OUT/prof.err )
find $OUT/prof -name "*stats*" | head; 
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do head -12 $f; done
cat $OUT/pytest.log $OUT/smoke.log; tail -60 $OUT/microbench.log; cat $OUT/bench.json $OUT/bench_prefill.json; tail -5 $OUT/bench.err
# keep the merge-back small: drop the raw trace, keep the stats
find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
