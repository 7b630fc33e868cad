#!/bin/bash
# quick GPU iteration: gemv parity + gemv microbench + bench + rocprof stats.   usage: bash tools/gpu_quick.sh TAG [pytest -k expr] [extra cmd]
TAG=${1:-q}
KEXPR=${2:-"gemv or forward or smoke"}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 600 python -m pytest tests -m gpu -x -q -k "$KEXPR" 2>&1 | tail -15 ) > $OUT/pytest.log
( timeout 600 python tools/microbench.py gemv 2>&1 ) > $OUT/microbench.log
( timeout 600 python bench.py --no-cpu-baseline 2>$OUT/bench.err ) > $OUT/bench.json
( timeout 600 python bench.py --no-cpu-baseline --gemv-mode factored 2>>$OUT/bench.err ) > $OUT/bench_factored.json
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --random-codes > $GRAFT_REPO_ROOT/$OUT/prof_bench.json 2>$GRAFT_REPO_ROOT/$OUT/prof.err )
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do head -4 $f | cut -c1-200; done
cat $OUT/pytest.log; cat $OUT/microbench.log; cat $OUT/bench.json $OUT/bench_factored.json; tail -5 $OUT/bench.err
if [ -n "$3" ]; then ( eval "$3" ) > $OUT/extra.log 2>&1; cat $OUT/extra.log; fi
find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
