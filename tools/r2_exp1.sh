#!/bin/bash
# round-2 experiment 1 (GPU box): solver mismatch survey, 3-op exact sequence check, bench before/after
set -x
mkdir -p gpurun_out/r2
python tools/solver_probe.py > gpurun_out/r2/solver_probe.txt 2>&1
python tools/sub_check.py > gpurun_out/r2/sub_check.txt 2>&1
for mode in exact sub factored; do
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --gemv-mode $mode > gpurun_out/r2/bench_int4_$mode.json 2> gpurun_out/r2/bench_int4_$mode.err
done
for mode in exact sub; do
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --nbits 2 --gemv-mode $mode > gpurun_out/r2/bench_int2_$mode.json 2> gpurun_out/r2/bench_int2_$mode.err
done
tail -n 50 gpurun_out/r2/solver_probe.txt
tail -n 70 gpurun_out/r2/sub_check.txt
cat gpurun_out/r2/bench_*.json
