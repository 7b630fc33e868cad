#!/bin/bash
mkdir -p gpurun_out/r2
python bench.py > gpurun_out/r2/bench_default.json 2> gpurun_out/r2/bench_default.err; echo "rc=$?"
cat gpurun_out/r2/bench_default.json; tail -3 gpurun_out/r2/bench_default.err
HQQ_BENCH_ONE_GPU=1 HQQ_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --blocks 2 --steps 3 --warmup 1 > gpurun_out/r2/bench_2rank_debug.json 2> gpurun_out/r2/bench_2rank_debug.err; echo "rc=$?"
cat gpurun_out/r2/bench_2rank_debug.json; tail -5 gpurun_out/r2/bench_2rank_debug.err
