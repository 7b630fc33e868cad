#!/bin/bash
# round 4 (collective default, peer opt-in): 2-rank dry run of the N > 1 bench path on ONE GPU (gloo carries the exchange; both ranks launch kernels on GPU 0): plumbing, not a measurement
mkdir -p gpurun_out/r4
export HQQ_BENCH_ONE_GPU=1 HQQ_BENCH_BACKEND=gloo HQQ_BENCH_EXCHANGE=rows1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --blocks 2 --steps 3 --warmup 1 --random-codes --no-single-gpu-reference > gpurun_out/r4/dist_dry.json 2> gpurun_out/r4/dist_dry.err
echo rc=$?; tail -n 3 gpurun_out/r4/dist_dry.err; python -c "
import json; d=json.loads(open('gpurun_out/r4/dist_dry.json').read().strip().split('\n')[-1]); print(d['ms_per_step'], d['config']['parallelism'][:60]); print(d['exchange'])"
HQQ_BENCH_EXCHANGE=gather timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --blocks 2 --steps 3 --warmup 1 --random-codes --no-single-gpu-reference 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step']); print(d['exchange'])"
# peer-memory exchange (csrc/exchange.hip): arenas mapped through IPC handles, validated against the gloo collective at start-up
HQQ_BENCH_EXCHANGE=peer timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --blocks 4 --steps 5 --warmup 2 --random-codes --no-single-gpu-reference 2> gpurun_out/r4/dist_dry_peer.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d['config']['parallelism']); print(d['exchange'])"
tail -n 3 gpurun_out/r4/dist_dry_peer.err
# the default (auto): the collective, no peer arena
unset HQQ_BENCH_EXCHANGE; export HQQ_BENCH_ONE_GPU=1 HQQ_BENCH_BACKEND=gloo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29520 bench.py --gpus 2 --blocks 2 --steps 3 --warmup 1 --random-codes --no-single-gpu-reference 2> gpurun_out/r4/dist_dry_auto.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('auto:', d['ms_per_step'], d['config']['parallelism']); print(d['exchange'])"
tail -n 2 gpurun_out/r4/dist_dry_auto.err
# prefill on two ranks (column shard + gather of [M, N / P])
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29521 bench.py --gpus 2 --workload prefill --prefill-tokens 2048 --steps 2 --warmup 1 --random-codes 2> gpurun_out/r4/dist_dry_prefill.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('prefill:', d['ms_per_step'], d['config']['parallelism'], d.get('tflops'))"
tail -n 2 gpurun_out/r4/dist_dry_prefill.err
