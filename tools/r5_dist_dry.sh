#!/bin/bash
# round 5: 2-rank dry run of the N > 1 bench path on ONE GPU (gloo carries the collectives; both ranks launch kernels on GPU 0; the peer arenas are mapped
# through IPC handles): plumbing, not a measurement.  New: the peer-memory exchange with a decode BATCH (32 rows: strided slab writes, hqq_hip_exchange M > 1)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
export HQQ_BENCH_ONE_GPU=1 HQQ_BENCH_BACKEND=gloo
port=29530
for cfg in "peer 1" "peer 32" "auto 1" "auto 32"; do
  set -- $cfg
  port=$((port+1))
  if [ "$1" = auto ]; then unset HQQ_BENCH_EXCHANGE; else export HQQ_BENCH_EXCHANGE=$1; fi
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 2 --bs $2 --blocks 4 --steps 5 --warmup 2 --random-codes --no-single-gpu-reference 2> gpurun_out/r5/dist_dry_$1_$2.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1 bs=$2:', d['ms_per_step'], d['config']['parallelism'][:80]); print('   ', str(d.get('exchange'))[:400])"
  tail -n 2 gpurun_out/r5/dist_dry_$1_$2.err
done 2>&1 | tee gpurun_out/r5/dist_dry.txt
