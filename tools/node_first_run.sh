#!/bin/bash
# First thing to run when a multi-GPU MI355X node is available (nothing here has been measured over xGMI yet — DESIGN.md section 6).
#   bash tools/node_first_run.sh [out_dir]
# 1. the two-process exchange test (IPC-mapped arenas, real waits) — on a node the ranks still share GPU 0 there, so also:
# 2. bench.py at N = 1, 2, 4, 8 with every exchange mode (peer-memory kernel, coalesced per-slab RCCL gathers, shard-wide gather + un-permute);
#    each JSON line's `exchange` block says which mode actually ran, whether the peer mode validated against the collective, and the
#    status word of its bounded waits; stderr says why a mode was dropped.
# 3. a rocprofv3 kernel trace of the N = 2 peer run (per-kernel durations of hqq::exchange_kernel beside the GEMVs).
OUT=${1:-gpurun_out/node}
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -m pytest tests/test_exchange_gpu.py -m gpu -q 2>&1 | tail -3 | tee $OUT/exchange_tests.txt
NG=$(python -c "import torch; print(torch.cuda.device_count())")
python bench.py --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench_n1.err
for N in 2 4 8; do
  [ $N -le $NG ] || continue
  for MODE in auto peer rows1 gather; do
    HQQ_BENCH_EXCHANGE=$MODE timeout 1800 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600 + N)) \
      bench.py --gpus $N --steps 20 --warmup 5 > $OUT/bench_n${N}_${MODE}.json 2> $OUT/bench_n${N}_${MODE}.err
    python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_n${N}_${MODE}.json").read().strip().split("\n")[-1])
    x = d.get("exchange", {})
    print("N=$N mode=$MODE:", d["ms_per_step"], "ms/token,", d["value"], d["unit"], "| exchange", x.get("ms_per_step"), "ms:", (x.get("mode") or "")[:60], "| peer status", x.get("peer_status"), "| single GPU", (d.get("single_gpu") or {}).get("ms_per_step"))
except Exception as e:
    print("N=$N mode=$MODE: no result (", e, ") — see $OUT/bench_n${N}_${MODE}.err")
PY
  done
  # the adaptive plan (hqq_amd.shard.plan_exchange_groups): groups too small to shard are computed whole by every rank; compare `value` with the auto line above
  timeout 1800 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29700 + N)) \
    bench.py --gpus $N --steps 20 --warmup 5 --plan adaptive --no-single-gpu-reference > $OUT/bench_n${N}_adaptive.json 2> $OUT/bench_n${N}_adaptive.err
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_n${N}_adaptive.json").read().strip().split("\n")[-1])
    print("N=$N plan=adaptive:", d["ms_per_step"], "ms/token,", d["value"], d["unit"], "| groups", d["plan"]["groups"], "| exchange us per point", d["exchange"]["us_per_point"])
except Exception as e:
    print("N=$N plan=adaptive: no result (", e, ")")
PY
  for BS in 32; do
    timeout 1800 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29800 + N)) \
      bench.py --gpus $N --bs $BS --steps 20 --warmup 5 --no-single-gpu-reference > $OUT/bench_n${N}_bs${BS}.json 2> $OUT/bench_n${N}_bs${BS}.err
  done
  timeout 1800 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29900 + N)) \
    bench.py --gpus $N --workload decode --steps 20 --warmup 5 > $OUT/bench_n${N}_weak.json 2> $OUT/bench_n${N}_weak.err
done
if [ 2 -le $NG ]; then
  export TMPDIR=/tmp
  (cd /tmp && HQQ_BENCH_EXCHANGE=peer timeout 1800 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_n2 -o p -- \
     python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29650 $OLDPWD/bench.py --gpus 2 --steps 10 --warmup 3 --no-single-gpu-reference > $OLDPWD/$OUT/bench_n2_under_rocprof.json 2> $OLDPWD/$OUT/prof_n2.err)
  find $OUT/prof_n2 -name "*kernel_stats.csv" | head -2 | xargs -r head -8 | cut -c1-200
fi
