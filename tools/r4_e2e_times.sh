#!/bin/bash
# per-kernel time of the fused decode loop (stream-ordered, 8 blocks, 40 tokens): where a token's non-linear time goes
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/e2e_t
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/e2e_t -o t -- python $GRAFT_REPO_ROOT/tools/e2e_kernels.py 8 40 ${1:-sdpa} > /tmp/e2e_t.log 2>&1
python - <<'PY' | tee $GRAFT_REPO_ROOT/gpurun_out/r4/e2e_kernel_times_${1:-sdpa}.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: [0, 0])
for f in glob.glob("/tmp/e2e_t/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:100]; agg[k][0] += 1; agg[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in agg.values())
print("kernel, calls, total us, avg us  (8 blocks, prefill of 16 + 39 decode steps, set-up kernels included)")
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]: print(f"{n:6d} {t/1e3:10.1f} {t/n/1e3:8.2f}  {k}")
PY
