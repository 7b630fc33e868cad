#!/bin/bash
# kernels per token / per decoder block of the fused decode loop: rocprofv3 --kernel-trace over three runs (NB blocks, T tokens)
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
cd /tmp
for cfg in "2 8" "2 24" "4 24"; do
  set -- $cfg
  rm -rf /tmp/e2e_$1_$2
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/e2e_$1_$2 -o t -- python $GRAFT_REPO_ROOT/tools/e2e_kernels.py $1 $2 ${ATT:-sdpa} > /tmp/e2e_$1_$2.log 2>&1
  tail -n 1 /tmp/e2e_$1_$2.log
done
python - <<'PY' | tee $GRAFT_REPO_ROOT/gpurun_out/r4/e2e_kernels_per_block.txt
import csv, glob, collections
def load(nb, t):
    c = collections.Counter()
    for f in glob.glob(f"/tmp/e2e_{nb}_{t}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)): c[r["Kernel_Name"][:90]] += 1
    return c
a, b, d = load(2, 8), load(2, 24), load(4, 24)
per_tok_2 = {k: (b[k] - a[k]) / 16 for k in b if b[k] != a[k]}
per_tok_4 = {k: (d[k] - b[k]) / 24 for k in d if d[k] != b[k]}   # 2 more blocks: prefill + 24 tokens each pass through them (prefill: HF's own modules, counted in: upper bound)
print("kernels per decoded token, 2 decoder blocks + embedding, final norm, lm_head, argmax:", sum(per_tok_2.values()))
for k, v in sorted(per_tok_2.items(), key=lambda kv: -kv[1]): print(f"   {v:6.2f}  {k}")
blk = {k: v / 2 for k, v in per_tok_4.items()}
print("kernels per decoder block per token (4-block run minus 2-block run, / 2 blocks / 24 tokens; the prefill pass through the extra blocks is in: an upper bound):", round(sum(blk.values()), 2))
for k, v in sorted(blk.items(), key=lambda kv: -kv[1]): print(f"   {v:6.2f}  {k}")
PY
