#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5
mkdir -p $OUT
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 | tee $OUT/pytest_gpu_full.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $OUT/smoke.txt
timeout 900 python bench.py > $OUT/bench_default3.json 2> $OUT/bench_default3.err; cut -c1-400 $OUT/bench_default3.json; echo
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5/bench_default3.json"))
print(json.dumps(d.get("shard_of_8"), indent=1)[:2500])
e = d.get("end_to_end", {})
print({k: e.get(k) for k in ("tok_s", "ms_per_token", "identity_check", "with_separate_glue_kernels", "with_decode_attention_kernel")})
for l in d.get("legs", []): print(l.get("name"), l.get("ms_per_step"), l.get("roofline_frac"), l.get("tflops"))
PY
