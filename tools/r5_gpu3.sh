#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_block_fold_gpu.py -x -q 2>&1 | tail -12 | tee $OUT/fold_tests2.txt
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "fused or bucket or glue" 2>&1 | tail -12 | tee $OUT/model_tests2.txt
ATT=sdpa bash tools/r4_e2e_kernels.sh > /dev/null 2>&1; cp gpurun_out/r4/e2e_kernels_per_block.txt $OUT/e2e_kernels_per_block.txt; head -32 $OUT/e2e_kernels_per_block.txt
HQQ_BENCH_E2E=1 timeout 900 python bench.py > $OUT/bench_default2.json 2> $OUT/bench_default2.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5/bench_default2.json"))
e = d.get("end_to_end", {})
print(d["ms_per_step"], d["roofline"]["frac"])
print({k: e.get(k) for k in ("tok_s", "ms_per_token", "identity_check", "with_separate_glue_kernels", "with_decode_attention_kernel", "glue")})
PY
