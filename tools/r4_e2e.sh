#!/bin/bash
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu --tb=short -x 2>&1 | tail -n 30
timeout 900 python bench.py --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/r4/bench_default.json 2> gpurun_out/r4/bench_default.err
tail -c 800 gpurun_out/r4/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4/bench_default.json").read().strip().splitlines()[-1])
print("headline:", d["ms_per_step"], "ms frac", d["roofline"]["frac"])
for l in d.get("legs", []): print(" leg:", l.get("name"), l.get("ms_per_step"), l.get("roofline_frac"), l.get("mfma_frac"), l.get("error"))
print("e2e:", d.get("end_to_end"))
print("quantize:", [(q["layer"], q["ms"]) for q in d.get("quantize", {}).get("layers", [])])
PY
