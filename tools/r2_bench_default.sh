#!/bin/bash
mkdir -p gpurun_out/r2
(time python bench.py) > gpurun_out/r2/bench_default.json 2> gpurun_out/r2/bench_default.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r2/bench_default.json"))
print(d["ms_per_step"], d["roofline"]["frac"])
print(d.get("quantize"))
print(d["cpu_baseline"].get("quantize"))
for l in d["legs"]: print(l["name"][:50], l["ms_per_step"])
PY
tail -4 gpurun_out/r2/bench_default.err
