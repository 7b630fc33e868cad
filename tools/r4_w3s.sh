#!/bin/bash
# round 4, first GPU call: the 3-bit stream layout — parity tests, then timing of the int3 7B stack (bs = 1 and bs = 32)
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_w3s_gpu.py -x -q -m gpu --tb=short -n 4 > gpurun_out/r4/pytest_w3s.txt 2>&1
tail -n 40 gpurun_out/r4/pytest_w3s.txt
timeout 600 python bench.py --nbits 3 --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/r4/bench_int3.json 2> gpurun_out/r4/bench_int3.err
tail -c 1500 gpurun_out/r4/bench_int3.err; python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r4/bench_int3.json").read().strip().splitlines()[-1])
    print("int3 bs=1:", d["ms_per_step"], "ms", d["value"], "GB/s frac", d["roofline"]["frac"], d["config"].get("layers_with_three_op_rebuild"))
    for l in d.get("legs", []): print(" leg:", l.get("name"), l.get("ms_per_step"), l.get("roofline_frac"), l.get("error"))
except Exception as e:
    print("bench parse failed", e)
PY
timeout 600 python bench.py --nbits 3 --bs 32 --no-cpu-baseline --no-legs --steps 20 --warmup 5 > gpurun_out/r4/bench_int3_bs32.json 2> gpurun_out/r4/bench_int3_bs32.err
tail -c 600 gpurun_out/r4/bench_int3_bs32.err; tail -c 900 gpurun_out/r4/bench_int3_bs32.json
