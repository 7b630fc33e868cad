#!/bin/bash
mkdir -p gpurun_out/r3
timeout 600 python -m pytest tests/test_model_gpu.py -q -m gpu --tb=short > gpurun_out/r3/pytest_model.txt 2>&1
tail -n 8 gpurun_out/r3/pytest_model.txt
( time timeout 900 python bench.py > gpurun_out/r3/bench_default.json 2> gpurun_out/r3/bench_default.err ) 2>&1 | grep real
tail -n 5 gpurun_out/r3/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3/bench_default.json').read().strip().split('\n')[-1])
print('headline', d['ms_per_step'], d['value'], d['roofline']['frac'])
for l in d.get('legs',[]): print({k:v for k,v in l.items() if k in ('name','ms_per_step','roofline_frac','mfma_frac','tflops','error','status')})
print('quantize', d.get('quantize',{}).get('layers'), d.get('quantize',{}).get('roofline'))
print('e2e', d.get('end_to_end'))
PY
