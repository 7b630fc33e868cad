#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6j; mkdir -p $OUT
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 ) > $OUT/pytest.log
( time python bench.py > $OUT/bench_default.json 2>$OUT/bench.err ) 2> $OUT/time.txt
cat $OUT/pytest.log $OUT/time.txt; tail -3 $OUT/bench.err
python - <<'PY'
import json,os
d=json.load(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r6j/bench_default.json'))
print(d['ms_per_step'], d['roofline']['frac'])
for l in d.get('legs',[]): print(l.get('name')[:100], l.get('ms_per_step'), l.get('roofline_frac'), l.get('mfma_frac'), l.get('error'))
PY
