#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6i; mkdir -p $OUT
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > $OUT/pytest.log
( time python bench.py > $OUT/bench_default.json 2>$OUT/bench.err ) 2> $OUT/time.txt
cat $OUT/pytest.log $OUT/time.txt; tail -3 $OUT/bench.err
python - <<'PY'
import json,os
d=json.load(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r6i/bench_default.json'))
print(d['ms_per_step'], d['roofline']['frac'])
print(json.dumps(d.get('shard_of_8'),indent=0)[:3000])
PY
