#!/bin/bash
# PMC passes (own runs, kernel-trace only beside them): HBM traffic of the decode GEMV.   usage: bash tools/gpu_pmc.sh TAG
TAG=${1:-pmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-legs --no-graph > $OUT/$C.json 2> $OUT/$C.err
done
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, json, collections
out = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob("gpurun_out/$TAG/%s/**/*counter_collection.csv" % C, recursive=True)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != C: continue
            k = r["Kernel_Name"][:60]
            agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    out[C] = {k: {"dispatches": n, "mean": v / n} for k, (n, v) in agg.items() if "gemv" in k}
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/$TAG/pmc_summary.json", "w"), indent=1)
PY
find gpurun_out/$TAG -name "*.csv" -size +5M -delete
ls -R gpurun_out/$TAG | head -30; tail -3 $OUT/FETCH_SIZE.err
