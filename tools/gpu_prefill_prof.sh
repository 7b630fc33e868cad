#!/bin/bash
# prefill (configs[2]) evidence: kernel stats + MFMA-busy PMC pass of `bench.py --workload prefill`.   usage: bash tools/gpu_prefill_prof.sh TAG
TAG=${1:-prefill}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --workload prefill --steps 3 --warmup 1 --no-cpu-baseline --random-codes --no-graph"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o p -- $CMD > $OUT/stats.json 2> $OUT/stats.err
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc -o p -- $CMD > $OUT/pmc.json 2> $OUT/pmc.err
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/pmc2 -o p -- $CMD > $OUT/pmc2.json 2> $OUT/pmc2.err
cd $GRAFT_REPO_ROOT
head -3 gpurun_out/$TAG/stats/*kernel_stats.csv | cut -c1-160
python - <<PY
import csv, glob, collections, json
res = {}
for d in ("pmc", "pmc2"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("gpurun_out/$TAG/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_f16" in r["Kernel_Name"]:
                agg[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        res.setdefault(k, {}).update({c: sum(x) / len(x) for c, x in v.items()})
        res[k]["dispatches"] = len(next(iter(v.values())))
print(json.dumps(res, indent=1))
json.dump(res, open("gpurun_out/$TAG/pmc_summary.json", "w"), indent=1)
PY
tail -2 $OUT/pmc.err $OUT/pmc2.err; cat $OUT/stats.json
find gpurun_out/$TAG -name "*.csv" -size +5M -delete
