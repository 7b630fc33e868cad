#!/bin/bash
# usage: bash tools/_pmc.sh script.py kernel_substr "CNT1 CNT2" "CNT3 ..."   (one rocprofv3 pass per quoted group)
S=$1; KN=$2; shift 2
export TMPDIR=/tmp; cd /tmp
i=0
for G in "$@"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcx/p$i -o p -- python $GRAFT_REPO_ROOT/$S > /dev/null 2>$GRAFT_REPO_ROOT/gpurun_out/pmcx_err_$i.log
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pmcx/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$KN" not in r["Kernel_Name"]: continue
        k = r["Counter_Name"]; agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
for k, (n, v) in sorted(agg.items()): print(f"{k:32s} mean per dispatch {v / n:16.1f}  ({n} dispatches)")
PY
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmcx
