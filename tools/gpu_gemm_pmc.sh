#!/bin/bash
TAG=${1:-gemmpmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --kernel-trace --output-format csv -d $OUT/p1 -o p -- python $GRAFT_REPO_ROOT/tools/gemm_probe.py > /dev/null 2> $OUT/p1.err
timeout 300 rocprofv3 --pmc TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $OUT/p2 -o p -- python $GRAFT_REPO_ROOT/tools/gemm_probe.py > /dev/null 2> $OUT/p2.err
timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/p3 -o p -- python $GRAFT_REPO_ROOT/tools/gemm_probe.py > /dev/null 2> $OUT/p3.err
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, collections
for d in ("p1","p2","p3"):
    agg = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/$TAG/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(d, k, "%.4g" % (sum(v)/len(v)), len(v))
PY
grep -i "error\|invalid\|not found" $OUT/p*.err | head -5
