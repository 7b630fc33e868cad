#!/bin/bash
TAG=${1:-gemmpmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/p1 -o p -- python $GRAFT_REPO_ROOT/tools/gemm_probe.py > /dev/null 2> $OUT/p1.err
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/p2 -o p -- python $GRAFT_REPO_ROOT/tools/gemm_probe.py > /dev/null 2> $OUT/p2.err
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, collections
for d in ("p1","p2"):
    agg = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/$TAG/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(d, k, "%.4g" % (sum(v)/len(v)), len(v))
PY
tail -2 $OUT/p1.err | cut -c1-200
