#!/bin/bash
# effective clock of the dense GEMM and of the library's: GRBM_GUI_ACTIVE / kernel duration in the same pass
mkdir -p gpurun_out/r4
export TMPDIR=/tmp; cd /tmp
for arm in "" lib; do
  rm -rf /tmp/clk
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d /tmp/clk -o p -- python $GRAFT_REPO_ROOT/tools/prof_gemm_dense.py ${1:-8192} ${2:-4096} ${3:-4096} $arm > /dev/null 2>/tmp/clk_err.log
  python - "$arm" <<'PY'
import csv, glob, collections, sys
dur = {}
for f in glob.glob("/tmp/clk/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/clk/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append((float(r["Counter_Value"]), dur.get(r["Dispatch_Id"], ("", 0))[1]))
for k, cs in agg.items():
    if "gemm" not in k.lower() and "Cijk" not in k: continue
    print("arm", sys.argv[1] or "in-tree", "|", k)
    for c, v in sorted(cs.items()):
        n = len(v); mv = sum(a for a, _ in v) / n; md = sum(d for _, d in v) / n
        print(f"   {c:28s} {mv:16.1f}   duration {md/1e3:8.1f} us   per us {mv/ (md/1e3):12.1f}")
PY
done
