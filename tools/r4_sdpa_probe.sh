#!/bin/bash
export TMPDIR=/tmp
cd /tmp
for v in a b; do
  rm -rf /tmp/sp_$v
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/sp_$v -o t -- python $GRAFT_REPO_ROOT/tools/sdpa_probe.py $v 2>&1 | grep variant
  python - $v <<'PY'
import csv, glob, collections, sys
c = collections.Counter(); t = collections.Counter()
for f in glob.glob(f"/tmp/sp_{sys.argv[1]}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        c[r["Kernel_Name"][:110]] += 1; t[r["Kernel_Name"][:110]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, n in c.most_common(12): print(f"   {n:4d}  {t[k]/n/1e3:7.2f} us  {k}")
PY
done
