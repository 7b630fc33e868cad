#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "4096 4096 128 0" "12288 4096 128 0" "22016 4096 128 0" "4096 11008 128 0" "12288 4096 256 0" "4096 4096 512 0"; do
  tag=$(echo $cfg | tr ' ' '_')
  rm -rf /tmp/prof_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- python $R/tools/prof_gemm_pipe.py $cfg > /dev/null 2>&1
  f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
  echo "== N K M KS(0 = built-in plan): $cfg"
  python - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'gemm_pipe' in r['Name']:
        print(f"{r['Name'][:70]:70s} calls={r['Calls']} avg_us={float(r['AverageNs'])/1e3:.2f} min_us={float(r['MinNs'])/1e3:.2f}")
PY
done
