#!/bin/bash
# interleaved same-box A/B with means:  bash tools/r6/abmean.sh "<bench args>" REPS name1 name2 ...   (cur = hqq_amd/lib)
ARGS=$1; shift; N=$1; shift
mkdir -p gpurun_out/r6ab; o=gpurun_out/r6ab/abmean_$$.txt; rm -f $o
REPS=$N bash tools/r6/ab.sh $o "$ARGS" "$@"
python - <<PY
import collections
d=collections.defaultdict(list)
for l in open("$o"):
    p=l.split(); d[p[0]].append(float(p[-2]))
for k,v in d.items(): print(k, "$ARGS", " ".join(f"{x:.4f}" for x in v), " mean %.4f" % (sum(v)/len(v)))
PY
