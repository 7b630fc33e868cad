#!/bin/bash
# lab: what the skinny kernel's fixed cost is made of — bs = 32 stack with one part compiled out.  Build the variants first:
#   for v in NOMETA NOX NOFIN; do bash tools/build_variant.sh sk_$v skinny.hip "-DSK_LAB_$v"; done;  VARIANTS="sk_NOMETA sk_NOX sk_NOFIN" bash tools/r2_lab_skinny.sh
R=$GRAFT_REPO_ROOT
run() { python $R/bench.py --bs 32 --no-cpu-baseline --no-legs --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
echo -n "shipped: "; run
for v in $VARIANTS; do echo -n "$v: "; HQQ_AMD_LIB=$R/tools/libhqq_hip_$v.so run; done
