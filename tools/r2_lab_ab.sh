#!/bin/bash
# lab: A/B of library variants (tools/libhqq_hip_<name>.so, built by tools/build_variant.sh) on tools/lab_pipe_cases.py
R=$GRAFT_REPO_ROOT
echo -n "shipped: "; python $R/tools/lab_pipe_cases.py $LAB_OPTS $LAB_BIG 2>/dev/null
for v in $VARIANTS; do echo -n "$v: "; HQQ_AMD_LIB=$R/tools/libhqq_hip_$v.so python $R/tools/lab_pipe_cases.py $LAB_OPTS $LAB_BIG 2>/dev/null; done
