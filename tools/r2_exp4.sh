#!/bin/bash
mkdir -p gpurun_out/r2
for v in nocompute_nometa; do
  HQQ_AMD_LIB=$PWD/tools/libhqq_hip_lab_$v.so timeout 300 python tools/engine_ts.py 4 2>&1 | grep -v amdgpu.ids | grep "total\|stage span\|stream phase\|wave0" > gpurun_out/r2/engine_ts_$v.txt
  echo "== $v"; cat gpurun_out/r2/engine_ts_$v.txt
done
