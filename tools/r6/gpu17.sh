#!/bin/bash
for v in ts ts0; do echo "######## $v"; HQQ_AMD_LIB=$PWD/tools/r6/libhqq_hip_$v.so python tools/r6/ts_run.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r6_ts.txt
cat gpurun_out/r6_ts.txt
