#!/bin/bash
mkdir -p gpurun_out/r2
timeout 600 python tools/engine_check.py > gpurun_out/r2/engine_check.txt 2>&1
echo "rc=$?" >> gpurun_out/r2/engine_check.txt
grep -v "^chain" gpurun_out/r2/engine_check.txt | tail -n 30
HQQ_AMD_LIB=$PWD/tools/libhqq_hip_lab.so timeout 300 python tools/engine_ts.py 4 > gpurun_out/r2/engine_ts.txt 2>&1
cat gpurun_out/r2/engine_ts.txt
