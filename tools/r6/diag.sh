#!/bin/bash
mkdir -p gpurun_out/r6diag
HQQ_AMD_LIB=$PWD/tools/r6/libhqq_hip_ts.so python tools/r6/ts_run.py > gpurun_out/r6diag/timeline.txt 2>&1
grep -v "XCC\|waves [0-9]*\.\.[0-9]* to start" gpurun_out/r6diag/timeline.txt
