#!/bin/bash
mkdir -p gpurun_out/r4
HQQ_AMD_LIB=$PWD/tools/libhqq_hip_gdtrace.so python tools/gd_trace.py "$@" 2>&1 | tee gpurun_out/r4/dense_trace.txt
