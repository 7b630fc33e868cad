#!/bin/bash
set -x
mkdir -p gpurun_out/r2
timeout 600 python tools/engine_check.py > gpurun_out/r2/engine_check.txt 2>&1
echo "rc=$?" >> gpurun_out/r2/engine_check.txt
tail -n 60 gpurun_out/r2/engine_check.txt
