#!/bin/bash
# kernel-argument preload (layer-0 fast path) vs the kernel as shipped, per launch type (graph replay); binaries: tools/build_gemv_labs.sh plain preload
mkdir -p gpurun_out/r2
for v in ${LAB_VARIANTS:-plain preload}; do echo "==== $v"; timeout 20 tools/gemv_lab_$v.bin | head -5; done 2>&1 | tee gpurun_out/r2/lab_preload.txt
