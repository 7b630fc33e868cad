#!/bin/bash
# stand-alone decode-kernel lab: per-launch times (graph replay over > 256 MiB of distinct layers) and in-kernel timelines, per variant
mkdir -p gpurun_out/r2
{
  for v in ${LAB_VARIANTS:-plain w8 plain}; do
    echo "==== variant $v"
    timeout 120 tools/gemv_lab_$v.bin | head -${LAB_HEAD:-5}
  done
  for c in "4096 4096 3" "4096 4096 1"; do
    echo "== timeline N K layers: $c"
    timeout 60 tools/gemv_lab_ts.bin $c
  done
} > gpurun_out/r2/lab_gemv.txt 2>&1
cat gpurun_out/r2/lab_gemv.txt
