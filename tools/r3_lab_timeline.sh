#!/bin/bash
# per-launch-type times and the in-kernel timeline of the decode kernel as shipped in round 3 (preload; 8 x 2 for single 2048-row layers)
mkdir -p gpurun_out/r3
{
  echo "==== per launch type (graph replay over > 256 MiB of distinct layers)"
  timeout 60 tools/gemv_lab_preload.bin | head -5
  for c in "4096 4096 3" "4096 4096 1" "11008 4096 2" "4096 11008 1"; do
    echo "== timeline N K layers: $c"
    timeout 60 tools/gemv_lab_preload_ts.bin $c
  done
} > gpurun_out/r3/lab_gemv_timeline.txt 2>&1
cat gpurun_out/r3/lab_gemv_timeline.txt
