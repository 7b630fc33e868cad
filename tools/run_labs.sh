#!/bin/bash
mkdir -p gpurun_out/labs
for b in tools/lab_*.bin; do echo "== $b"; timeout 120 $b $1; done 2>&1 | tee gpurun_out/labs/$(date +%s).log
