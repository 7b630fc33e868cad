#!/bin/bash
mkdir -p gpurun_out/r4
for s in o qkv gateup down; do tools/batch_lab.bin $s 32; done
tools/batch_lab.bin o 8
tools/batch_lab_ts.bin o 32
tools/batch_lab_ts.bin o 8
tools/batch_lab_ts.bin qkv 32
tools/batch_lab_ts.bin down 32
timeout 600 python -m pytest tests/test_w3s_gpu.py -q -m gpu --tb=line -n 4 2>&1 | tail -n 8
