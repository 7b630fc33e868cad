#!/bin/bash
for v in r2 r3; do echo "== $v"; for s in o qkv gateup down; do tools/batch_lab_$v.bin $s 32; done; tools/batch_lab_$v.bin o 8; tools/batch_lab_$v.bin qkv 64; done
tools/batch_lab_ts.bin o 32
tools/batch_lab_ts.bin qkv 32
timeout 600 python -m pytest tests/test_batch_gpu.py -q -m gpu --tb=line -n 4 -x 2>&1 | tail -n 5
