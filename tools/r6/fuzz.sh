#!/bin/bash
# Round-6 fuzz pass on the final build (development aid; writes gpurun_out/r6fuzz/).
mkdir -p gpurun_out/r6fuzz
o=gpurun_out/r6fuzz
for s in 61 62 63; do timeout 500 python tools/fuzz_forward.py 500 $s > $o/forward_$s.txt 2>&1; tail -1 $o/forward_$s.txt; done
timeout 400 python tools/fuzz_forward.py 400 64 factored > $o/forward_factored.txt 2>&1; tail -1 $o/forward_factored.txt
for s in 61 62; do timeout 500 python tools/fuzz_ops.py 400 $s > $o/ops_$s.txt 2>&1; tail -1 $o/ops_$s.txt; done
for s in 61 62; do timeout 600 python tools/fuzz_layers.py 120 $s > $o/layers_$s.txt 2>&1; tail -1 $o/layers_$s.txt; done
for s in 61 62; do timeout 600 python tools/fuzz_block.py 300 $s > $o/block_$s.txt 2>&1; tail -1 $o/block_$s.txt; done
