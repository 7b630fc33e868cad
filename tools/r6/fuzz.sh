#!/bin/bash
# Round-6 fuzz pass on the final build (development aid; writes gpurun_out/r6fuzz/).   bash tools/r6/fuzz.sh [first seed]
S=${1:-61}
mkdir -p gpurun_out/r6fuzz
o=gpurun_out/r6fuzz
for s in $S $((S+1)) $((S+2)); do timeout 500 python tools/fuzz_forward.py 500 $s > $o/forward_$s.txt 2>&1; tail -1 $o/forward_$s.txt; done
timeout 400 python tools/fuzz_forward.py 400 $((S+3)) factored > $o/forward_factored_$((S+3)).txt 2>&1; tail -1 $o/forward_factored_$((S+3)).txt
for s in $S $((S+1)); do timeout 500 python tools/fuzz_ops.py 400 $s > $o/ops_$s.txt 2>&1; tail -1 $o/ops_$s.txt; done
for s in $S $((S+1)); do timeout 600 python tools/fuzz_layers.py 120 $s > $o/layers_$s.txt 2>&1; tail -1 $o/layers_$s.txt; done
for s in $S $((S+1)); do timeout 600 python tools/fuzz_block.py 300 $s > $o/block_$s.txt 2>&1; tail -1 $o/block_$s.txt; done
