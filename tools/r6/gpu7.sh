#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6g; mkdir -p $OUT
bash tools/r6/ab.sh $OUT/ab.txt "--nbits 2" base nopad cur
bash tools/r6/ab.sh $OUT/ab.txt "" base nopad cur
bash tools/r6/ab.sh $OUT/ab.txt "--nbits 3" base nopad cur
cat $OUT/ab.txt
