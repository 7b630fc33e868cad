#!/bin/bash
OUT=gpurun_out/r6b; mkdir -p $OUT
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $OUT/pytest.log
bash tools/r6/ab.sh $OUT/ab.txt "" base cur
bash tools/r6/ab.sh $OUT/ab.txt "--nbits 2" base cur
bash tools/r6/ab.sh $OUT/ab.txt "--nbits 3" base cur
cat $OUT/pytest.log $OUT/ab.txt
