#!/bin/bash
OUT=gpurun_out/r6c; mkdir -p $OUT
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 ) > $OUT/pytest.log
bash tools/r6/ab.sh $OUT/ab.txt "" base cur u1n2 u1n3 u1n4 u2n3
bash tools/r6/ab.sh $OUT/ab.txt "--nbits 2" base cur u1n3 u2n3
cat $OUT/pytest.log $OUT/ab.txt
