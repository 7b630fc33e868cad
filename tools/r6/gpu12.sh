#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6l; mkdir -p $OUT
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > $OUT/pytest.log
bash tools/r6/ab.sh $OUT/ab.txt "" base prev cur
bash tools/r6/ab.sh $OUT/ab.txt "--nbits 2" base prev cur
bash tools/r6/ab.sh $OUT/ab.txt "--nbits 3" base prev cur
cat $OUT/pytest.log $OUT/ab.txt
