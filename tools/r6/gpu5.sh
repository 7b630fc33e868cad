#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6e; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o t -- python $GRAFT_REPO_ROOT/bench.py --no-legs --no-cpu-baseline --steps 30 --warmup 5 > $OUT/bench.json 2>$OUT/err.txt
cd $GRAFT_REPO_ROOT
python tools/r6/per_launch.py $OUT/prof > $OUT/per_launch.txt 2>&1
head -c 600 $OUT/bench.json; echo; cat $OUT/per_launch.txt
find $OUT/prof -name "*.csv" -size +1M -delete
