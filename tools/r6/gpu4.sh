#!/bin/bash
OUT=gpurun_out/r6d; mkdir -p $OUT
bash tools/r6/ab.sh $OUT/ab.txt "" base cur wg3 wg5 wg6 wg7
bash tools/r6/ab.sh $OUT/ab.txt "--nbits 2" base cur wg5 wg6
cat $OUT/ab.txt
