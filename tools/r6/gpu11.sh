#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6k; mkdir -p $OUT
for v in dx4 dx6 cur; do
  if [ "$v" = cur ]; then unset HQQ_AMD_LIB; else export HQQ_AMD_LIB=$PWD/tools/r6/libhqq_hip_$v.so; fi
  echo "== $v" >> $OUT/bs128.txt
  python tools/r6/bs128.py >> $OUT/bs128.txt 2>$OUT/err_$v.txt
done
unset HQQ_AMD_LIB
( timeout 900 python -m pytest tests/test_gemm_pipe_gpu.py -m gpu -x -q 2>&1 | tail -4 ) >> $OUT/bs128.txt
cat $OUT/bs128.txt; tail -2 $OUT/err_cur.txt
