#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6m; mkdir -p $OUT
HQQ_AMD_LIB=$PWD/tools/r6/libhqq_hip_prev.so python tools/r6/dbg_tokens.py 3 $OUT/prev.pt 2>/dev/null | tail -1
python tools/r6/dbg_tokens.py 3 $OUT/cur.pt 2>/dev/null | tail -1
HQQ_AMD_LIB=$PWD/tools/r6/libhqq_hip_prev.so python -m pytest tests/test_model_gpu.py -m gpu -q -k "emits_the_tokens and 3-folded" 2>&1 | tail -2
