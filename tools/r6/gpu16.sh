#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6p; mkdir -p $OUT
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 ) > $OUT/pytest.log
HQQ_AMD_LIB=$PWD/tools/r6/libhqq_hip_prev.so python tools/r6/dbg_bits2.py $OUT/prev.pt 2>/dev/null | tail -1
python tools/r6/dbg_bits2.py $OUT/cur.pt 2>/dev/null | tail -1
python - <<PY
import torch
a=torch.load("$OUT/prev.pt"); b=torch.load("$OUT/cur.pt")
n=0
for k in a:
    A = a[k] if isinstance(a[k], list) else [a[k]]; B = b[k] if isinstance(b[k], list) else [b[k]]
    n += sum(int((u != v).sum()) for u, v in zip(A, B))
print("prev vs cur differing elements over", len(a), "cases:", n)
PY
bash tools/r6/ab.sh $OUT/ab.txt "" prev prev2 cur
bash tools/r6/ab.sh $OUT/ab.txt "--nbits 2" prev prev2 cur
cat $OUT/pytest.log $OUT/ab.txt
