#!/bin/bash
# A/B of dense GEMM builds on one box, interleaved, three rounds:  tools/r4_dense_ab.sh <variant> ...
for rep in 1 2 3; do
for v in default "$@"; do
  if [ "$v" = default ]; then unset HQQ_AMD_LIB; else export HQQ_AMD_LIB=$PWD/tools/libhqq_hip_$v.so; fi
  python - "$v" <<'PY'
import sys, torch
from hqq_amd import ops
torch.manual_seed(0)
out = []
for (M, N, K) in ((8192, 4096, 4096), (8192, 11008, 4096), (8192, 4096, 11008)):
    x = torch.randn(M, K, device="cuda").half(); W = (torch.randn(N, K, device="cuda") * 0.02).half()
    y = torch.empty(M, N, device="cuda", dtype=torch.float16)
    for _ in range(10): ops.gemm_dense(x, W, out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): ops.gemm_dense(x, W, out=y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 30
    out.append(2.0 * M * N * K / ms / 1e9)
print(f"{sys.argv[1]:10s} " + "  ".join(f"{v:6.0f}" for v in out) + "  TFLOP/s (8192x4096x4096, 8192x11008x4096, 8192x4096x11008)")
PY
done
done
