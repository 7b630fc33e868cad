#!/bin/bash
# the dense GEMM's loop with parts removed (lab builds, wrong results): what each part costs
for v in default "$@"; do
  if [ "$v" = default ]; then unset HQQ_AMD_LIB; else export HQQ_AMD_LIB=$PWD/tools/libhqq_hip_$v.so; fi
  python - "$v" <<'PY'
import sys, torch
from hqq_amd import ops
torch.manual_seed(0)
out = []
for (M, N, K) in ((8192, 4096, 4096), (8192, 4096, 11008)):
    x = torch.randn(M, K, device="cuda").half(); W = (torch.randn(N, K, device="cuda") * 0.02).half()
    y = torch.empty(M, N, device="cuda", dtype=torch.float16)
    for _ in range(3): ops.gemm_dense(x, W, out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.gemm_dense(x, W, out=y)
    e1.record(); torch.cuda.synchronize()
    out.append(e0.elapsed_time(e1) / 10 * 1e3)
print(f"{sys.argv[1]:10s} K=4096: {out[0]:7.1f} us   K=11008: {out[1]:7.1f} us   per K tile pass {(out[1]-out[0])/108:6.3f} us")
PY
done
