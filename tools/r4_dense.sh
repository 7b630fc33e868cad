#!/bin/bash
timeout 900 python -m pytest tests/test_gemm_dense_gpu.py -x -q -m gpu --tb=short -n 4 2>&1 | tail -n 25
python - <<'PY'
import torch, time
from hqq_amd import ops
torch.manual_seed(0)
for (M, N, K) in ((8192, 4096, 4096), (8192, 11008, 4096), (8192, 4096, 11008), (4096, 4096, 4096), (2048, 4096, 4096)):
    x = torch.randn(M, K, device="cuda").half()
    W = (torch.randn(N, K, device="cuda") * 0.02).half()
    y = torch.empty(M, N, device="cuda", dtype=torch.float16)
    for name, fn in (("in-tree", lambda: ops.gemm_dense(x, W, out=y)), ("library", lambda: torch.matmul(x, W.t(), out=y))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{M}x{N}x{K} {name}: {ms*1e3:.1f} us  {2.0*M*N*K/ms/1e9:.0f} TFLOP/s")
PY
