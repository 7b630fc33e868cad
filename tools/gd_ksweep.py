#!/usr/bin/env python3
"""Dense GEMM time against K at fixed M, N (development aid; needs an MI355X): the intercept is the per-tile fixed cost.
    python tools/gd_ksweep.py [M N]"""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from hqq_amd import ops  # noqa: E402

M, N = (int(v) for v in sys.argv[1:3]) if len(sys.argv) > 2 else (8192, 4096)
torch.manual_seed(0)
for K in (64, 256, 1024, 2048, 4096, 8192, 11008):
    x = torch.randn(M, K, device="cuda").half()
    W = (torch.randn(N, K, device="cuda") * 0.02).half()
    y = torch.empty(M, N, device="cuda", dtype=torch.float16)
    row = []
    for fn in (lambda: ops.gemm_dense(x, W, out=y), lambda: torch.matmul(x, W.t(), out=y)):
        for _ in range(10): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): fn()
        e1.record(); torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) / 30 * 1e3)
    print(f"M {M} N {N} K {K:6d}: in-tree {row[0]:8.1f} us   library {row[1]:8.1f} us")
