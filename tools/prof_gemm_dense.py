#!/usr/bin/env python3
"""The in-tree dense GEMM (or the library's, `lib`) back to back, for rocprofv3 (development aid; needs an MI355X).
    python tools/prof_gemm_dense.py M N K [lib]"""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from hqq_amd import ops  # noqa: E402

M, N, K = (int(v) for v in sys.argv[1:4])
lib = len(sys.argv) > 4
torch.manual_seed(0)
x = torch.randn(M, K, device="cuda").half()
W = (torch.randn(N, K, device="cuda") * 0.02).half()
y = torch.empty(M, N, device="cuda", dtype=torch.float16)
for _ in range(30):
    if lib:
        torch.matmul(x, W.t(), out=y)
    else:
        ops.gemm_dense(x, W, out=y)
torch.cuda.synchronize()
