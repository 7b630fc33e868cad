#!/usr/bin/env python3
"""one batched-decode layer shape in a loop, for rocprofv3 (kernel trace / PMC passes):  python tools/prof_skinny.py [N K M]"""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from hqq_amd import ops  # noqa: E402
from tools.microbench import rand_layer  # noqa: E402

N, K, M = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (11008, 4096, 32)
pool = [rand_layer(N, K, 4) for _ in range(30)]
x = torch.randn(M, K, device="cuda", dtype=torch.float16)
y = torch.empty(M, N, device="cuda", dtype=torch.float16)
for _ in range(2):
    for W in pool:
        ops.gemv(x, W[0], W[1], W[2], None, N, K, 64, 4, out=y)
torch.cuda.synchronize()
