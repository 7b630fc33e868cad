#!/usr/bin/env python3
"""A few pipelined-GEMM cases back to back, for rocprofv3 --kernel-trace --stats (development aid; needs an MI355X).
    python tools/prof_gemm_pipe.py N K M KS [nbits]"""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from hqq_amd import ops  # noqa: E402

N, K, M, KS = (int(v) for v in sys.argv[1:5])
nbits = int(sys.argv[5]) if len(sys.argv) > 5 else 4
gs = 64
g = torch.Generator().manual_seed(0)
R = N * K // gs
U = torch.randint(0, 2 ** nbits, (R, gs), generator=g, dtype=torch.uint8).cuda()
P = ops.pack(nbits, U)
s = (torch.rand(R, 1, generator=g) * 0.004 + 0.001).half().cuda()
z = (torch.rand(R, 1, generator=g) * (2 ** nbits - 1)).round().half().cuda()
base = ops.OPT_META_SCALABLE if ops.meta_scalable(s, z, N, K, gs, nbits) else 0
x = torch.randn(M, K, generator=g).half().cuda()
for _ in range(50):
    ops.gemm(x, P, s, z, None, N, K, gs, nbits, opts=base | (KS << 24))
torch.cuda.synchronize()
