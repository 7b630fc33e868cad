#!/usr/bin/env python3
"""Prefill routes against M per layer shape (development aid; needs an MI355X): fused MFMA dequant-GEMM (gemm_pipe.hip) vs dequantise + in-tree
dense GEMM (gemm_dense.hip) vs dequantise + library GEMM; what hqq_hip_forward_prefers_fused answers.   python tools/prefill_routes.py [nbits]"""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from hqq_amd import _C, ops  # noqa: E402

nbits = int(sys.argv[1]) if len(sys.argv) > 1 else 4
g = torch.Generator().manual_seed(0)


def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (N, K) in ((4096, 4096), (11008, 4096), (4096, 11008)):
    R = N * K // 64
    U = torch.randint(0, 2 ** nbits, (R, 64), generator=g, dtype=torch.uint8).cuda()
    P = ops.pack(nbits, U)
    s = (torch.rand(R, 1, generator=g) * 0.004 + 0.001).half().cuda()
    z = (torch.rand(R, 1, generator=g) * (2 ** nbits - 1)).round().half().cuda()
    base = ops.OPT_META_SCALABLE if ops.meta_scalable(s, z, N, K, 64, nbits) else 0
    for M in (256, 512, 768, 1024, 1536, 2048, 3072, 4096, 8192):
        x = torch.randn(M, K, generator=g).half().cuda()
        y = torch.empty(M, N, device="cuda", dtype=torch.float16)
        a = (x, P, s, z, None, N, K, 64, nbits)
        tf = t(lambda: ops.forward(*a, out=y, fused=True, opts=base))
        td = t(lambda: ops.forward(*a, out=y, fused=False, opts=base))
        tl = t(lambda: ops.forward(*a, out=y, fused=False, opts=base, library_gemm=True))
        pf = _C.lib().hqq_hip_forward_prefers_fused(nbits, M, N, K, 64, 1)
        fl = 2.0 * M * N * K / 1e6
        print(f"{N:5d}x{K:5d} M {M:5d}: fused {tf:8.1f} us ({fl/tf:6.0f} TF)  dense {td:8.1f} us ({fl/td:6.0f} TF)  library {tl:8.1f} us ({fl/tl:6.0f} TF)   prefers_fused={pf}"
              + ("   <-- policy picks the slower" if (pf == 1) != (tf <= td) and abs(tf - td) / min(tf, td) > 0.03 else ""))
