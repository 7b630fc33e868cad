#!/usr/bin/env python3
"""lab: the three-op exact sequence (subnormal field, mode 2) must give the same bits as the four-op one (mode 0)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from hqq_amd import ops

torch.manual_seed(0)
for nbits in (4, 2, 8, 1):
    for (N, K, gs) in ((4096, 4096, 64), (11008, 4096, 64), (4096, 11008, 64), (512, 1024, 128), (256, 2048, 32)):
        if nbits == 1 and gs != 64: continue
        W = (torch.randn(N, K, device="cuda") * 0.02).half()
        Wq, s, z = ops.quantize(W, nbits=nbits, group_size=gs, round_zero=(nbits == 4))
        s, z = s.half(), z.half()
        for M in (1, 2, 4):
            x = torch.randn(M, K, device="cuda").half()
            y0 = ops.gemv(x, Wq, s, z, None, N, K, gs, nbits, opts=0)
            y2 = ops.gemv(x, Wq, s, z, None, N, K, gs, nbits, opts=ops.OPT_META_SCALABLE)
            torch.cuda.synchronize()
            neq = int((y0.view(torch.int16) != y2.view(torch.int16)).sum())
            print(f"nbits={nbits} N={N} K={K} gs={gs} M={M}: outputs differing {neq} of {y0.numel()}  zmin={float(z.abs().min()):.4g}  meta_check says scalable={ops.meta_scalable(s, z, N, K, gs, nbits)}", flush=True)
# one-hot probes: every weight of a small layer, both modes, vs the dequantise kernel
for nbits in (4, 2, 8):
    N, K = 64, 1024
    W = (torch.randn(N, K, device="cuda") * 0.02).half()
    Wq, s, z = ops.quantize(W, nbits=nbits, group_size=64, round_zero=(nbits == 4))
    s, z = s.half(), z.half()
    Wd = ops.dequantize(Wq, s.reshape(-1), z.reshape(-1), N, K, 64, nbits)
    bad = 0
    for k0 in range(0, K, 4):
        e = torch.zeros(4, K, dtype=torch.float16, device="cuda")
        for i in range(4): e[i, k0 + i] = 1.0
        y = ops.gemv(e, Wq, s, z, None, N, K, 64, nbits, opts=ops.OPT_META_SCALABLE)
        bad += int((y.view(torch.int16) != Wd[:, k0:k0 + 4].t().contiguous().view(torch.int16)).sum())
    print(f"one-hot nbits={nbits}: {bad} of {N*K} weights differ from hqq_hip_dequantize", flush=True)
