#!/usr/bin/env python3
"""lab: a fixed set of pipelined-GEMM cases, device time per call (graph-captured) — for A/B runs of library variants (HQQ_AMD_LIB)"""
import sys, torch
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from hqq_amd import ops
gs, nbits = 64, 4
g = torch.Generator().manual_seed(0)
CASES = [(2048, 4096, 4096), (8192, 4096, 4096), (8192, 12288, 4096), (4096, 22016, 4096), (8192, 22016, 4096), (8192, 4096, 11008)] if len(sys.argv) > 2 else [(128, 4096, 4096), (128, 22016, 4096), (256, 12288, 4096), (512, 4096, 4096), (512, 22016, 4096), (1024, 4096, 4096), (1024, 12288, 4096), (1024, 4096, 11008)]
res = []
for M, N, K in CASES:
    R = N * K // gs
    P = ops.pack(nbits, torch.randint(0, 16, (R, gs), generator=g, dtype=torch.uint8).cuda())
    s = (torch.rand(R, 1, generator=g) * 0.004 + 0.001).half().cuda()
    z = (torch.rand(R, 1, generator=g) * 15).round().half().cuda()
    x = torch.randn(M, K, generator=g).half().cuda()
    y = torch.empty(M, N, dtype=torch.float16, device="cuda")
    extra = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    f = lambda: ops.gemm(x, P, s, z, None, N, K, gs, nbits, out=y, opts=ops.OPT_META_SCALABLE | extra)
    for _ in range(3): f()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(10): f()
    for _ in range(10): gr.replay()   # warm replays (clocks ramp with load: a cold first timing reads up to 25 % slow)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): gr.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 30
    ref = x.float() @ ops.dequantize(P, s.reshape(-1), z.reshape(-1), N, K, gs, nbits, 1).float().t()
    err = ((y.float() - ref).abs().max() / ref.abs().max()).item()
    res.append(f"{M}x{N}x{K}: {us:.1f} ({2.0 * M * N * K / us / 1e6:.0f}){'' if err < 1e-3 else ' ERR %.1e' % err}")
    del P, s, z
print("  ".join(res))
