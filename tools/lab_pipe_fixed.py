#!/usr/bin/env python3
"""lab: fixed cost of the pipelined GEMM — time vs number of K steps at a fixed grid (development aid; needs an MI355X)"""
import sys, torch
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from hqq_amd import ops
gs, nbits = 64, 4
g = torch.Generator().manual_seed(0)
def run(M, N, K, KS):
    R = N * K // gs
    P = ops.pack(nbits, torch.randint(0, 16, (R, gs), generator=g, dtype=torch.uint8).cuda())
    s = (torch.rand(R, 1, generator=g) * 0.004 + 0.001).half().cuda()
    z = (torch.rand(R, 1, generator=g) * 15).round().half().cuda()
    x = torch.randn(M, K, generator=g).half().cuda()
    y = torch.empty(M, N, dtype=torch.float16, device="cuda")
    f = lambda: ops.gemm(x, P, s, z, None, N, K, gs, nbits, out=y, opts=ops.OPT_META_SCALABLE | (KS << 24))
    for _ in range(3): f()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(20): f()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 60
for M, N in ((128, 4096), (128, 32768), (1024, 4096)):
    print(f"M={M} N={N} KS=1:", "  ".join(f"K={K}: {run(M, N, K, 1):.1f}" for K in (128, 512, 1024, 2048, 4096)))
