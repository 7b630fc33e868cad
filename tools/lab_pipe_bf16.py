#!/usr/bin/env python3
"""lab: bf16 pipelined GEMM vs bf16 dequantise + library GEMM (device time, graph-captured)"""
import sys, torch
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from hqq_amd import ops
gs, nbits = 64, 4
g = torch.Generator().manual_seed(0)
def t(f, reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps): f()
    for _ in range(10): gr.replay()   # warm replays (clocks ramp with load: a cold first timing reads up to 25 % slow)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (3 * reps)
for N, K in ((4096, 4096), (22016, 4096), (4096, 11008)):
    R = N * K // gs
    P = ops.pack(nbits, torch.randint(0, 16, (R, gs), generator=g, dtype=torch.uint8).cuda())
    s = (torch.rand(R, 1, generator=g) * 0.004 + 0.001).bfloat16().cuda()
    z = (torch.rand(R, 1, generator=g) * 15).round().bfloat16().cuda()
    out = []
    for M in (128, 256, 512, 1024, 8192):
        x = torch.randn(M, K, generator=g).bfloat16().cuda()
        a = t(lambda: ops.gemm(x, P, s, z, None, N, K, gs, nbits))
        b = t(lambda: ops.forward(x, P, s, z, None, N, K, gs, nbits, fused=False))
        out.append(f"M={M}: fused {a:.1f} comp {b:.1f}")
    print(f"{N}x{K}: " + "  ".join(out))
