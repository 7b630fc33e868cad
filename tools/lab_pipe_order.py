#!/usr/bin/env python3
"""lab: the built-in plan against the forced tile it should equal, measured alternately (development aid)"""
import sys, torch
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from hqq_amd import ops
gs, nbits, N, K, M = 64, 4, 12288, 4096, 1024
g = torch.Generator().manual_seed(0)
R = N * K // gs
P = ops.pack(nbits, torch.randint(0, 16, (R, gs), generator=g, dtype=torch.uint8).cuda())
s = (torch.rand(R, 1, generator=g) * 0.004 + 0.001).half().cuda()
z = (torch.rand(R, 1, generator=g) * 15).round().half().cuda()
x = torch.randn(M, K, generator=g).half().cuda()
y = torch.empty(M, N, dtype=torch.float16, device="cuda")
def t(o, reps=10):
    f = lambda: ops.gemm(x, P, s, z, None, N, K, gs, nbits, out=y, opts=o)
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
B = ops.OPT_META_SCALABLE
for name, o in (("auto", B), ("8x256/KS1", B | 192 | (1 << 24)), ("auto", B), ("8x256 (KS by plan)", B | 192), ("8x128/KS1", B | 128 | (1 << 24)), ("auto", B), ("4x128/KS1", B | 64 | (1 << 24))):
    print(f"{name}: {t(o):.1f}")
