#!/usr/bin/env python3
"""lab: every (tile, K-split) plan of the pipelined GEMM for a few shapes, against the built-in choice (development aid; needs an MI355X)"""
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
    best = None
    for _ in range(3):   # best of three timing passes
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): gr.replay()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (3 * reps)
        best = us if best is None or us < best else best
    return best
TILES = {"4x128": ops.OPT_GEMM_NARROW, "8x128": ops.OPT_GEMM_WIDE, "8x256": ops.OPT_GEMM_NARROW | ops.OPT_GEMM_WIDE}
for N, K in ((4096, 4096), (12288, 4096), (11008, 4096), (22016, 4096), (4096, 11008)):
    R = N * K // gs
    P = ops.pack(nbits, torch.randint(0, 16, (R, gs), generator=g, dtype=torch.uint8).cuda())
    s = (torch.rand(R, 1, generator=g) * 0.004 + 0.001).half().cuda()
    z = (torch.rand(R, 1, generator=g) * 15).round().half().cuda()
    for M in (int(a) for a in sys.argv[1:]) if len(sys.argv) > 1 else (768, 1024, 2048):
        x = torch.randn(M, K, generator=g).half().cuda()
        y = torch.empty(M, N, dtype=torch.float16, device="cuda")
        auto = t(lambda: ops.gemm(x, P, s, z, None, N, K, gs, nbits, out=y, opts=ops.OPT_META_SCALABLE))
        comp = t(lambda: ops.forward(x, P, s, z, None, N, K, gs, nbits, fused=False))
        row = []
        for name, bits in TILES.items():
            for ks in (1, 2, 3, 4, 6, 8, 16):
                try:
                    row.append(f"{name}/KS{ks} {t(lambda: ops.gemm(x, P, s, z, None, N, K, gs, nbits, out=y, opts=ops.OPT_META_SCALABLE | bits | (ks << 24))):.0f}")
                except Exception as e:
                    row.append(f"{name}/KS{ks} n/a")
        print(f"{N}x{K} M={M}: plan {auto:.0f} composition {comp:.0f} | " + "  ".join(row), flush=True)
