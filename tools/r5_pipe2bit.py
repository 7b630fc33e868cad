#!/usr/bin/env python3
"""lab: the pipelined fused GEMM at 2 bits (4-wave tile), us per call on the 7B shapes — A/B of the 256-register build (spills 14-80 dwords into
scratch inside the loop) against -DGD_2BIT_ONE_WAVE_PER_SIMD=1 (512-register budget, accumulators in AGPRs).  HQQ_AMD_LIB selects the build."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hqq_amd import ops
g = torch.Generator().manual_seed(0)
print("lib", os.environ.get("HQQ_AMD_LIB", "default"))
for N, K in ((4096, 4096), (11008, 4096), (4096, 11008)):
    R = N * K // 64
    P = ops.pack(2, torch.randint(0, 4, (R, 64), generator=g, dtype=torch.uint8).cuda())
    s = (torch.rand(R, 1, generator=g) * 0.004 + 0.001).half().cuda()
    z = (torch.rand(R, 1, generator=g) * 3).round().half().cuda()
    sub = ops.OPT_META_SCALABLE if ops.meta_scalable(s, z, N, K, 64, 2) else 0
    row = []
    for M in (128, 512, 1024, 2048):
        x = torch.randn(M, K, generator=g).half().cuda()
        y = torch.empty(M, N, dtype=torch.float16, device="cuda")
        for nw, bits in (("4w", ops.OPT_GEMM_NARROW), ("8w", ops.OPT_GEMM_WIDE)):
            f = lambda: ops.gemm(x, P, s, z, None, N, K, 64, 2, out=y, opts=sub | bits)
            for _ in range(3): f()
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for _ in range(10): f()
            for _ in range(5): gr.replay()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); gr.replay(); gr.replay(); e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / 20)
            row.append(f"M={M}/{nw} {best:.1f}")
    print(f"{N}x{K}: " + "  ".join(row), flush=True)
