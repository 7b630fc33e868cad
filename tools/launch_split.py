#!/usr/bin/env python3
"""lab: time per launch type of the 7B decode step (graph of 32 dependent launches over distinct layers), modes exact / sub / factored."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from hqq_amd import ops
dev = torch.device("cuda")
nbits = int(sys.argv[1]) if len(sys.argv) > 1 else 4
SH = {"q|k|v": [(4096, 4096)] * 3, "o": [(4096, 4096)], "gate|up": [(11008, 4096)] * 2, "down": [(4096, 11008)], "gate alone": [(11008, 4096)]}


def qlayer(N, K, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    W = (torch.randn(N, K, device=dev, generator=g) * 0.02).half()
    Wq, s, z = ops.quantize(W, nbits=nbits, group_size=64, round_zero=(nbits == 4))
    return Wq, s.half(), z.half()


for name, shapes in SH.items():
    K = shapes[0][1]
    reps = 32
    groups = [[(N,) + qlayer(N, K, 100 * r + i) for i, (N, _) in enumerate(shapes)] for r in range(reps)]
    x = torch.randn(1, K, device=dev).half()
    outs = [torch.empty(1, N, device=dev, dtype=torch.float16) for (N, _) in shapes]
    nb = sum(N * K * nbits // 8 + 4 * (N * K // 64) + 2 * K + 2 * N for (N, _) in shapes)
    line = f"{name:10s} {nb/1e6:6.1f} MB:"
    for mode, o in ((0, 0), (2, ops.OPT_META_SCALABLE), (1, ops.OPT_FACTORED)):
        def step():
            for g in groups:
                ops.gemv_grouped(x, [(L[1], L[2], L[3], None, L[0]) for L in g], K, 64, nbits, outs=outs, opts=o)
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s): step()
        torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr): step()
        for _ in range(3): gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): gr.replay()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 / reps * 1e3
        line += f"  mode{mode} {us:6.2f} us ({nb/us/1e6:5.2f} TB/s)"
    print(line, flush=True)
    del groups
