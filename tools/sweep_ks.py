"""K-split sweep for the skinny GEMM (opts=OPT_SKINNY_KS(n) overrides the built-in rule): device us per launch, graph replay over a
pool of distinct layers (> 256 MiB), int4 gs=64 fp16.  Usage: python tools/sweep_ks.py [M ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hqq_amd import ops  # noqa: E402

LAUNCHES = {"o": [(4096, 4096)], "q|k|v": [(4096, 4096)] * 3, "gate|up": [(11008, 4096)] * 2, "down": [(4096, 11008)]}


def layer(N, K, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    Wq = torch.randint(0, 256, (N // 2, K), device="cuda", dtype=torch.uint8, generator=g)
    s = (torch.rand(N * K // 64, 1, device="cuda", generator=g) * 0.004 + 0.001).half()
    z = (torch.rand(N * K // 64, 1, device="cuda", generator=g) * 15).half()
    return Wq, s, z


def timed(fn, n, reps=5):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (reps * n)


def main():
    Ms = [int(v) for v in sys.argv[1:]] or [32]
    for name, shapes in LAUNCHES.items():
        K = shapes[0][1]
        nbytes = sum(N * K // 2 for N, _ in shapes)
        pool_n = max(4, int(400e6 / nbytes) + 1)
        pool = [[layer(N, K, 100 * i + j) for j, (N, _) in enumerate(shapes)] for i in range(pool_n)]
        for M in Ms:
            x = torch.randn(M, K, device="cuda", dtype=torch.float16)
            outs = [torch.empty(M, N, device="cuda", dtype=torch.float16) for N, _ in shapes]

            def sweep(ks):
                for Ls in pool:
                    ops.gemv_grouped(x, [(W, s, z, None, N) for (W, s, z), (N, _) in zip(Ls, shapes)], K, 64, 4, outs=outs, opts=ops.OPT_SKINNY_KS(ks))
            row = []
            for ks in (0, 1, 2, 3, 4, 6, 8, 11, 16):
                if ks > K // 256: continue
                row.append(f"{'rule' if not ks else ks}:{timed(lambda: sweep(ks), pool_n):6.2f}")
            print(f"{name:8s} M={M:3d}  " + "  ".join(row), flush=True)
        del pool


if __name__ == "__main__":
    main()
