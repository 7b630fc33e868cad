"""3-bit decode launches of a Llama-2-7B block: device us per launch for the slab-sharing kernel (gemv3s.hip) and the
row-per-wave kernel (gemv3.hip, opts=OPT_GEMV3_ROWWISE); graph replay over a pool of distinct layers.  Usage: python tools/sweep_int3.py [M ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hqq_amd import ops  # noqa: E402

LAUNCHES = {"o": [(4096, 4096)], "q|k|v": [(4096, 4096)] * 3, "gate|up": [(11008, 4096)] * 2, "down": [(4096, 11008)]}


def layer(N, K, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    R = N * K // 64
    Wq = torch.randint(0, 2 ** 30, ((R + 9) // 10, 64), device="cuda", dtype=torch.int32, generator=g)
    s = (torch.rand(R, 1, device="cuda", generator=g) * 0.004 + 0.001).half()
    z = (torch.rand(R, 1, device="cuda", generator=g) * 7).half()
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
    Ms = [int(v) for v in sys.argv[1:]] or [1]
    for name, shapes in LAUNCHES.items():
        K = shapes[0][1]
        nbytes = sum(4 * 64 * ((N * K // 64 + 9) // 10) for N, _ in shapes)
        pool_n = max(4, int(400e6 / nbytes) + 1)
        pool = [[layer(N, K, 100 * i + j) for j, (N, _) in enumerate(shapes)] for i in range(pool_n)]
        for M in Ms:
            x = torch.randn(M, K, device="cuda", dtype=torch.float16)
            outs = [torch.empty(M, N, device="cuda", dtype=torch.float16) for N, _ in shapes]

            def sweep(opts):
                for Ls in pool:
                    ops.gemv_grouped(x, [(W, s, z, None, N) for (W, s, z), (N, _) in zip(Ls, shapes)], K, 64, 3, outs=outs, opts=opts)
            # force each kernel in turn (the library picks by launch size otherwise)
            t2 = timed(lambda: sweep(ops.OPT_GEMV3_SLABS), pool_n)
            t1 = timed(lambda: sweep(ops.OPT_GEMV3_ROWWISE), pool_n)
            print(f"{name:8s} M={M}  {nbytes / 1e6:6.1f} MB   slab-sharing {t2:6.2f} us ({nbytes / t2 / 1e6:5.2f} TB/s)   row-per-wave {t1:6.2f} us ({nbytes / t1 / 1e6:5.2f} TB/s)", flush=True)
        del pool


if __name__ == "__main__":
    main()
