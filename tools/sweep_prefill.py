#!/usr/bin/env python3
"""Prefill / batched-decode sweep: fused dequant-GEMM kernels vs HIP dequantise + library GEMM, per activation-row count
(development aid; needs an MI355X).
    python tools/sweep_prefill.py [nbits] [out.md]"""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from hqq_amd import ops  # noqa: E402

SHAPES = [("o 4096x4096", 4096, 4096), ("q|k|v 12288x4096", 12288, 4096), ("gate|up 22016x4096", 22016, 4096), ("down 4096x11008", 4096, 11008)]
MS = [65, 96, 128, 192, 256, 384, 512, 768, 1024, 2048, 4096, 8192]


def timeit(fn, reps):
    """device time per call: `reps` calls captured in one hipGraph (Python / ctypes launch overhead, ~10-20 us per call, stays out)"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    for _ in range(max(2, int(3000 / (reps * 30)))):   # warm replays (clocks ramp with load: a cold first timing reads up to 25 % slow)
        g.replay()
    torch.cuda.synchronize()
    best = None
    for _ in range(3):   # best of three timing passes
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (3 * reps)
        best = us if best is None or us < best else best
    return best


def main():
    nbits = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    if len(sys.argv) > 2:
        import os
        os.makedirs(os.path.dirname(os.path.abspath(sys.argv[2])), exist_ok=True)
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    gs = 64
    g = torch.Generator().manual_seed(0)
    ksweep = len(sys.argv) > 3 and sys.argv[3].startswith("ks")
    wide = {"ks": 0, "ks4": ops.OPT_GEMM_NARROW, "ks8": ops.OPT_GEMM_WIDE}.get(sys.argv[3] if len(sys.argv) > 3 else "ks", 0)
    print(f"## int{nbits} fp16, us per call (TFLOP/s): fused kernels vs dequantise + library GEMM\n", file=out)
    if ksweep:
        KSS = [1, 2, 4, 8, 16]
        print("| layer | M | rule | " + " | ".join(f"KS={k}" for k in KSS) + " | max err vs fp32 |", file=out)
        print("| --- | --- | --- | " + " | ".join("---" for _ in KSS) + " | --- |", file=out)
    else:
        print("| layer | M | fused pipelined | fused tile (round 1) | dequant + GEMM | dequant alone | library GEMM alone |", file=out)
        print("| --- | --- | --- | --- | --- | --- | --- |", file=out)
    for name, N, K in SHAPES:
        R = N * K // gs
        U = torch.randint(0, 2 ** nbits, (R, gs), generator=g, dtype=torch.uint8).cuda()
        P = ops.pack(nbits, U)
        s = (torch.rand(R, 1, generator=g) * 0.004 + 0.001).half().cuda()
        z = (torch.rand(R, 1, generator=g) * (2 ** nbits - 1)).round().half().cuda()
        Wd = ops.dequantize(P, s.reshape(-1), z.reshape(-1), N, K, gs, nbits, 1)
        base = ops.OPT_META_SCALABLE if ops.meta_scalable(s, z, N, K, gs, nbits) else 0
        for M in MS:
            x = torch.randn(M, K, generator=g).half().cuda()
            fl = 2.0 * M * N * K
            reps = 20 if M <= 1024 else 5

            def t(fn):
                try:
                    us = timeit(fn, reps)
                    return f"{us:.1f} ({fl / us / 1e6:.0f})"
                except Exception as e:   # noqa: BLE001
                    return "n/a " + str(e)[:30]
            if ksweep:
                if M > 1024:
                    continue
                y = ops.gemm(x, P, s, z, None, N, K, gs, nbits, opts=base)
                ref = x.float() @ Wd.float().t()
                err = ((y.float() - ref).abs().max() / ref.abs().max()).item()
                cols = [t(lambda: ops.gemm(x, P, s, z, None, N, K, gs, nbits, opts=base))]
                cols += [t(lambda k=k: ops.gemm(x, P, s, z, None, N, K, gs, nbits, opts=base | wide | (k << 24))) for k in KSS]
                print(f"| {name} | {M} | " + " | ".join(cols) + f" | {err:.1e} |", file=out)
            else:
                a = t(lambda: ops.gemm(x, P, s, z, None, N, K, gs, nbits, opts=base))
                b = t(lambda: ops.gemm(x, P, s, z, None, N, K, gs, nbits, opts=base | ops.OPT_GEMM_CLASSIC))
                c = t(lambda: ops.forward(x, P, s, z, None, N, K, gs, nbits, fused=False))
                d = t(lambda: ops.dequantize(P, s.reshape(-1), z.reshape(-1), N, K, gs, nbits, 1))
                e = t(lambda: torch.matmul(x, Wd.t()))
                print(f"| {name} | {M} | {a} | {b} | {c} | {d} | {e} |", file=out)
            out.flush()
        del U, P, s, z, Wd
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
