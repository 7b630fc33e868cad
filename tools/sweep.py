#!/usr/bin/env python3
"""Round sweep over BASELINE.json configs[1..4] on one MI355X -> markdown (profiles/rNN_sweep.md).
    python tools/sweep.py > profiles/r01_sweep.md
All timings: device time from one hipGraph replay over a rotating pool of distinct layers > 256 MiB (no host launch cost, no
Infinity-Cache hits).  70B shards: the per-rank packed-row block of hqq_amd.shard (what rank r of P would run), on this one GPU."""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from hqq_amd import ops  # noqa: E402
from tools.microbench import gemv_bytes, graph_time, rand_layer  # noqa: E402

HBM = 8.0e12


def gemv_time(N, K, nbits, M=1, n_group=1):
    nbytes = n_group * gemv_bytes(N, K, nbits, M)
    pool_n = max(3, int(500e6 / nbytes) + 1)
    pool = [[rand_layer(N, K, nbits) for _ in range(n_group)] for _ in range(pool_n)]
    x = torch.randn(M, K, device="cuda", dtype=torch.float16)
    ys = [torch.empty(M, N, device="cuda", dtype=torch.float16) for _ in range(n_group)]

    def sweep():
        for grp in pool:
            ops.gemv_grouped(x, [(Wq, s, z, None, N) for (Wq, s, z) in grp], K, 64, nbits, outs=ys)
    t = graph_time(sweep, pool_n)
    del pool
    return t, nbytes


def lib_time(N, K, M, n_group=1):
    """what HQQBackend.PYTORCH does on the GPU, minus its temporaries: HIP dequantise kernel + torch.matmul (hipBLASLt)"""
    nbytes = n_group * gemv_bytes(N, K, 4, M)
    pool_n = max(3, int(500e6 / nbytes) + 1)
    pool = [[rand_layer(N, K, 4) for _ in range(n_group)] for _ in range(pool_n)]
    x = torch.randn(M, K, device="cuda", dtype=torch.float16)

    def sweep():
        for grp in pool:
            for (Wq, s, z) in grp:
                torch.matmul(x, ops.dequantize(Wq, s.reshape(-1), z.reshape(-1), N, K, 64, 4).t())
    t = graph_time(sweep, pool_n)
    del pool
    return t


def row(*c):
    print("| " + " | ".join(str(v) for v in c) + " |")


def main():
    print(f"# Sweep on {torch.cuda.get_device_name(0)} (torch {torch.__version__}), {time.strftime('%Y-%m-%d')}\n")
    print("## configs[1] — bs=1 fused dequant-GEMV at Llama-2-7B shapes (fp16, gs=64)\n")
    row("nbits", "mode", "launch", "µs", "GB/s", "% of 8 TB/s")
    row("---", "---", "---", "---", "---", "---")
    for mode, mname in ((ops.GEMV_EXACT, "exact"), (ops.GEMV_FACTORED, "factored")):
        ops.set_gemv_mode(mode)
        for nbits in (4, 2):
            for label, N, K, g in (("o 4096x4096", 4096, 4096, 1), ("q|k|v 3x4096x4096", 4096, 4096, 3), ("gate|up 2x11008x4096", 11008, 4096, 2),
                                   ("down 4096x11008", 4096, 11008, 1)):
                t, nb = gemv_time(N, K, nbits, 1, g)
                row(nbits, mname, label, f"{t * 1e6:.2f}", f"{nb / t / 1e9:.0f}", f"{nb / t / HBM * 100:.1f}")
    ops.set_gemv_mode(ops.GEMV_EXACT)
    for label, N, K, g in (("o 4096x4096", 4096, 4096, 1), ("q|k|v 3x4096x4096", 4096, 4096, 3), ("gate|up 2x11008x4096", 11008, 4096, 2),
                           ("down 4096x11008", 4096, 11008, 1)):
        t, nb = gemv_time(N, K, 3, 1, g)
        row(3, "exact", label, f"{t * 1e6:.2f}", f"{nb / t / 1e9:.0f}", f"{nb / t / HBM * 100:.1f}")
    print("\n## decode with a batch (exact weights, int4): fused kernels vs dequantise kernel + hipBLASLt\n")
    row("launch", "M", "fused µs", "GB/s", "tok/s of this launch", "kernel", "dequantise + library GEMM µs")
    row("---", "---", "---", "---", "---", "---", "---")
    for label, N, K, g in (("o 4096x4096", 4096, 4096, 1), ("q|k|v 3x4096x4096", 4096, 4096, 3), ("gate|up 2x11008x4096", 11008, 4096, 2),
                           ("down 4096x11008", 4096, 11008, 1)):
        for M in (1, 2, 4, 8, 16, 32, 64):
            t, nb = gemv_time(N, K, 4, M, g)
            tl = lib_time(N, K, M, g) if M >= 8 else None
            kern = "row-per-wave diag-MFMA (gemv.hip)" if M <= 4 else "skinny GEMM (skinny.hip)"
            row(label, M, f"{t * 1e6:.2f}", f"{nb / t / 1e9:.0f}", f"{M / t:.0f}", kern, "-" if tl is None else f"{tl * 1e6:.2f}")
    print("\n## configs[4] — Llama-2-70B shapes, per-rank shard of an output-column shard over P GPUs (bs=1 and bs=32, int4, exact)\n")
    row("layer", "P", "shard N x K", "bs=1 µs", "GB/s per GPU", "all-gather payload per rank (bs=1)", "bs=32 µs", "GB/s per GPU")
    row("---", "---", "---", "---", "---", "---", "---", "---")
    for label, N, K in (("q/o 8192x8192", 8192, 8192), ("k/v 1024x8192", 1024, 8192), ("gate/up 28672x8192", 28672, 8192), ("down 8192x28672", 8192, 28672)):
        for P in (1, 2, 4, 8):
            if (N // 2) % P:
                continue
            t, nb = gemv_time(N // P, K, 4, 1)
            t32, nb32 = gemv_time(N // P, K, 4, 32)
            row(label, P, f"{N // P}x{K}", f"{t * 1e6:.2f}", f"{nb / t / 1e9:.0f}", f"{2 * N // P} B", f"{t32 * 1e6:.2f}", f"{nb32 / t32 / 1e9:.0f}")
    print("\n## configs[3] — pack / dequantise / solver per layer (fp16 weights N(0, 0.02^2))\n")
    row("nbits", "shape", "quantize (solver 20 it + pack) ms", "G elem-iter/s", "dequantize µs", "dequant GB/s")
    row("---", "---", "---", "---", "---", "---")
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for (N, K) in ((4096, 4096), (11008, 4096)):
        W = (torch.randn(N, K, device="cuda") * 0.02).half()
        for nbits in (4, 3, 2):
            ops.quantize(W, nbits=nbits, group_size=64, round_zero=(nbits == 4))
            a.record()
            for _ in range(3):
                Wq, s, z = ops.quantize(W, nbits=nbits, group_size=64, round_zero=(nbits == 4))
            b.record(); torch.cuda.synchronize()
            tq = a.elapsed_time(b) / 3 * 1e-3
            s16, z16 = s.half().reshape(-1), z.half().reshape(-1)
            ops.dequantize(Wq, s16, z16, N, K, 64, nbits)
            a.record()
            for _ in range(10):
                ops.dequantize(Wq, s16, z16, N, K, 64, nbits)
            b.record(); torch.cuda.synchronize()
            td = a.elapsed_time(b) / 10 * 1e-3
            row(nbits, f"{N}x{K}", f"{tq * 1e3:.2f}", f"{N * K * 20 / tq / 1e9:.0f}", f"{td * 1e6:.1f}", f"{(2 * N * K + Wq.numel() * Wq.element_size()) / td / 1e9:.0f}")
    print("\n## configs[3] — stand-alone bit-packing kernels (BitPack.pack_* / unpack_*), 4096x4096 (uint8 levels in, fp16 out of unpack)\n")
    row("nbits", "pack µs", "pack GB/s (read levels + write packed)", "unpack µs", "unpack GB/s (read packed + write fp16)")
    row("---", "---", "---", "---", "---")
    R = 4096 * 4096 // 64
    for nbits in (4, 3, 2):
        U = torch.randint(0, 2 ** nbits, (R, 64), dtype=torch.uint8, device="cuda")
        P = ops.pack(nbits, U)
        res = []
        for fn, nbytes in ((lambda: ops.pack(nbits, U), U.numel() + P.numel() * P.element_size()),
                           (lambda: ops.unpack(nbits, P, torch.float16), 2 * U.numel() + P.numel() * P.element_size())):
            for _ in range(3):
                fn()
            a.record()
            for _ in range(20):
                fn()
            b.record(); torch.cuda.synchronize()
            t = a.elapsed_time(b) / 20 * 1e-3
            res += [f"{t * 1e6:.1f}", f"{nbytes / t / 1e9:.0f}"]
        row(nbits, *res)
    print("\n## configs[2] — prefill, M = 8192 tokens (a 4x2048 chunk of the 32x2048 batch), int4\n")
    row("shape", "fused MFMA dequant-GEMM ms", "TFLOP/s", "dequant + hipBLASLt ms", "TFLOP/s")
    row("---", "---", "---", "---", "---")
    M = 8192
    for (N, K) in ((4096, 4096), (11008, 4096), (4096, 11008)):
        Wq, s, z = rand_layer(N, K, 4)
        x = torch.randn(M, K, device="cuda", dtype=torch.float16)
        y = torch.empty(M, N, device="cuda", dtype=torch.float16)
        res = []
        for fused in (True, False):
            ops.forward(x, Wq, s, z, None, N, K, 64, 4, out=y, fused=fused)
            a.record()
            for _ in range(5):
                ops.forward(x, Wq, s, z, None, N, K, 64, 4, out=y, fused=fused)
            b.record(); torch.cuda.synchronize()
            t = a.elapsed_time(b) / 5 * 1e-3
            res += [f"{t * 1e3:.3f}", f"{2.0 * M * N * K / t / 1e12:.0f}"]
        row(f"{N}x{K}", *res)
    print("\n(CPU baseline: `bench.py` times the oracle on the host cores in its own run — `cpu_baseline` in its JSON line.)")

if __name__ == "__main__":
    main()
