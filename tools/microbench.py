#!/usr/bin/env python3
"""Kernel micro-benchmarks on one MI355X (development aid; bench.py is the contract benchmark).

    python tools/microbench.py [gemv] [gemm] [quant] [pack]

GEMV timing cycles through a pool of distinct layers larger than the 256 MiB Infinity Cache so the
weights really come from HBM (SURVEY.md §7 "hard parts").
"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from hqq_amd import ops  # noqa: E402

HBM_PEAK = 8.0e12


def ev_time(fn, iters, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e-3 / iters


def rand_layer(N, K, nbits, gs=64, dt=torch.float16):
    per = 8 // nbits if nbits != 3 else 10
    R = N * K // gs
    prow = (R + 9) // 10 if nbits == 3 else R // per
    if nbits == 3:
        Wq = torch.randint(0, 2 ** 30, (prow, gs), dtype=torch.int32, device="cuda")
    else:
        Wq = torch.randint(0, 256, (prow, gs), dtype=torch.uint8, device="cuda")
    s = (torch.rand(R, 1, device="cuda") * 0.004 + 0.001).to(dt)
    z = (torch.rand(R, 1, device="cuda") * (2 ** nbits - 1)).to(dt)
    return Wq, s, z


def gemv_bytes(N, K, nbits, M=1, gs=64):
    wq = 4 * gs * ((N * K // gs + 9) // 10) if nbits == 3 else N * K * nbits // 8
    return wq + 4 * (N * K // gs) + 2 * K * M + 2 * N * M


def graph_time(fn_sweep, n_calls, reps=5):
    """device time per call of a sweep captured in one hipGraph (no host launch overhead in the number)"""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn_sweep()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn_sweep()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e-3 / (reps * n_calls)


def bench_gemv(nbits_list=(4, 2), shapes=None, Ms=(1, 4)):
    print("== fused dequant-GEMV, fp16, gs=64 (pool > 256 MiB, back-to-back in one hipGraph) ==")
    shapes = shapes or [(4096, 4096), (11008, 4096), (4096, 11008), (12288, 4096), (22016, 4096), (8192, 8192), (28672, 8192)]
    for nbits in nbits_list:
        for (N, K) in shapes:
            for M in Ms:
                nbytes = gemv_bytes(N, K, nbits, M)
                pool_n = max(4, int(600e6 / nbytes) + 1)
                pool = [rand_layer(N, K, nbits) for _ in range(pool_n)]
                x = torch.randn(M, K, device="cuda", dtype=torch.float16)
                y = torch.empty(M, N, device="cuda", dtype=torch.float16)

                def sweep():
                    for (Wq, s, z) in pool:
                        ops.gemv(x, Wq, s, z, None, N, K, 64, nbits, out=y)

                t = graph_time(sweep, pool_n)
                print(f"int{nbits} {N:6d}x{K:<6d} M={M}  {t * 1e6:8.2f} us  {nbytes / t / 1e9:8.1f} GB/s  {nbytes / t / HBM_PEAK * 100:5.1f}% of 8 TB/s  (pool {pool_n})")
                del pool
    print("== grouped launches (q|k|v and gate|up read the same x) ==")
    for nbits in nbits_list:
        for (n, N, K) in [(3, 4096, 4096), (2, 11008, 4096), (4, 4096, 4096)]:
            nbytes = n * gemv_bytes(N, K, nbits, 1)
            pool_n = max(3, int(600e6 / nbytes) + 1)
            pool = [[rand_layer(N, K, nbits) for _ in range(n)] for _ in range(pool_n)]
            x = torch.randn(1, K, device="cuda", dtype=torch.float16)
            ys = [torch.empty(1, N, device="cuda", dtype=torch.float16) for _ in range(n)]

            def sweep():
                for grp in pool:
                    ops.gemv_grouped(x, [(Wq, s, z, None, N) for (Wq, s, z) in grp], K, 64, nbits, outs=ys)

            t = graph_time(sweep, pool_n)
            print(f"int{nbits} {n} x {N}x{K} M=1  {t * 1e6:8.2f} us  {nbytes / t / 1e9:8.1f} GB/s  {nbytes / t / HBM_PEAK * 100:5.1f}% of 8 TB/s  (pool {pool_n})")
            del pool


def bench_gemm():
    print("== fused dequant-GEMM (MFMA f16), gs=64 ==")
    for nbits in (4, 2):
        for (M, N, K) in [(8192, 4096, 4096), (8192, 11008, 4096), (8192, 4096, 11008), (16, 4096, 4096), (64, 4096, 4096), (256, 4096, 4096)]:
            Wq, s, z = rand_layer(N, K, nbits)
            x = torch.randn(M, K, device="cuda", dtype=torch.float16)
            y = torch.empty(M, N, device="cuda", dtype=torch.float16)
            t = ev_time(lambda: ops.gemm(x, Wq, s, z, None, N, K, 64, nbits, out=y), 10)
            fl = 2.0 * M * N * K
            print(f"int{nbits} M={M:6d} {N:6d}x{K:<6d} {t * 1e3:8.3f} ms  {fl / t / 1e12:8.1f} TFLOP/s  {fl / t / 2.5e15 * 100:5.1f}% of 2.5 PF")
    # library comparison: dequantize kernel + hipBLASLt GEMM (what HQQBackend.PYTORCH does on the GPU, minus 3 temporaries)
    M, N, K = 8192, 4096, 4096
    Wq, s, z = rand_layer(N, K, 4)
    x = torch.randn(M, K, device="cuda", dtype=torch.float16)
    t = ev_time(lambda: torch.matmul(x, ops.dequantize(Wq, s, z, N, K, 64, 4).t()), 10)
    print(f"[ref] dequant kernel + torch.matmul  M={M} {N}x{K}: {t * 1e3:.3f} ms  {2.0 * M * N * K / t / 1e12:.1f} TFLOP/s")


def bench_batch():
    """decode with a batch (5 <= M <= 128): fused paths vs dequantise + library GEMM, weights from HBM (pool > 256 MiB)"""
    print("== small / medium batches, int4 gs=64 fp16: device us per call (graph replay over a pool of distinct layers) ==")
    for (N, K) in [(4096, 4096), (11008, 4096), (4096, 11008)]:
        nb = gemv_bytes(N, K, 4, 1)
        pool_n = max(4, int(600e6 / nb) + 1)
        pool = [rand_layer(N, K, 4) for _ in range(pool_n)]
        for M in (8, 16, 32, 64, 128):
            x = torch.randn(M, K, device="cuda", dtype=torch.float16)
            y = torch.empty(M, N, device="cuda", dtype=torch.float16)
            res = []
            for name, fn in (("forward(default)", lambda W: ops.forward(x, W[0], W[1], W[2], None, N, K, 64, 4, out=y)),
                             ("dequant+matmul", lambda W: torch.matmul(x, ops.dequantize(W[0], W[1].reshape(-1), W[2].reshape(-1), N, K, 64, 4).t()))):
                def sweep():
                    for W in pool:
                        fn(W)
                try:
                    t = graph_time(sweep, pool_n)
                    res.append(f"{name} {t * 1e6:7.2f} us ({gemv_bytes(N, K, 4, M) / t / 1e9:6.0f} GB/s)")
                except Exception as e:  # noqa: BLE001
                    res.append(f"{name} failed: {str(e)[:60]}")
            print(f"{N}x{K} M={M:4d}: " + "   ".join(res))
        del pool


def bench_quant():
    print("== Quantizer.quantize (HQ solver 20 iters + pack), fp16 weights N(0,0.02^2) ==")
    for (N, K) in [(4096, 4096), (11008, 4096)]:
        W = (torch.randn(N, K, device="cuda") * 0.02).half()
        for nbits in (4, 3, 2):
            t = ev_time(lambda: ops.quantize(W, nbits=nbits, group_size=64, round_zero=(nbits == 4)), 3, warmup=1)
            print(f"int{nbits} {N}x{K}: {t * 1e3:8.2f} ms  {N * K * 20 / t / 1e9:8.1f} G elem-iter/s")


def bench_pack():
    print("== pack / unpack / dequantize bandwidth (4096x4096x64 = 1 GiB of fp32-free traffic) ==")
    R, C = 4096 * 4096 // 64 * 16, 64
    for nbits in (4, 3, 2):
        U = torch.randint(0, 2 ** nbits, (R, C), dtype=torch.uint8, device="cuda")
        t = ev_time(lambda: ops.pack(nbits, U), 10)
        P = ops.pack(nbits, U)
        print(f"pack   int{nbits}: {t * 1e6:8.1f} us  {(U.numel() + P.numel() * P.element_size()) / t / 1e9:8.1f} GB/s")
        t = ev_time(lambda: ops.unpack(nbits, P, torch.float16), 10)
        print(f"unpack int{nbits}: {t * 1e6:8.1f} us  {(2 * U.numel() + P.numel() * P.element_size()) / t / 1e9:8.1f} GB/s")
    N = K = 8192
    for nbits in (4, 3, 2):
        Wq, s, z = rand_layer(N, K, nbits)
        t = ev_time(lambda: ops.dequantize(Wq, s, z, N, K, 64, nbits), 10)
        print(f"dequant int{nbits} {N}x{K}: {t * 1e6:8.1f} us  {(2 * N * K + Wq.numel() * Wq.element_size()) / t / 1e9:8.1f} GB/s")


if __name__ == "__main__":
    which = sys.argv[1:] or ["gemv", "gemm", "quant", "pack"]
    print(torch.cuda.get_device_name(0), "| torch", torch.__version__)
    t0 = time.time()
    for w in which:
        {"gemv": bench_gemv, "gemm": bench_gemm, "quant": bench_quant, "pack": bench_pack, "batch": bench_batch}[w]()
    print(f"[done in {time.time() - t0:.1f}s]")
