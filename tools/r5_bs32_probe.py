#!/usr/bin/env python3
"""bs = 32 launches of a 7B decoder block, one row per kernel variant (us per launch; graph replay over > 256 MiB of distinct layers per shape).

    python tools/r5_bs32_probe.py [nbits] [M list]        HQQ_AMD_LIB=tools/libhqq_hip_<lab>.so selects a lab build (tools/r5_build_labs.sh)

Columns (lab library tools/libhqq_hip_kwave.so / _kwnoarith.so only; the shipped library has the one column sk): kw = tools/lab_kwave/kwave.hip with its own choice of tiles per workgroup, kw1..kw6 = forced tiles per workgroup, sk = the split-K kernel (skinny.hip).
Lab libraries: kwnoarith / sknoarith = the same launches with loads only (the floor of each launch shape), sknofin = split-K without its finish.
"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hqq_amd import ops  # noqa: E402

nbits = int(sys.argv[1]) if len(sys.argv) > 1 else 4
Ms = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [8, 32, 64]
gs = 64
g = torch.Generator().manual_seed(0)


def layer(N, K):
    R = N * K // gs
    U = torch.randint(0, 2 ** nbits, (R, gs), generator=g, dtype=torch.uint8).cuda()
    P = ops.pack(nbits, U)
    s = (torch.rand(R, 1, generator=g) * 0.004 + 0.001).half().cuda()
    z = (torch.rand(R, 1, generator=g) * (2 ** nbits - 1)).round().half().cuda()
    if nbits == 3:
        P = ops.w3s_pack(P, N, K)
    return (P, s, z, None, N)


SH = {"o": ([4096], 4096, 36), "qkv": ([4096] * 3, 4096, 12), "gateup": ([11008] * 2, 4096, 7), "down": ([4096], 11008, 13)}
BYTES = {"o": 9453568, "qkv": 3 * 9453568, "gateup": 2 * 25392640, "down": 25392640}   # int4 algorithmic bytes at bs = 1
LAB_KW = "hqq_hip_kw" in os.environ.get("HQQ_AMD_LIB", "")   # tools/lab_kwave/build.sh: the no-split kernel is the lab library's default for 5..64 rows
VARS = [("kw", 0), ("kw1", ops.OPT_SKINNY_KS(1)), ("kw2", ops.OPT_SKINNY_KS(2)), ("kw3", ops.OPT_SKINNY_KS(3)), ("kw4", ops.OPT_SKINNY_KS(4)), ("kw6", ops.OPT_SKINNY_KS(6)),
        ("sk", ops.OPT_BATCH_SPLITK)] if LAB_KW else [("sk", 0)]
print(f"lib={os.environ.get('HQQ_AMD_LIB', 'default')} nbits={nbits}", flush=True)
tot = {}
for name, (Ns, K, nl) in SH.items():
    groups = [[layer(N, K) for N in Ns] for _ in range(nl)]
    if nbits == 3:
        sub = all(ops.w3s_meta_scalable(L[1], L[2], L[4], K) for G in groups for L in G)
    else:
        sub = all(ops.meta_scalable(L[1], L[2], L[4], K, gs, nbits) for G in groups for L in G)
    base = (ops.OPT_META_SCALABLE if sub else 0) | (ops.OPT_W3S if nbits == 3 else 0)
    for M in Ms:
        x = (torch.randn(M, K, generator=g) * 0.5).half().cuda()
        row = []
        for vn, vo in VARS:
            try:
                f = lambda: [ops.gemv_grouped(x, G, K, gs, nbits, opts=base | vo) for G in groups]
                f(); torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    f()
                for _ in range(2): gr.replay()
                torch.cuda.synchronize()
                best = 1e9
                for _ in range(4):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); gr.replay(); gr.replay(); e1.record(); torch.cuda.synchronize()
                    best = min(best, e0.elapsed_time(e1) * 1e3 / (2 * nl))
                row.append(f"{vn}:{best:6.2f}")
                tot.setdefault((M, vn), {})[name] = best
            except Exception as e:   # noqa: BLE001
                row.append(f"{vn}:ERR({str(e)[:30]})")
        print(f"  {name:7s} M={M:2d} sub={int(sub)}  " + "  ".join(row), flush=True)
    del groups
    torch.cuda.empty_cache()
if nbits == 4:
    stack_bytes = 32 * sum(BYTES.values())
    for M in Ms:
        for vn, _ in VARS:
            d = tot.get((M, vn), {})
            if len(d) == 4:
                t = 32 * sum(d.values())
                print(f"  stack M={M:2d} {vn:4s}: {t / 1e3:.3f} ms per step  {stack_bytes / (t * 1e-6) / 8e12:.3f} of 8 TB/s (bs=1 algorithmic bytes)")
        if LAB_KW:
            best = {n: min(tot[(M, vn)][n] for vn, _ in VARS[:6] if n in tot.get((M, vn), {})) for n in SH}
            t = 32 * sum(best.values())
            print(f"  stack M={M:2d} best kw per launch: {t / 1e3:.3f} ms  {stack_bytes / (t * 1e-6) / 8e12:.3f}")
