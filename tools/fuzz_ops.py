#!/usr/bin/env python3
"""Randomised shape sweep of pack / unpack / dequantise / grouped GEMV / bf16 forward / quantise invariants (development aid; MI355X).
    python tools/fuzz_ops.py [cases] [seed]"""
import random
import sys

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from hqq_amd import ops  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0

    def fail(msg):
        nonlocal bad
        bad += 1
        print("FAIL", msg)

    for it in range(cases):
        g = torch.Generator().manual_seed(1000 + it)
        nbits = rnd.choice([8, 4, 4, 3, 2, 1])
        per = 10 if nbits == 3 else 8 // nbits
        gs = rnd.choice([64, 64, 64, 16, 32, 128])
        K = gs * rnd.randint(1, 24)
        N = (1 if nbits == 3 else per) * rnd.randint(1, 300)
        R = N * K // gs
        U = torch.randint(0, 2 ** nbits, (R, gs), generator=g, dtype=torch.uint8).cuda()
        # ---- pack / unpack round trip ----
        P = ops.pack(nbits, U)
        U2 = ops.unpack(nbits, P)
        if not torch.equal(U2[:R].to(torch.uint8), U):
            fail(f"[{it}] pack/unpack int{nbits} R={R} gs={gs}")
        # ---- dequantise == the reference arithmetic done by torch on the GPU (two roundings in the compute dtype) ----
        for dt in (torch.float16, torch.bfloat16, torch.float32):
            s = (torch.rand(R, 1, generator=g) * 0.004 + 0.001).to(dt).cuda()
            z = (torch.rand(R, 1, generator=g) * (2 ** nbits - 1)).to(dt).cuda()
            Wd = ops.dequantize(P, s.reshape(-1), z.reshape(-1), N, K, gs, nbits)
            want = ((U.to(dt) - z) * s).reshape(N, K)
            if not torch.equal(Wd, want):
                fail(f"[{it}] dequantize int{nbits} N={N} K={K} gs={gs} {dt}: {int((Wd != want).sum())} elements differ")
        # ---- grouped GEMV == single launches (fp16; decode rows) ----
        if nbits != 3 or gs == 64:
            s = (torch.rand(R, 1, generator=g) * 0.004 + 0.001).half().cuda()
            z = (torch.rand(R, 1, generator=g) * (2 ** nbits - 1)).half().cuda()
            M = rnd.choice([1, 2, 4]) if nbits == 3 else rnd.choice([1, 2, 3, 4, 6, 16, 24, 64])
            x = torch.randn(M, K, generator=g).half().cuda()
            layers = [(P, s, z, None, N)]
            for j in range(rnd.randint(0, 3)):
                N2 = (1 if nbits == 3 else per) * rnd.randint(1, 200)
                if nbits == 3 and (N2 * (K // gs) + 9) // 10 < K // gs:
                    continue
                R2 = N2 * K // gs
                Uj = torch.randint(0, 2 ** nbits, (R2, gs), generator=g, dtype=torch.uint8).cuda()
                sj = (torch.rand(R2, 1, generator=g) * 0.004 + 0.001).half().cuda()
                zj = (torch.rand(R2, 1, generator=g) * (2 ** nbits - 1)).half().cuda()
                bj = torch.randn(N2, generator=g).half().cuda() if rnd.random() < 0.5 else None
                layers.append((ops.pack(nbits, Uj), sj, zj, bj, N2))
            if nbits == 3 and (N * (K // gs) + 9) // 10 < K // gs:
                continue
            if M > 16 and not all(ops.skinny_covers(torch.float16, M, L[4], K, gs, nbits) for L in layers):
                M = 16
                x = x[:16].contiguous()
            if 4 < M <= 16 and K % 64:
                M = 4
                x = x[:4].contiguous()
            try:
                ys = ops.gemv_grouped(x, layers, K, gs, nbits)
                for L, y in zip(layers, ys):
                    y1 = ops.gemv(x, L[0], L[1], L[2], L[3], L[4], K, gs, nbits)
                    if not torch.equal(y, y1):
                        d = (y != y1).nonzero().tolist()
                        fail(f"[{it}] grouped != single int{nbits} N={L[4]} K={K} gs={gs} M={M} layers {[q[4] for q in layers]} bias {L[3] is not None}: {len(d)} differ at {d[:3]} "
                             f"{[(y[i, j].item(), y1[i, j].item()) for i, j in d[:3]]}")
            except NotImplementedError:
                pass
        # ---- bf16 decode forward vs bf16 dequantise + fp32 matmul ----
        if nbits in (4, 2):
            s = (torch.rand(R, 1, generator=g) * 0.004 + 0.001).bfloat16().cuda()
            z = (torch.rand(R, 1, generator=g) * (2 ** nbits - 1)).bfloat16().cuda()
            M = rnd.choice([1, 2, 3, 4, 9])
            x = torch.randn(M, K, generator=g).bfloat16().cuda()
            y = ops.forward(x, P, s, z, None, N, K, gs, nbits)
            Wd = ops.dequantize(P, s.reshape(-1), z.reshape(-1), N, K, gs, nbits)
            ref = (x.double() @ Wd.double().t())
            err = (y.double() - ref).abs()
            tol = 2.0 ** -7 * ref.abs().clamp(min=2.0 ** -4) * 1.01 + 4e-7 * (x.float().abs() @ Wd.float().abs().t()).double() + 1e-3
            if not bool((err <= tol).all()):
                fail(f"[{it}] bf16 forward int{nbits} N={N} K={K} gs={gs} M={M}: max err {err.max().item():.3e}")
    # ---- quantise invariants on random shapes ----
    for it in range(max(4, cases // 10)):
        g = torch.Generator().manual_seed(5000 + it)
        nbits = rnd.choice([8, 4, 3, 2, 1])
        gs = rnd.choice([64, 64, 32, 128])
        N, K = 8 * rnd.randint(1, 64), gs * rnd.randint(1, 16)
        if nbits == 3 and (N * K // gs) < 10:
            continue
        W = (torch.randn(N, K, generator=g) * 0.02).half().cuda()
        Wq, s, z = ops.quantize(W, nbits=nbits, group_size=gs, round_zero=(nbits == 4))
        R = N * K // gs
        U = ops.unpack(nbits, Wq)[:R]
        if int(U.max()) > 2 ** nbits - 1:
            fail(f"[q{it}] level out of range int{nbits}")
        Wd = ((U.float() - z) * s).reshape(N, K)
        rel = ((Wd - W.float()).abs().mean() / W.float().abs().mean()).item()
        lim = {8: 0.01, 4: 0.12, 3: 0.25, 2: 0.55, 1: 1.25}[nbits]
        if not rel < lim or not torch.isfinite(s).all() or not torch.isfinite(z).all():
            fail(f"[q{it}] quantise int{nbits} {N}x{K} gs={gs}: mean rel error {rel:.3f} (limit {lim})")
    print(f"{cases} cases, {bad} failures")


if __name__ == "__main__":
    main()
