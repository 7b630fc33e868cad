#!/usr/bin/env python3
"""Randomised shape sweep of the fused forward against HIP dequantise + fp32 matmul (development aid; needs an MI355X).
    python tools/fuzz_forward.py [cases] [seed] [factored]"""
import random
import sys

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from hqq_amd import ops  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    factored = len(sys.argv) > 3 and sys.argv[3] == "factored"
    if factored:
        ops.set_gemv_mode(ops.GEMV_FACTORED)
    bad = 0
    for it in range(cases):
        nbits = rnd.choice([8, 4, 4, 4, 2, 2, 1, 3])
        per = 10 if nbits == 3 else 8 // nbits
        gs = 64 if nbits == 3 or rnd.random() < 0.8 else rnd.choice([16, 32, 128])
        K = gs * rnd.randint(1, 40) if rnd.random() < 0.4 else 256 * rnd.randint(1, 24) if rnd.random() < 0.7 else 1024 * rnd.randint(8, 28)
        K = (K // gs) * gs or gs
        N = (per if nbits != 3 else 1) * (rnd.randint(1, 700) if K < 8192 else rnd.randint(1, 90))
        M = rnd.choice([1, 2, 3, 4, 5, 7, 8, 13, 16, 17, 31, 32, 33, 48, 64, 65, 100, 128, 129, 200, 257, 300, 513, 700, 2561, 3001])   # (beyond 2560: dequantise + in-tree dense GEMM)
        g = torch.Generator().manual_seed(it)
        R = N * K // gs
        U = torch.randint(0, 2 ** nbits, (R, gs), generator=g, dtype=torch.uint8)
        s = (torch.rand(R, 1, generator=g) * 0.004 + 0.001).half().cuda()
        z = (torch.rand(R, 1, generator=g) * (2 ** nbits - 1)).half().cuda()
        P = ops.pack(nbits, U.cuda())
        x = torch.randn(M, K, generator=g).half().cuda()
        b = torch.randn(N, generator=g).half().cuda() if rnd.random() < 0.4 else None
        try:
            y = ops.forward(x, P, s, z, b, N, K, gs, nbits)
        except NotImplementedError as e:   # forward() composes whatever the fused kernels do not cover: raising is a failure
            bad += 1
            print(f"[{it}] FAIL int{nbits} N={N} K={K} gs={gs} M={M}: raised ({str(e)[:90]})")
            continue
        Wd = ops.dequantize(P, s.reshape(-1), z.reshape(-1), N, K, gs, nbits)
        ref = x.double() @ Wd.double().t()
        refh = ref.half()
        pre = refh.float().abs()
        if b is not None:
            refh = refh + b
        err = (y.float() - refh.float()).abs()
        # one fp16 ulp of the rounded matmul result (and of the sum, with a bias: two roundings) + the fp32 accumulation noise
        mag = (x.float().abs() @ Wd.float().abs().t())
        tol = 2.0 ** -10 * (pre + refh.float().abs()).clamp(min=2.0 ** -4) * 1.01 + 4e-7 * mag + 1e-4
        if factored:   # weights not rounded to fp16 one by one: the documented tolerance of the FACTORED mode
            # (each weight deviates from its fp16-rounded value by up to half an ulp: noise ~ 2^-11 * sqrt(sum x^2 w^2), 6 sigma allowed)
            tol = tol + 1e-3 + 1e-3 * refh.float().abs() + 6 * 2.0 ** -11 * 0.6 * ((x.float() ** 2) @ (Wd.float() ** 2).t()).sqrt()
        ok = bool((err <= tol).all()) and torch.isfinite(y).all()
        if not ok:
            bad += 1
            i = int(err.argmax())
            print(f"[{it}] FAIL int{nbits} N={N} K={K} gs={gs} M={M} bias={b is not None}: max err {err.max().item():.3e} at {divmod(i, N)} ref {refh.flatten()[i].item():.4f}")
    print(f"{cases} cases, {bad} failures")


if __name__ == "__main__":
    main()
