#!/usr/bin/env python3
"""Randomised sweep of the round-4 kernels outside the dequant path: the dense GEMM (hqq_hip_gemm_dense) against an fp32 matmul of the same 16-bit
inputs, the decode-attention kernels against float64 softmax attention (and the rotary form against rope_cache + attn_decode, bit for bit), and
add_rmsnorm against the formula in float64 (development aid; needs an MI355X).    python tools/fuzz_block.py [cases] [seed]"""
import random
import sys

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from hqq_amd import ops  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    for it in range(cases):
        g = torch.Generator(device="cuda").manual_seed(it)
        dt = rnd.choice([torch.float16, torch.bfloat16])
        ulp = 2.0 ** -10 if dt == torch.float16 else 2.0 ** -7
        kind = rnd.choice(["dense", "dense", "attn", "attn", "norm"])
        try:
            if kind == "dense":
                M = rnd.choice([1, 3, 17, 64, 255, 256, 257, 300, 511, 700, 1025])
                N = 4 * rnd.randint(1, 330)
                K = 64 * rnd.randint(1, 40)
                x = torch.randn(M, K, device="cuda", generator=g).to(dt)
                W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(dt)
                b = torch.randn(N, device="cuda", generator=g).to(dt) if rnd.random() < 0.5 else None
                y = ops.gemm_dense(x, W, b)
                ref = x.float() @ W.float().t()
                want = ref.to(dt).float() + (0 if b is None else b.float())
                tol = 1e-3 + 1e-3 * want.abs() + 2 * ulp * (want.abs() + ref.abs())
                ok = bool(((y.float() - want).abs() <= tol).all()) and bool(torch.isfinite(y).all())
                what = f"dense {M}x{N}x{K} {dt} bias={b is not None}"
            elif kind == "attn":
                hd = rnd.choice([64, 128, 128, 256])
                n_kv = rnd.choice([1, 2, 4, 8])
                rep = rnd.choice([1, 1, 2, 4])
                n_heads = n_kv * rep
                L = rnd.choice([64, 100, 256, 1000, 2048, 5000])
                pos = rnd.randint(0, L - 1)
                S = rnd.choice([1, 1, 2, 3, 8])
                q = torch.randn(1, n_heads * hd, device="cuda", generator=g).to(dt)
                k = torch.randn(1, n_kv * hd, device="cuda", generator=g).to(dt)
                v = torch.randn(1, n_kv * hd, device="cuda", generator=g).to(dt)
                ang = torch.rand(hd // 2, device="cuda", generator=g) * 6.28
                cos, sin = torch.cat([ang.cos(), ang.cos()]).to(dt), torch.cat([ang.sin(), ang.sin()]).to(dt)
                kc = torch.randn(n_kv, L, hd, device="cuda", generator=g).to(dt)
                vc = torch.randn(n_kv, L, hd, device="cuda", generator=g).to(dt)
                kc[:, pos:] = float("nan"); vc[:, pos:] = float("nan")
                p = torch.tensor([pos], device="cuda")
                kc1, vc1, kc2, vc2 = kc.clone(), vc.clone(), kc.clone(), vc.clone()
                qr = torch.empty(1, n_heads, 1, hd, dtype=dt, device="cuda")
                ops.rope_cache(q, k, v, cos, sin, p, kc1, vc1, qr)
                a = torch.empty(n_heads * hd, dtype=dt, device="cuda")
                b2 = torch.empty_like(a)
                ops.attn_decode(qr, kc1, vc1, p, a, hd ** -0.5, splits=S)
                ops.rope_attn_decode(q, k, v, cos, sin, p, kc2, vc2, b2, hd ** -0.5, splits=S)
                kk = kc1[:, :pos + 1].repeat_interleave(rep, 0).double()
                vv = vc1[:, :pos + 1].repeat_interleave(rep, 0).double()
                want = torch.einsum("hj,hjd->hd", torch.softmax(torch.einsum("hd,hjd->hj", qr.view(n_heads, hd).double(), kk) * hd ** -0.5, -1), vv)
                tol = 1e-3 + 1e-3 * want.abs() + want.abs() * ulp
                ok = torch.equal(a, b2) and torch.equal(kc1[:, :pos + 1], kc2[:, :pos + 1]) and bool(((a.view(n_heads, hd).double() - want).abs() <= tol).all())
                what = f"attn heads {n_heads}/{n_kv} hd {hd} L {L} pos {pos} splits {S} {dt}"
            else:
                H = 8 * rnd.randint(1, 2100)
                rows = rnd.randint(1, 4)
                h = torch.randn(rows, H, device="cuda", generator=g).to(dt)
                d = (torch.randn(rows, H, device="cuda", generator=g) * 0.3).to(dt) if rnd.random() < 0.7 else None
                w = (1 + 0.1 * torch.randn(H, device="cuda", generator=g)).to(dt)
                eps = rnd.choice([1e-5, 1e-6])
                hh = h.clone()
                y = ops.add_rmsnorm(hh, d, w, eps)
                hs = h if d is None else (h.float() + d.float()).to(dt)
                hf = hs.float()   # LlamaRMSNorm's own arithmetic: fp32 mean of squares, fp32 product, one rounding, then the product with the weight in T
                xn = (hf * torch.rsqrt(hf.pow(2).mean(-1, keepdim=True) + eps)).to(dt)
                want = w * xn
                diff = (y.view(torch.int16).int() - want.view(torch.int16).int()).abs()
                # (the fp32 sum of squares is taken in another order: r differs in its last bits, which moves a few roundings by one ulp)
                ok = torch.equal(hh, hs) and int(diff.max()) <= 2 and int((diff > 1).sum()) <= 4 * rows and int((diff > 0).sum()) <= max(16, H // 250) * rows   # (one ulp on the normalised value can become two on its product with the weight)
                what = f"norm rows {rows} H {H} {dt} delta={d is not None}: max ulp {int(diff.max())}, {int((diff > 0).sum())} of {rows * H} differ"
        except Exception as e:  # noqa: BLE001
            ok, what = False, f"{kind}: {type(e).__name__}: {str(e)[:200]}"
        if not ok:
            bad += 1
            print("FAIL", f"[{it}]", what)
    print(f"{cases} cases, {bad} failures")


if __name__ == "__main__":
    main()
