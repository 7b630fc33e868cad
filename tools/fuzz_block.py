#!/usr/bin/env python3
"""Randomised sweep of the round-4 / round-5 kernels around the dequant path: the decoder block's folded launches (hqq_hip_gemv_block) against the launches they
replace, and of the round-4 kernels outside the dequant path: the dense GEMM (hqq_hip_gemm_dense) against an fp32 matmul of the same 16-bit
inputs, the decode-attention kernels against float64 softmax attention (and the rotary form against rope_cache + attn_decode, bit for bit), and
add_rmsnorm against the formula in float64 (development aid; needs an MI355X).    python tools/fuzz_block.py [cases] [seed]"""
import random
import sys

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from hqq_amd import ops  # noqa: E402


def near(a, b, ulps=2.0):
    """|a - b| within `ulps` units in the last place of the compute dtype AT THE SCALE OF THE OUTPUTS (their rms), not at the scale of an output that happens to fall
    next to zero: one element of the normalised row rounding the other way moves every output by ~1e-5, which is hundreds of ulps of an output of 7e-5"""
    a32, b32 = a.float(), b.float()
    scale = torch.maximum(torch.maximum(a32.abs(), b32.abs()), b32.pow(2).mean().sqrt().expand_as(b32) * 0.25)
    ulp = torch.pow(2.0, torch.floor(torch.log2(scale.clamp_min(1e-20))) - (10 if a.dtype == torch.float16 else 7))
    return (a32 - b32).abs() <= ulps * ulp


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    for it in range(cases):
        g = torch.Generator(device="cuda").manual_seed(it)
        dt = rnd.choice([torch.float16, torch.bfloat16])
        ulp = 2.0 ** -10 if dt == torch.float16 else 2.0 ** -7
        kind = rnd.choice(["dense", "dense", "attn", "attn", "norm", "fold", "fold", "fold"])
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
            elif kind == "fold":
                # the decoder block's launches with the glue folded in (hqq_hip_gemv_block, round 5) against the launches they replace
                nbits = rnd.choice([4, 4, 3, 2])
                w3 = nbits == 3
                K = 64 * rnd.randint(1, 128)
                mode = rnd.choice(["resid", "resid", "norm", "silu", "rope"])
                if mode in ("norm", "silu", "rope"):
                    K = min(K, 8192)

                def mk(N, seed):
                    gl = torch.Generator(device="cuda").manual_seed(seed)
                    R = N * K // 64
                    U = torch.randint(0, 2 ** nbits, (R, 64), device="cuda", generator=gl, dtype=torch.uint8)
                    s_ = (torch.rand(R, 1, device="cuda", generator=gl) * 0.004 + 0.001).to(dt)
                    z_ = (torch.rand(R, 1, device="cuda", generator=gl) * (2 ** nbits - 1)).to(dt)
                    if rnd.random() < 0.5:
                        z_ = z_.round()
                    Wq_ = ops.pack(nbits, U)
                    if w3:
                        Wq_ = ops.w3s_pack(Wq_, N, K)
                    return (Wq_, s_, z_, N)

                def sub_of(L_):
                    if dt != torch.float16:
                        return 0
                    return ops.OPT_META_SCALABLE if (ops.w3s_meta_scalable(L_[1], L_[2], L_[3], K) if w3 else ops.meta_scalable(L_[1], L_[2], L_[3], K, 64, nbits)) else 0
                base = ops.OPT_W3S if w3 else 0
                x = torch.randn(1, K, device="cuda", generator=g).to(dt)
                gam = (1 + 0.1 * torch.randn(K, device="cuda", generator=g)).to(dt)
                if mode == "resid":
                    N = 8 * rnd.randint(1, 700)
                    L_ = mk(N, it)
                    o_ = base | sub_of(L_)
                    h = (torch.randn(1, N, device="cuda", generator=g) * 2).to(dt)
                    want = h + ops.gemv(x, L_[0], L_[1], L_[2], None, N, K, 64, nbits, opts=o_)
                    ops.gemv_block(x, None, 0.0, [L_], K, 64, nbits, [h], ops.BLOCK_RESID, opts=o_)
                    ok = torch.equal(h, want)
                    what = f"fold resid int{nbits} {N}x{K} {dt}"
                elif mode == "norm":
                    Ls = [mk(8 * rnd.randint(1, 300), it * 7 + j) for j in range(rnd.randint(1, 4))]
                    o_ = base | min(sub_of(L_) for L_ in Ls)
                    outs = [torch.empty(1, L_[3], dtype=dt, device="cuda") for L_ in Ls]
                    ops.gemv_block(x, gam, 1e-5, Ls, K, 64, nbits, outs, ops.BLOCK_NORM, opts=o_)
                    xk = ops.add_rmsnorm(x.clone(), None, gam, 1e-5)
                    ys = ops.gemv_grouped(xk, [L_[:3] + (None, L_[3]) for L_ in Ls], K, 64, nbits, opts=o_)
                    ok = all(bool(near(a_, b_).all()) and int((a_ != b_).sum()) <= max(8, a_.numel() // 5) for a_, b_ in zip(outs, ys))   # (a few elements of the normalised row round the other way; each moves ~2 % of the outputs by an ulp)
                    what = f"fold norm int{nbits} {[L_[3] for L_ in Ls]}x{K} {dt}: {[int((a_ != b_).sum()) for a_, b_ in zip(outs, ys)]} outputs differ"
                elif mode == "silu":
                    I = 8 * rnd.randint(1, 400)
                    ga, up = mk(I, it * 3), mk(I, it * 3 + 1)
                    pair = ops.pair_layers(ga, up, K, 64, nbits, w3s=w3)
                    o_ = base | sub_of(pair)
                    a_ = torch.empty(1, I, dtype=dt, device="cuda")
                    ops.gemv_block(x, gam, 1e-6, [pair], K, 64, nbits, [a_], ops.BLOCK_NORM | ops.BLOCK_SILU, opts=o_)
                    xk = ops.add_rmsnorm(x.clone(), None, gam, 1e-6)
                    yg, yu = ops.gemv_grouped(xk, [ga[:3] + (None, I), up[:3] + (None, I)], K, 64, nbits, opts=base)
                    ref_ = ops.silu_mul(yg, yu)
                    ok = bool(near(a_, ref_, 4.0).all()) and int((a_ != ref_).sum()) <= max(8, I // 4)
                    what = f"fold silu int{nbits} {I}x{K} {dt}: {int((a_ != ref_).sum())} differ"
                else:
                    hd = rnd.choice([64, 128])
                    nkv = rnd.choice([1, 2, 4])
                    nh = nkv * rnd.choice([1, 2, 4])
                    Lc = rnd.choice([32, 100])
                    q_, k_, v_ = mk(nh * hd, it * 5), mk(nkv * hd, it * 5 + 1), mk(nkv * hd, it * 5 + 2)
                    qp, kp = ops.rotary_pair_layout(q_, K, 64, nbits, hd, w3s=w3), ops.rotary_pair_layout(k_, K, 64, nbits, hd, w3s=w3)
                    cos = torch.randn(hd, device="cuda", generator=g).to(dt); sin = torch.randn(hd, device="cuda", generator=g).to(dt)
                    pos = torch.tensor([rnd.randint(-1, Lc)], device="cuda")
                    yq = torch.empty(1, nh * hd, dtype=dt, device="cuda"); yk = torch.empty(1, nkv * hd, dtype=dt, device="cuda"); yv = torch.empty_like(yk)
                    ops.gemv_block(x, gam, 1e-5, [q_, k_, v_], K, 64, nbits, [yq, yk, yv], ops.BLOCK_NORM, opts=base)
                    kc1 = torch.full((nkv, Lc, hd), 3.0, dtype=dt, device="cuda"); vc1 = torch.full_like(kc1, 2.0); kc2, vc2 = kc1.clone(), vc1.clone()
                    qr1 = torch.empty(1, nh, 1, hd, dtype=dt, device="cuda"); qr2 = torch.empty_like(qr1)
                    ops.rope_cache(yq, yk, yv, cos, sin, pos, kc1, vc1, qr1)
                    ops.gemv_block(x, gam, 1e-5, [qp, kp, v_], K, 64, nbits, [qr2, kc2, vc2], ops.BLOCK_NORM | ops.BLOCK_ROPE, opts=base, rope=(cos, sin, pos, hd, Lc))
                    ok = torch.equal(qr1, qr2) and torch.equal(kc1, kc2) and torch.equal(vc1, vc2)
                    what = f"fold rope int{nbits} heads {nh}/{nkv} hd {hd} K {K} pos {int(pos)} of {Lc} {dt}"
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
