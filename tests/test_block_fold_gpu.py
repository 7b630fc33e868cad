"""GPU parity tests of the decoder block's launches with the glue folded in (hqq_amd/csrc/gemv_block.hip, hqq_hip_gemv_block).

What they replace, in the reference's generate loop (hqq/utils/generation_hf.py:117-540): HF's LlamaDecoderLayer.forward around HQQLinear.forward
(hqq/core/quantize.py:880-898) — input_layernorm -> q|k|v, o -> residual add, post_attention_layernorm -> gate|up -> act_fn(gate) * up, down -> residual add.
Bars:
  * residual epilogue: h' == h + gemv(x) BIT FOR BIT (the same row results, `residual + hidden_states` rounded once) — the launches it replaces;
  * RMSNorm prologue: against the CPU oracle on an fp64 RMSNorm restated with HF's roundings (x.float() * rsqrt(mean + eps) -> T -> weight * T): the
    forward tolerance (1e-3 fp16); against the launches it replaces (add_rmsnorm + gemv_grouped) at most a few outputs an ulp or two (at the outputs' scale) apart (the
    fp32 sum of squares has another fixed order);
  * SiLU * up epilogue on the PAIRED layer: the paired layout holds the two layers' own levels (unpacking it gives them back bit for bit), and the output
    equals silu_mul(gemv(gate), gemv(up)) within the same bound.
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "gpu tests need a ROCm device"
    from hqq_amd import ops as o
    assert o.is_available(), "libhqq_hip.so must load on the GPU box (no fallback)"
    return o


def _layer(ops, N, K, nbits, seed, dt, sub_friendly=True):
    g = torch.Generator().manual_seed(seed)
    R = N * K // 64
    U = torch.randint(0, 2 ** nbits, (R, 64), generator=g, dtype=torch.uint8)
    s = (torch.rand(R, 1, generator=g) * 0.004 + 0.001).to(dt)
    z = (torch.rand(R, 1, generator=g) * (2 ** nbits - 1)).to(dt)
    if sub_friendly:
        z = z.float().clamp_min(0.0625).to(dt)
    else:
        z.view(-1)[::5] = 0.00836
    Wq = ops.pack(nbits, U.cuda())
    sd, zd = s.cuda(), z.cuda()
    Wd = ops.dequantize(Wq, sd.reshape(-1), zd.reshape(-1), N, K, 64, nbits)      # bit-identical to the oracle's (tests/test_hip_parity.py)
    opts = 0
    if nbits == 3:
        Wq = ops.w3s_pack(Wq, N, K)
        opts = ops.OPT_W3S | (ops.OPT_META_SCALABLE if dt == torch.float16 and ops.w3s_meta_scalable(sd, zd, N, K) else 0)
    elif dt == torch.float16 and ops.meta_scalable(sd, zd, N, K, 64, nbits):
        opts = ops.OPT_META_SCALABLE
    return (Wq, sd, zd, N), opts, Wd


def _rmsnorm_hf(h, w, eps):
    """LlamaRMSNorm.forward with its roundings, the mean in float64 (what any fp32 summation order approximates)"""
    x = h.float()
    var = x.double().pow(2).mean(-1, keepdim=True)
    return w * (x * torch.rsqrt(var + eps).float()).to(h.dtype)


def _near(a, b, ulps=2.0):
    """|a - b| within `ulps` units in the last place of the compute dtype at the SCALE OF THE OUTPUTS (a quarter of their rms at least), not at the scale of an
    output that happens to fall next to zero: one element of the normalised row rounding the other way moves every output by ~1e-5"""
    a32, b32 = a.float(), b.float()
    scale = torch.maximum(torch.maximum(a32.abs(), b32.abs()), b32.pow(2).mean().sqrt().expand_as(b32) * 0.25)
    ulp = torch.pow(2.0, torch.floor(torch.log2(scale.clamp_min(1e-20))) - (10 if a.dtype == torch.float16 else 7))
    return (a32 - b32).abs() <= ulps * ulp


CASES = [(4096, 4096), (1024, 8192), (200, 2048 + 768), (64, 64), (344, 1024)]   # 8192: two passes of the workgroup over h; 2816 / 64: ragged passes; 344: N % 8 != 0


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("nbits", [4, 3, 2])
@pytest.mark.parametrize("NK", CASES)
def test_rmsnorm_prologue(ops, dt, nbits, NK):
    N, K = NK
    if N % 8: N += 8 - N % 8
    Ls, Wds, opts = [], [], None
    for i in range(3):                                   # q | k | v: three layers, one launch
        L, o, Wd = _layer(ops, N if i == 0 else max(N // 4 // 8 * 8, 8), K, nbits, seed=N + K + nbits + i, dt=dt, sub_friendly=(i < 3))
        Ls.append(L); Wds.append(Wd)
        opts = o if opts is None else (opts & o) | (o & ops.OPT_W3S)
    g = torch.Generator(device="cuda").manual_seed(K)
    h = (torch.randn(1, K, device="cuda", generator=g) * 1.7).to(dt)
    w = (1 + 0.1 * torch.randn(K, device="cuda", generator=g)).to(dt)
    eps = 1e-5
    outs = [torch.full((1, L[3]), float("nan"), dtype=dt, device="cuda") for L in Ls]
    h0 = h.clone()
    ops.gemv_block(h, w, eps, Ls, K, 64, nbits, outs, ops.BLOCK_NORM, opts=opts)
    assert torch.equal(h, h0)                            # the residual stream is only read
    xn = _rmsnorm_hf(h, w, eps)
    tol = dict(rtol=2.0 ** -7, atol=2e-3) if dt == torch.bfloat16 else dict(rtol=1e-3, atol=1e-3)
    for y, Wd in zip(outs, Wds):
        want = (xn.double() @ Wd.double().t()).float()
        torch.testing.assert_close(y.float(), want, **tol)
    # the launches it replaces: add_rmsnorm + gemv_grouped; the normalised row may differ by an ulp in a few elements (another order of the fp32 sum of squares)
    xk = ops.add_rmsnorm(h.clone(), None, w, eps)
    ys = ops.gemv_grouped(xk, [L[:3] + (None, L[3]) for L in Ls], K, 64, nbits, opts=opts)
    for y, y2 in zip(outs, ys):
        assert bool(_near(y, y2).all()) and int((y != y2).sum()) <= max(8, y.numel() // 5), int((y != y2).sum())   # (each element of the normalised row that rounds the other way moves ~2 % of the outputs by an ulp)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("nbits", [4, 3, 2])
@pytest.mark.parametrize("NK", [(4096, 4096), (4096, 11008), (1024, 28672), (200, 2048 + 768), (64, 64), (344, 1024)])   # o, down (x in two passes), few rows x long K (row shared by a workgroup's waves)
def test_residual_epilogue_is_bit_exact(ops, dt, nbits, NK):
    N, K = NK
    if N % 8: N += 8 - N % 8
    L, opts, _ = _layer(ops, N, K, nbits, seed=7 * N + K + nbits, dt=dt, sub_friendly=(K % 128 == 0))
    g = torch.Generator(device="cuda").manual_seed(N + K)
    x = torch.randn(1, K, device="cuda", generator=g).to(dt)
    h = (torch.randn(1, N, device="cuda", generator=g) * 3).to(dt)
    y = ops.gemv(x, L[0], L[1], L[2], None, N, K, 64, nbits, opts=opts)
    want = h + y                                          # `residual + hidden_states`: one rounding in T
    ops.gemv_block(x, None, 0.0, [L], K, 64, nbits, [h], ops.BLOCK_RESID, opts=opts)
    assert torch.equal(h, want)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("nbits", [4, 3, 2])
@pytest.mark.parametrize("IK", [(11008, 4096), (2048, 8192), (152, 2048 + 768), (64, 64)])
def test_paired_gate_up_with_silu_epilogue(ops, dt, nbits, IK):
    I, K = IK
    gate, og, Wg = _layer(ops, I, K, nbits, seed=I + K + nbits, dt=dt)
    up, ou, Wu = _layer(ops, I, K, nbits, seed=I + K + nbits + 1, dt=dt, sub_friendly=(nbits != 2))
    pair = ops.pair_layers(gate, up, K, 64, nbits, w3s=(nbits == 3))
    assert pair[3] == 2 * I
    # the paired layer holds the two layers' own weights: dequantising it gives gate's matrix on top of up's, bit for bit
    Wq_ref = ops.w3s_unpack(pair[0], 2 * I, K) if nbits == 3 else pair[0]
    Wp = ops.dequantize(Wq_ref, pair[1], pair[2], 2 * I, K, 64, nbits)
    assert torch.equal(Wp[:I], Wg) and torch.equal(Wp[I:], Wu)
    if dt != torch.float16:
        po = ops.OPT_W3S if nbits == 3 else 0
    elif nbits == 3:
        po = ops.OPT_W3S | (ops.OPT_META_SCALABLE if ops.w3s_meta_scalable(pair[1], pair[2], 2 * I, K) else 0)
    else:
        po = ops.OPT_META_SCALABLE if ops.meta_scalable(pair[1], pair[2], 2 * I, K, 64, nbits) else 0
    g = torch.Generator(device="cuda").manual_seed(I)
    h = (torch.randn(1, K, device="cuda", generator=g) * 0.8).to(dt)
    w = (1 + 0.1 * torch.randn(K, device="cuda", generator=g)).to(dt)
    eps = 1e-6
    a = torch.full((1, I), float("nan"), dtype=dt, device="cuda")
    ops.gemv_block(h, w, eps, [pair], K, 64, nbits, [a], ops.BLOCK_NORM | ops.BLOCK_SILU, opts=po)
    for use_sub in ((True, False) if po & ops.OPT_META_SCALABLE else (False,)):     # the three-op rebuild gives the four-op one's bits
        a2 = torch.empty_like(a)
        ops.gemv_block(h, w, eps, [pair], K, 64, nbits, [a2], ops.BLOCK_NORM | ops.BLOCK_SILU, opts=po if use_sub else po & ~ops.OPT_META_SCALABLE)
        assert torch.equal(a, a2)
    # the launches it replaces
    xk = ops.add_rmsnorm(h.clone(), None, w, eps)
    yg, yu = ops.gemv_grouped(xk, [gate[:3] + (None, I), up[:3] + (None, I)], K, 64, nbits, opts=(og & ou) | (og & ops.OPT_W3S))
    a_k = ops.silu_mul(yg, yu)
    assert bool(_near(a, a_k, 4.0).all()) and int((a != a_k).sum()) <= max(8, I // 4), int((a != a_k).sum())
    # and the arithmetic itself: fp64 matmul on HF's normalised row, silu in fp32 on the rounded gate, product in T
    xn = _rmsnorm_hf(h, w, eps)
    gg = (xn.double() @ Wg.double().t()).float().to(dt)
    uu = (xn.double() @ Wu.double().t()).float().to(dt)
    want = torch.nn.functional.silu(gg.float()).to(dt) * uu
    tol = dict(rtol=2.0 ** -6, atol=4e-3) if dt == torch.bfloat16 else dict(rtol=4e-3, atol=2e-3)
    torch.testing.assert_close(a.float(), want.float(), **tol)


def test_gemv_block_argument_errors_are_loud(ops):
    L, opts, _ = _layer(ops, 64, 64, 4, seed=1, dt=torch.float16)
    x = torch.zeros(1, 64, dtype=torch.float16, device="cuda")
    y = torch.zeros(1, 64, dtype=torch.float16, device="cuda")
    with pytest.raises(RuntimeError):
        ops.gemv_block(x, None, 0.0, [L], 64, 64, 4, [y], ops.BLOCK_NORM)                 # NORM without a weight
    with pytest.raises(RuntimeError):
        ops.gemv_block(x, None, 0.0, [L, L], 64, 64, 4, [y, y], ops.BLOCK_RESID)         # the residual epilogue serves one layer
    with pytest.raises(RuntimeError):
        ops.gemv_block(x, x, 0.0, [L], 64, 64, 4, [y], ops.BLOCK_RESID | ops.BLOCK_NORM)  # not a combination
    with pytest.raises(NotImplementedError):
        ops.gemv_block(x, x, 0.0, [L], 64, 32, 4, [y], ops.BLOCK_NORM)                    # group_size 32
    with pytest.raises(ValueError):
        ops.gemv_block(torch.zeros(2, 64, dtype=torch.float16, device="cuda"), x, 0.0, [L], 64, 64, 4, [y], ops.BLOCK_NORM)   # one activation row


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("nbits", [4, 3, 2])
@pytest.mark.parametrize("shape", [(32, 32, 128, 4096), (8, 2, 64, 1024), (4, 4, 64, 256), (16, 4, 128, 8192)])   # (n_heads, n_kv_heads, head_dim, hidden)
def test_rotary_epilogue_equals_rope_cache_on_the_same_projections(ops, dt, nbits, shape):
    """q | k | v with q and k in the rotary-paired row order: the launch's epilogue must leave exactly what hqq_hip_rope_cache leaves when it is run on
    the projections of the launch WITHOUT that epilogue (same kernel text, same prologue: the rows' results are the same bits, permuted) — rotated q,
    the key / value caches at the position, nothing else touched; and a position outside the cache writes nothing"""
    nh, nkv, hd, K = shape
    L = 96
    q, oq, _ = _layer(ops, nh * hd, K, nbits, seed=11 + nbits + K, dt=dt)
    k, ok_, _ = _layer(ops, nkv * hd, K, nbits, seed=12 + nbits + K, dt=dt, sub_friendly=False)
    v, ov, _ = _layer(ops, nkv * hd, K, nbits, seed=13 + nbits + K, dt=dt)
    w3 = nbits == 3
    qp = ops.rotary_pair_layout(q, K, 64, nbits, hd, w3s=w3)
    kp = ops.rotary_pair_layout(k, K, 64, nbits, hd, w3s=w3)
    base = ops.OPT_W3S if w3 else 0
    g = torch.Generator(device="cuda").manual_seed(K + nh)
    h = (torch.randn(1, K, device="cuda", generator=g) * 1.3).to(dt)
    w = (1 + 0.1 * torch.randn(K, device="cuda", generator=g)).to(dt)
    cos = torch.randn(hd, device="cuda", generator=g).to(dt)
    sin = torch.randn(hd, device="cuda", generator=g).to(dt)
    eps = 1e-5
    yq = torch.empty(1, nh * hd, dtype=dt, device="cuda"); yk = torch.empty(1, nkv * hd, dtype=dt, device="cuda"); yv = torch.empty_like(yk)
    ops.gemv_block(h, w, eps, [q, k, v], K, 64, nbits, [yq, yk, yv], ops.BLOCK_NORM, opts=base)
    for pos_v in (0, 37, L - 1, L, -1):
        pos = torch.tensor([pos_v], device="cuda")
        kc1 = torch.full((nkv, L, hd), 7.0, dtype=dt, device="cuda"); vc1 = torch.full((nkv, L, hd), 5.0, dtype=dt, device="cuda")
        kc2, vc2 = kc1.clone(), vc1.clone()
        qr1 = torch.empty(1, nh, 1, hd, dtype=dt, device="cuda"); qr2 = torch.full_like(qr1, float("nan"))
        ops.rope_cache(yq, yk, yv, cos, sin, pos, kc1, vc1, qr1)
        ops.gemv_block(h, w, eps, [qp, kp, v], K, 64, nbits, [qr2, kc2, vc2], ops.BLOCK_NORM | ops.BLOCK_ROPE, opts=base, rope=(cos, sin, pos, hd, L))
        assert torch.equal(qr1, qr2), pos_v
        assert torch.equal(kc1, kc2) and torch.equal(vc1, vc2), pos_v
        if not 0 <= pos_v < L:
            assert bool((kc2 == 7.0).all()) and bool((vc2 == 5.0).all())


@pytest.mark.parametrize("nbits", [4, 3, 2])
def test_folded_launches_against_the_oracles_own_weights(ops, oracle, nbits):
    """The tests above contract with weights rebuilt by the HIP dequantise kernel (itself pinned to the goldens); here the whole chain goes through the CPU
    oracle instead: levels packed by oracle.pack, weights by oracle.dequantize (Quantizer.dequantize restated, hqq/core/quantize.py:183-199),
    contraction by oracle.matmul (double accumulation) — for the RMSNorm prologue (q|k|v form), the SiLU * up epilogue and the residual epilogue, fp16."""
    dt, K, eps = torch.float16, 2048, 1e-5
    g = torch.Generator().manual_seed(40 + nbits)

    def olayer(N, seed):
        gg = torch.Generator().manual_seed(seed)
        R = N * K // 64
        U = torch.randint(0, 2 ** nbits, (R, 64), generator=gg, dtype=torch.uint8)
        sc = (torch.rand(R, 1, generator=gg) * 0.004 + 0.001).to(dt)
        z = (torch.rand(R, 1, generator=gg) * (2 ** nbits - 1)).float().clamp_min(0.0625).to(dt)
        P = oracle.pack(nbits, U.numpy())                                                         # BitPack.pack_* restated (hqq/core/bitpack.py)
        Wd = oracle.dequantize(nbits, P, sc.numpy(), z.numpy(), N, K, 64, 1)                      # [N, K] fp16, the reference's two roundings
        Wq = torch.from_numpy(P).cuda()
        sd, zd = sc.cuda(), z.cuda()
        opts = 0
        if nbits == 3:
            Wq = ops.w3s_pack(Wq, N, K)
            opts = ops.OPT_W3S | (ops.OPT_META_SCALABLE if ops.w3s_meta_scalable(sd, zd, N, K) else 0)
        elif ops.meta_scalable(sd, zd, N, K, 64, nbits):
            opts = ops.OPT_META_SCALABLE
        return (Wq, sd, zd, N), opts, Wd

    h = (torch.randn(1, K, generator=g) * 1.3).to(dt)
    w = (1 + 0.1 * torch.randn(K, generator=g)).to(dt)
    xn = _rmsnorm_hf(h, w, eps)                                                                   # LlamaRMSNorm with its roundings, on the CPU
    # RMSNorm prologue, three layers in one launch
    Ls, Wds, opts = [], [], None
    for i, N in enumerate((512, 128, 128)):
        L, o, Wd = olayer(N, 900 + 10 * nbits + i)
        Ls.append(L); Wds.append(Wd)
        opts = o if opts is None else (opts & o) | (o & ops.OPT_W3S)
    outs = [torch.full((1, L[3]), float("nan"), dtype=dt, device="cuda") for L in Ls]
    ops.gemv_block(h.cuda(), w.cuda(), eps, Ls, K, 64, nbits, outs, ops.BLOCK_NORM, opts=opts)
    for y, Wd in zip(outs, Wds):
        _, y32 = oracle.matmul(xn.numpy(), Wd, None, 1)
        np.testing.assert_allclose(y.float().cpu().numpy(), y32, rtol=1e-3, atol=1e-3)
    # SiLU * up epilogue on the paired layer
    I = 256
    gate, og, Wg = olayer(I, 950 + nbits)
    up, ou, Wu = olayer(I, 960 + nbits)
    pair = ops.pair_layers(gate, up, K, 64, nbits, w3s=(nbits == 3))
    po = (ops.OPT_W3S if nbits == 3 else 0) | ((ops.OPT_META_SCALABLE) if (ops.w3s_meta_scalable(pair[1], pair[2], 2 * I, K) if nbits == 3 else ops.meta_scalable(pair[1], pair[2], 2 * I, K, 64, nbits)) else 0)
    a = torch.full((1, I), float("nan"), dtype=dt, device="cuda")
    ops.gemv_block(h.cuda(), w.cuda(), eps, [pair], K, 64, nbits, [a], ops.BLOCK_NORM | ops.BLOCK_SILU, opts=po)
    yg, _ = oracle.matmul(xn.numpy(), Wg, None, 1)                                               # T(gate_proj(x)), T(up_proj(x)): the linears' own outputs
    yu, _ = oracle.matmul(xn.numpy(), Wu, None, 1)
    tg, tu = torch.from_numpy(yg.astype(np.float16)), torch.from_numpy(yu.astype(np.float16))
    want = (torch.nn.functional.silu(tg.float()).to(dt) * tu).float()                            # LlamaMLP: act_fn(gate) * up with its roundings
    assert bool(_near(a.cpu(), want.to(dt), ulps=4.0).all())                                      # (an ulp of the fp32-accumulated gate / up moves the product by up to two)
    # residual epilogue
    Lr, orr, Wr = olayer(384, 970 + nbits)
    x = torch.randn(1, K, generator=g).to(dt)
    hres = (torch.randn(1, 384, generator=g) * 3).to(dt)
    yr, y32 = oracle.matmul(x.numpy(), Wr, None, 1)
    hd = hres.cuda()
    ops.gemv_block(x.cuda(), None, 0.0, [Lr], K, 64, nbits, [hd], ops.BLOCK_RESID, opts=orr)
    want_h = hres.float() + torch.from_numpy(y32)                                                 # h + y in exact arithmetic: the kernel rounds y to T, then the sum to T
    np.testing.assert_allclose(hd.float().cpu().numpy(), want_h.numpy(), rtol=2e-3, atol=4e-3)
