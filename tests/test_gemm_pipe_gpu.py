"""GPU parity tests of the pipelined split-K fused GEMM (csrc/gemm_pipe.hip: 65..1024 activation rows; all operands by LDS-DMA,
weights rebuilt into MFMA fragments) against the oracle and against hqq_hip_dequantize."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "gpu tests need a ROCm device"
    from hqq_amd import ops as o
    assert o.is_available(), "libhqq_hip.so must load on the GPU box (no fallback)"
    return o


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _layer(N, K, nbits, seed, round_zero):
    g = torch.Generator().manual_seed(seed)
    R = N * K // 64
    U = torch.randint(0, 2 ** nbits, (R, 64), generator=g, dtype=torch.uint8)
    s = (torch.rand(R, 1, generator=g) * 0.004 + 0.001).half()
    z = torch.rand(R, 1, generator=g) * (2 ** nbits - 1)
    z = (z.round() if round_zero else z).half()
    return U, s, z


@pytest.mark.parametrize("nbits", [8, 4, 2])
@pytest.mark.parametrize("M,N,K", [(65, 256, 128), (128, 200, 256), (129, 1024, 512), (300, 264, 1024), (512, 128, 2048), (96, 72, 1280)])
@pytest.mark.parametrize("round_zero", [True, False])
def test_pipelined_gemm_vs_oracle(ops, oracle, nbits, M, N, K, round_zero):
    """y against the C oracle's dequantise + fp32-accumulate matmul (tolerance: fp32 summation order + one fp16 rounding, 1e-3 of the
    output scale), and the weights themselves bit for bit: one-hot activation rows pick single columns of W exactly.  round_zero=True
    layers pass hqq_hip_meta_check and take the three-op rebuild, the others the four-op one.  Shapes: ragged N (rows per slab not a
    multiple of the 64-row tile), M not a multiple of 128, K of two steps up to twenty."""
    per = 8 // nbits
    if N % per or (N // per) % 4:
        pytest.skip("N not packable at this width")
    U, s, z = _layer(N, K, nbits, M + N + K, round_zero)
    P = oracle.pack(nbits, U.numpy())
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(1)).half()
    bias = torch.randn(N, generator=torch.Generator().manual_seed(2)).half()
    Wd = oracle.dequantize(nbits, P, s.numpy(), z.numpy(), N, K, 64, 1)
    yo, _ = oracle.matmul(x.numpy(), Wd, bias.numpy(), 1)
    sd, zd, Pd = s.cuda(), z.cuda(), dev(P)
    scalable = ops.meta_scalable(sd, zd, N, K, 64, nbits)
    assert scalable == round_zero or not round_zero   # integer zero-points always qualify
    opts = ops.OPT_META_SCALABLE if scalable else 0
    y = ops.gemm(x.cuda(), Pd, sd, zd, bias.cuda(), N, K, 64, nbits, opts=opts)
    torch.testing.assert_close(y.float().cpu(), torch.from_numpy(yo.astype(np.float32)), rtol=1e-3, atol=2e-3)
    e = torch.zeros(M, K, dtype=torch.float16, device="cuda")
    ks = torch.arange(M, device="cuda") * 7 % K
    e[torch.arange(M, device="cuda"), ks] = 1.0
    Y = ops.gemm(e, Pd, sd, zd, None, N, K, 64, nbits, opts=opts)
    Wdev = ops.dequantize(Pd, sd.reshape(-1), zd.reshape(-1), N, K, 64, nbits)
    assert np.array_equal(Wdev.cpu().numpy().view(np.uint16), Wd.view(np.uint16))
    assert torch.equal(Y, Wdev[:, ks].t().contiguous())
    if scalable:   # the three-op and the four-op rebuild give the same bits
        assert torch.equal(Y, ops.gemm(e, Pd, sd, zd, None, N, K, 64, nbits, opts=0))


@pytest.mark.parametrize("nbits", [8, 4, 2])
@pytest.mark.parametrize("N,K", [(11008, 4096), (4096, 11008)])
def test_pipelined_gemm_8192_rows_full_size_layers_vs_oracle(ops, oracle, nbits, N, K):
    """the kernel at BASELINE.json configs[2]'s chunk size (8192 tokens: full rounds of 256-token tiles, no split) on the 7B MLP shapes,
    against the oracle on a row sample (the oracle's matmul over all 8192 rows would take minutes; rows are independent); one-hot rows:
    single weight columns, exactly"""
    M = 8192
    U, s, z = _layer(N, K, nbits, N + K + nbits, True)
    P = oracle.pack(nbits, U.numpy())
    Wd = oracle.dequantize(nbits, P, s.numpy(), z.numpy(), N, K, 64, 1)
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(3)).half()
    hot = [5, 4097, 8190]
    hot_k = [0, K // 2 + 3, K - 1]
    for r_, k_ in zip(hot, hot_k):
        x[r_] = 0
        x[r_, k_] = 1.0
    sd, zd, Pd = s.cuda(), z.cuda(), dev(P)
    opts = ops.OPT_META_SCALABLE if ops.meta_scalable(sd, zd, N, K, 64, nbits) else 0
    y = ops.gemm(x.cuda(), Pd, sd, zd, None, N, K, 64, nbits, opts=opts)
    rows = [0, 1, 127, 128, 255, 256, 2048, 4095, 4096, 6000, 8191]
    yo, _ = oracle.matmul(x[rows].numpy(), Wd, None, 1)
    torch.testing.assert_close(y[rows].float().cpu(), torch.from_numpy(yo.astype(np.float32)), rtol=1e-3, atol=2e-3)
    for r_, k_ in zip(hot, hot_k):
        assert np.array_equal(y[r_].cpu().numpy().view(np.uint16), np.ascontiguousarray(Wd[:, k_]).view(np.uint16)), f"row {r_}: column {k_} of W"


@pytest.mark.parametrize("nbits", [4, 2, 8])
def test_split_k_is_reproducible_and_order_is_fixed(ops, nbits):
    """K = 11008 (172 steps: uneven splits, every forced split count), bias, one-hot exactness for every split count; two runs of the
    same call give the same bits (fixed summation order, no atomics); different split counts agree to fp32 rounding"""
    M, N, K = 160, 512, 11008
    U, s, z = _layer(N, K, nbits, 7, True)
    P = ops.pack(nbits, U.cuda())
    s, z = s.cuda(), z.cuda()
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(3)).half().cuda()
    bias = torch.randn(N, generator=torch.Generator().manual_seed(4)).half().cuda()
    Wd = ops.dequantize(P, s.reshape(-1), z.reshape(-1), N, K, 64, nbits)
    ref = (x.float() @ Wd.float().t() + bias.float())
    e = torch.zeros(M, K, dtype=torch.float16, device="cuda")
    cols = torch.arange(M, device="cuda") * 67 % K
    e[torch.arange(M, device="cuda"), cols] = 1.0
    base = None
    for ks in (0, 1, 2, 3, 5, 8, 10, 16):
        o = ops.OPT_META_SCALABLE | (ks << 24)
        y = ops.gemm(x, P, s, z, bias, N, K, 64, nbits, opts=o)
        assert torch.equal(y, ops.gemm(x, P, s, z, bias, N, K, 64, nbits, opts=o))
        torch.testing.assert_close(y.float(), ref, rtol=1e-3, atol=2e-3 * float(ref.abs().max()) / 4)
        if base is None:
            base = y
        torch.testing.assert_close(y.float(), base.float(), rtol=2e-3, atol=1e-2)
        assert torch.equal(ops.gemm(e, P, s, z, None, N, K, 64, nbits, opts=o), Wd[:, cols].t().contiguous())


@pytest.mark.parametrize("nbits", [8, 4, 2])
def test_every_tile_shape_forced(ops, oracle, nbits):
    """the three tile shapes (4 waves x 128 tokens, 8 x 128, 8 x 256; the last not at 2 bits) forced on ragged M and N: same exact
    weights (one-hot probes), results within the oracle tolerance, and the plan's own choice equals one of them bit for bit"""
    for (M, N, K) in ((300, 272, 1024), (520, 1040, 512)):
        U, s, z = _layer(N, K, nbits, M + N, True)
        P = oracle.pack(nbits, U.numpy())
        x = torch.randn(M, K, generator=torch.Generator().manual_seed(1)).half()
        bias = torch.randn(N, generator=torch.Generator().manual_seed(2)).half()
        yo, _ = oracle.matmul(x.numpy(), oracle.dequantize(nbits, P, s.numpy(), z.numpy(), N, K, 64, 1), bias.numpy(), 1)
        Pd, sd, zd, xd, bd = dev(P), s.cuda(), z.cuda(), x.cuda(), bias.cuda()
        Wdev = ops.dequantize(Pd, sd.reshape(-1), zd.reshape(-1), N, K, 64, nbits)
        e = torch.zeros(M, K, dtype=torch.float16, device="cuda")
        cols = torch.arange(M, device="cuda") * 13 % K
        e[torch.arange(M, device="cuda"), cols] = 1.0
        outs = []
        for tile in (ops.OPT_GEMM_NARROW, ops.OPT_GEMM_WIDE, ops.OPT_GEMM_NARROW | ops.OPT_GEMM_WIDE):
            o = ops.OPT_META_SCALABLE | tile
            y = ops.gemm(xd, Pd, sd, zd, bd, N, K, 64, nbits, opts=o)
            torch.testing.assert_close(y.float().cpu(), torch.from_numpy(yo.astype(np.float32)), rtol=2e-3, atol=2e-3)   # (one fp16 ulp: fp32 summation order vs the oracle's double)
            assert torch.equal(ops.gemm(e, Pd, sd, zd, None, N, K, 64, nbits, opts=o), Wdev[:, cols].t().contiguous())
            outs.append(y)
        auto = ops.gemm(xd, Pd, sd, zd, bd, N, K, 64, nbits, opts=ops.OPT_META_SCALABLE)
        assert any(torch.equal(auto, y) for y in outs)


@pytest.mark.parametrize("nbits", [8, 4, 2])
@pytest.mark.parametrize("M,N,K", [(65, 256, 128), (300, 272, 1024), (520, 1040, 512), (130, 512, 11008)])
def test_pipelined_gemm_bf16(ops, oracle, nbits, M, N, K):
    """bf16 compute dtype (two bf16 roundings per weight, quantize.py:198 on bf16 tensors): one-hot probes bit-exact against the bf16
    dequantise kernel (itself bit-exact against the oracle), results within one bf16 ulp (2^-7 relative, fp32 summation order) of the
    oracle's double-accumulated matmul; every tile shape forced"""
    per = 8 // nbits
    g = torch.Generator().manual_seed(M + N + K + nbits)
    R = N * K // 64
    U = torch.randint(0, 2 ** nbits, (R, 64), generator=g, dtype=torch.uint8)
    s = (torch.rand(R, 1, generator=g) * 0.004 + 0.001).bfloat16()
    z = (torch.rand(R, 1, generator=g) * (2 ** nbits - 1)).bfloat16()
    P = oracle.pack(nbits, U.numpy())
    x = torch.randn(M, K, generator=g).bfloat16()
    bias = torch.randn(N, generator=g).bfloat16()
    z.view(-1)[::5] = 0.00836                                   # zero-points far below one level: q - z must still round once
    raw = lambda t: t.view(torch.int16).numpy().view(np.uint16)   # noqa: E731  (the oracle takes raw bf16 bits)
    Wd = oracle.dequantize(nbits, P, raw(s), raw(z), N, K, 64, 2)
    yo, _ = oracle.matmul(raw(x), Wd, raw(bias), 2)
    yo32 = torch.from_numpy(yo.view(np.int16).copy()).view(torch.bfloat16).float()
    Pd, sd, zd, xd, bd = dev(P), s.cuda(), z.cuda(), x.cuda(), bias.cuda()
    Wdev = ops.dequantize(Pd, sd.reshape(-1), zd.reshape(-1), N, K, 64, nbits)
    assert Wdev.dtype == torch.bfloat16 and np.array_equal(Wdev.view(torch.int16).cpu().numpy().view(np.uint16), np.asarray(Wd).reshape(N, K))
    e = torch.zeros(M, K, dtype=torch.bfloat16, device="cuda")
    cols = torch.arange(M, device="cuda") * 29 % K
    e[torch.arange(M, device="cuda"), cols] = 1.0
    for tile in (0, ops.OPT_GEMM_NARROW, ops.OPT_GEMM_WIDE, ops.OPT_GEMM_NARROW | ops.OPT_GEMM_WIDE):
        y = ops.gemm(xd, Pd, sd, zd, bd, N, K, 64, nbits, opts=tile)
        assert y.dtype == torch.bfloat16
        # one bf16 ulp of the matmul result BEFORE the bias add (out = bf16(bf16(acc) + bias): a last-bit flip of bf16(acc) survives the add)
        bound = 2.0 ** -7 * (yo32.abs() + bias.float().abs()[None, :]) + 2e-3
        assert bool(((y.float().cpu() - yo32).abs() <= bound).all())
        assert torch.equal(ops.gemm(e, Pd, sd, zd, None, N, K, 64, nbits, opts=tile), Wdev[:, cols].t().contiguous())


@pytest.mark.parametrize("nbits,M,N,K", [(4, 2176, 4096, 4096), (4, 1664, 5120, 4096), (8, 1150, 4096, 4096), (4, 3328, 5120, 2048)])
def test_hybrid_plan_full_rounds_whole_last_round_split(ops, nbits, M, N, K):
    """more tiles than CUs with a partly filled last round: the plan runs the full rounds unsplit and splits only the last round's
    tiles (hqq_hip_gemm_plan reports it) — same exact weights on every row of y (one-hot probes incl. rows served by split tiles),
    results equal to the unsplit plan's up to fp32 summation order, reproducible bits"""
    import ctypes
    from hqq_amd import _C
    plan = (ctypes.c_int * 8)()
    assert _C.lib().hqq_hip_gemm_plan(nbits, M, N, K, 64, 1, 0, plan) == 0
    assert plan[6] > 0 and plan[6] % 256 == 0 and plan[4] > 1, list(plan)      # a hybrid plan on a 256-CU device
    U, s, z = _layer(N, K, nbits, M + N + K, True)
    P = ops.pack(nbits, U.cuda())
    s, z = s.cuda(), z.cuda()
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(2)).half().cuda()
    bias = torch.randn(N, generator=torch.Generator().manual_seed(3)).half().cuda()
    y = ops.gemm(x, P, s, z, bias, N, K, 64, nbits, opts=ops.OPT_META_SCALABLE)
    assert torch.equal(y, ops.gemm(x, P, s, z, bias, N, K, 64, nbits, opts=ops.OPT_META_SCALABLE))
    y1 = ops.gemm(x, P, s, z, bias, N, K, 64, nbits, opts=ops.OPT_META_SCALABLE | ops.OPT_GEMM_NOHYBRID)
    torch.testing.assert_close(y.float(), y1.float(), rtol=2e-3, atol=2e-2)
    Wd = ops.dequantize(P, s.reshape(-1), z.reshape(-1), N, K, 64, nbits)
    torch.testing.assert_close(y.float(), x.float() @ Wd.float().t() + bias.float(), rtol=2e-3, atol=2e-3 * float(y.float().abs().max()) / 4 + 2e-3)
    e = torch.zeros(M, K, dtype=torch.float16, device="cuda")
    cols = torch.arange(M, device="cuda") * 37 % K
    e[torch.arange(M, device="cuda"), cols] = 1.0
    assert torch.equal(ops.gemm(e, P, s, z, None, N, K, 64, nbits, opts=ops.OPT_META_SCALABLE), Wd[:, cols].t().contiguous())


def test_full_size_layers_one_hot_exact_and_linear(ops):
    """Llama-2-7B shapes at 128 and 1000 rows: every weight the kernel multiplies is the dequantised weight (one-hot rows), and the
    result is linear in x — size-independent properties, no oracle needed"""
    for (N, K) in [(4096, 4096), (11008, 4096), (4096, 11008)]:
        U, s, z = _layer(N, K, 4, N ^ K, True)
        P = ops.pack(4, U.cuda())
        s, z = s.cuda(), z.cuda()
        Wd = ops.dequantize(P, s.reshape(-1), z.reshape(-1), N, K, 64, 4)
        for M in (128, 1000):
            e = torch.zeros(M, K, dtype=torch.float16, device="cuda")
            cols = torch.arange(M, device="cuda") * 131 % K
            e[torch.arange(M, device="cuda"), cols] = 1.0
            assert torch.equal(ops.forward(e, P, s, z, None, N, K, 64, 4, fused=True, opts=ops.OPT_META_SCALABLE), Wd[:, cols].t().contiguous())
            x = torch.randn(M, K, generator=torch.Generator().manual_seed(M)).half().cuda()
            y = ops.forward(x, P, s, z, None, N, K, 64, 4, fused=True, opts=ops.OPT_META_SCALABLE).float()
            torch.testing.assert_close(y, x.float() @ Wd.float().t(), rtol=1e-3, atol=2e-3)


def test_routing_workspace_and_variants(ops):
    """forward() takes the fused kernel where the ABI's hint says it wins and the composition elsewhere — same weights either way;
    the output-tile kernels stay reachable (OPT_GEMM_CLASSIC); a split-K call without its workspace is refused, not computed wrong"""
    from hqq_amd import _C
    L = _C.lib()
    assert L.hqq_hip_forward_prefers_fused(4, 128, 4096, 4096, 64, 1) == 1
    assert L.hqq_hip_forward_prefers_fused(4, 640, 22016, 4096, 64, 1) == 1
    assert L.hqq_hip_forward_prefers_fused(4, 1024, 4096, 4096, 64, 1) == 1
    assert L.hqq_hip_forward_prefers_fused(4, 2048, 11008, 4096, 64, 1) == 1    # ahead of dequantise + in-tree GEMM to ~2000 tokens
    assert L.hqq_hip_forward_prefers_fused(4, 4096, 4096, 4096, 64, 1) == 0     # long prompts: rebuild the weights once, stream them as fp16
    assert L.hqq_hip_forward_prefers_fused(4, 128, 4096, 4096, 128, 1) == 0     # group_size 128: not this kernel
    assert L.hqq_hip_forward_prefers_fused(3, 128, 4096, 4096, 64, 1) == 0
    assert L.hqq_hip_forward_workspace_bytes(4, 128, 4096, 4096, 64, 1, 0) > 0 and L.hqq_hip_forward_workspace_bytes(4, 8192, 12288, 4096, 64, 1, 0) == 0
    N, K, M = 1024, 2048, 128
    U, s, z = _layer(N, K, 4, 11, True)
    P = ops.pack(4, U.cuda())
    s, z = s.cuda(), z.cuda()
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(5)).half().cuda()
    y = ops.forward(x, P, s, z, None, N, K, 64, 4)
    assert torch.equal(y, ops.gemm(x, P, s, z, None, N, K, 64, 4))
    comp = ops.forward(x, P, s, z, None, N, K, 64, 4, fused=False)
    classic = ops.gemm(x, P, s, z, None, N, K, 64, 4, opts=ops.OPT_GEMM_CLASSIC)
    torch.testing.assert_close(y.float(), comp.float(), rtol=2e-3, atol=1e-2)
    torch.testing.assert_close(y.float(), classic.float(), rtol=2e-3, atol=1e-2)
    out = torch.empty(M, N, dtype=torch.float16, device="cuda")
    rc = L.hqq_hip_gemm(4, x.data_ptr(), P.data_ptr(), s.data_ptr(), z.data_ptr(), None, out.data_ptr(), M, N, K, 64, 1, 0, None, 0, None)
    assert rc == -5 and b"workspace" in L.hqq_hip_last_error()
    # captured in a hipGraph (workspace reserved beforehand), replayed: same bits
    g = torch.cuda.CUDAGraph()
    yg = torch.empty_like(y)
    with torch.cuda.graph(g):
        ops.gemm(x, P, s, z, None, N, K, 64, 4, out=yg)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(yg, y)


@pytest.mark.parametrize("nbits", [8, 4, 2])
@pytest.mark.parametrize("M,Ns,K,dt", [(65, (256, 64, 64), 256, "f16"), (128, (200, 264), 512, "f16"), (300, (1024, 128, 128, 72), 1024, "f16"),
                                        (700, (512, 512), 2048, "bf16"), (130, (264,), 1280, "f16")])
def test_grouped_pipelined_gemm(ops, oracle, nbits, M, Ns, K, dt):
    """hqq_hip_gemm_grouped (round 6): layers that read the same x through ONE launch of the pipelined kernel over their concatenated feature tiles
    (layer widths that are NOT multiples of the tile: a tile never straddles two layers).  Per layer: the oracle's dequantise + fp32-accumulate matmul within
    the forward tolerance; one-hot rows pick the dequantise kernel's columns EXACTLY (the layer lookup and every offset are right, not just close); within
    tolerance of the same layer launched alone (another K split for the wider group: another association, not other weights)."""
    per = 8 // nbits
    if any(N % per or (N // per) % 4 for N in Ns):
        pytest.skip("N not packable at this width")
    cd = torch.float16 if dt == "f16" else torch.bfloat16
    code = 1 if dt == "f16" else 2
    raw = (lambda t: t.numpy()) if dt == "f16" else (lambda t: t.view(torch.int16).numpy().view(np.uint16))   # (the oracle takes raw bf16 bits)
    layers, ref = [], []
    for i, N in enumerate(Ns):
        U, s_, z_ = _layer(N, K, nbits, 17 * i + N + K, round_zero=(i % 2 == 0))
        s_, z_ = s_.to(cd), z_.to(cd)
        P = oracle.pack(nbits, U.numpy())
        bias = None if i == 1 else torch.randn(N, generator=torch.Generator().manual_seed(i)).to(cd)
        Wd = oracle.dequantize(nbits, P, raw(s_), raw(z_), N, K, 64, code)
        layers.append((dev(P), s_.cuda(), z_.cuda(), None if bias is None else bias.cuda(), N))
        ref.append((Wd, bias))
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(5)).to(cd)
    assert ops.gemm_grouped_covers(cd, list(Ns), M, K, 64, nbits)
    ys = ops.gemm_grouped(x.cuda(), layers, K, 64, nbits)
    for (Pd, sd, zd, b, N), (Wd, bias), y in zip(layers, ref, ys):
        assert tuple(y.shape) == (M, N) and y.dtype == cd
        yo, _ = oracle.matmul(raw(x), Wd, None if bias is None else raw(bias), code)   # (rounded as the kernels round: the product, then `+= bias`)
        want = torch.from_numpy(yo.astype(np.float32)) if dt == "f16" else torch.from_numpy(yo.view(np.int16).copy()).view(torch.bfloat16).float()
        # fp16: two roundings (the product, then `+= bias`) against the oracle's double-accumulated two: up to two ulps = 2^-9 of the pre-bias magnitude in ~1e-5 of
        # 400 k outputs (another fp32 summation order flips the first rounding, the add rounds again); bf16: one ulp of the result before the bias add (as test_pipelined_gemm_bf16)
        babs = 0 if bias is None else bias.float().abs()[None, :]
        bound = (2e-3 * (want.abs() + babs) + 2e-3) if dt == "f16" else (2.0 ** -7 * (want.abs() + babs) + 2e-3)
        assert float(((y.float().cpu() - want).abs() > 1e-3 * (want.abs() + babs) + 2e-3).float().mean()) < 1e-3   # ... and beyond ONE ulp almost nowhere
        assert bool(((y.float().cpu() - want).abs() <= bound).all()), float((y.float().cpu() - want).abs().max())
        alone = ops.gemm(x.cuda(), Pd, sd, zd, b, N, K, 64, nbits)
        assert bool(((y.float() - alone.float()).abs().cpu() <= 2 * bound).all())
    e = torch.zeros(M, K, dtype=cd, device="cuda")
    ks = torch.arange(M, device="cuda") * 11 % K
    e[torch.arange(M, device="cuda"), ks] = 1.0
    Ys = ops.gemm_grouped(e, [(P_, s_, z_, None, N_) for (P_, s_, z_, _, N_) in layers], K, 64, nbits)
    for (Pd, sd, zd, _, N), Y in zip(layers, Ys):
        Wdev = ops.dequantize(Pd, sd.reshape(-1), zd.reshape(-1), N, K, 64, nbits)
        assert torch.equal(Y, Wdev[:, ks].t().contiguous())
    # same call twice: same bits (fixed split order)
    ys2 = ops.gemm_grouped(x.cuda(), layers, K, 64, nbits)
    assert all(torch.equal(a_, b_) for a_, b_ in zip(ys, ys2))


def test_grouped_gemm_rejects_what_it_does_not_cover(ops):
    x = torch.zeros(100, 192, dtype=torch.float16, device="cuda")
    W = torch.zeros(64 * 192 // 2 // 64, 64, dtype=torch.uint8, device="cuda")
    sc = torch.ones(64 * 192 // 64, 1, dtype=torch.float16, device="cuda")
    assert not ops.gemm_grouped_covers(torch.float16, [64, 64], 100, 192, 64, 4)   # K % 128 != 0
    with pytest.raises((NotImplementedError, RuntimeError, ValueError)):
        ops.gemm_grouped(x, [(W, sc, sc, None, 64), (W, sc, sc, None, 64)], 192, 64, 4)
    with pytest.raises(ValueError):
        ops.gemm_grouped(x, [(W, sc, sc, None, 64)] * 5, 192, 64, 4)


def test_grouped_members_take_the_grouped_gemm_for_batches(ops):
    """group_projections: beyond the decode rows the siblings are served by ONE hqq_hip_gemm_grouped launch (the first member called computes all, the
    others return their parked output) — the same results as the layers called one by one, within the forward tolerance"""
    from hqq_amd.backends.hip import HQQLinearHIP, group_projections
    from hqq_amd.core.quantize import BaseQuantizeConfig, HQQLinear
    torch.manual_seed(11)
    K = 512
    cfg = BaseQuantizeConfig(nbits=4, group_size=64, axis=1)
    parent = torch.nn.Module()
    for name, N in (("q_proj", 512), ("k_proj", 128), ("v_proj", 128)):
        setattr(parent, name, HQQLinearHIP(HQQLinear(torch.nn.Linear(K, N, bias=(name == "q_proj")), cfg, compute_dtype=torch.float16, device="cuda")))
    singles = {n: getattr(parent, n) for n in ("q_proj", "k_proj", "v_proj")}
    assert group_projections(parent, ("q_proj", "k_proj", "v_proj"))
    calls = []
    real = ops.gemm_grouped
    ops.gemm_grouped = lambda *a_, **k_: (calls.append(1), real(*a_, **k_))[1]
    try:
        for rows in (200, 1, 70):
            x = torch.randn(rows, K, device="cuda").half()
            got = {n: getattr(parent, n)(x) for n in ("q_proj", "k_proj", "v_proj")}
            for n in got:
                torch.testing.assert_close(got[n].float(), singles[n](x).float(), rtol=1e-3, atol=2e-3)
    finally:
        ops.gemm_grouped = real
    assert len(calls) == 2   # 200 and 70 rows: one grouped GEMM launch each; the single row took the grouped decode kernel
