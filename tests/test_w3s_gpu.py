"""GPU parity tests of the 3-bit STREAM layout (hqq_amd/csrc/w3s.h): the patch-time re-layout of the reference's 3-bit container and the
decode kernels that run on it — through the C ABI, against the CPU oracle (oracle/hqq_oracle.c + the numpy restatement of the layout).

Bar: the re-layout is bit-exact both ways (state_dict() of a patched layer carries the reference's bytes); a one-hot activation row
reads a column of the dequantised matrix BIT FOR BIT (the weights the kernels multiply are the reference's: two roundings in the
compute dtype, hqq/core/quantize.py:198); forward within rtol = atol = 1e-3 (fp16) of the double-accumulated oracle.
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


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def raw16(t):
    return t.contiguous().view(torch.int16).numpy().view(np.uint16)


def _layer(N, K, seed, dt=torch.float16, sub_friendly=False):
    """random 3-bit levels + group constants; sub_friendly: zero-points that keep z 2^-9 exact (the three-op rebuild's condition)"""
    g = torch.Generator().manual_seed(seed)
    R = N * K // 64
    U = torch.randint(0, 8, (R, 64), generator=g, dtype=torch.uint8)
    s = (torch.rand(R, 1, generator=g) * 0.004 + 0.001).to(dt)
    z = (torch.rand(R, 1, generator=g) * 7).to(dt)
    if sub_friendly:
        z = z.float().clamp_min(0.0625).to(dt)   # lowest bit of every zero-point >= 2^-15: z 2^-9 is exact in fp16
    else:   # a few zero-points far below one level: q - z must still round once; and one whose lowest bit (2^-16) fails the meta check
        z.view(-1)[::5] = 0.00836
        z.view(-1)[1::11] = 2.0 ** -12
        z.view(-1)[2] = 2.0 ** -6 * (1 + 2.0 ** -10)
    return U, s, z


def _oracle_W(oracle, U, s, z, N, K, code):
    P = oracle.pack(3, U.numpy())
    if code == 2:
        return P, oracle.dequantize(3, P, raw16(s), raw16(z), N, K, 64, 2)
    return P, oracle.dequantize(3, P, s.numpy(), z.numpy(), N, K, 64, 1)


@pytest.mark.parametrize("NK", [(2, 64), (512, 1024), (130, 192), (1002, 4096), (4096, 4096), (64, 11008)])
def test_w3s_relayout_is_bit_exact_both_ways(ops, oracle, NK):
    N, K = NK
    U = torch.randint(0, 8, (N * K // 64, 64), generator=torch.Generator().manual_seed(N + K), dtype=torch.uint8)
    P = oracle.pack(3, U.numpy())                                       # the reference's container (step not row-aligned for most shapes)
    want = oracle.w3s_pack_np(P, N, K).reshape(N // 2, -1).view(np.int32)
    got = ops.w3s_pack(dev(P), N, K)
    assert got.shape == (N // 2, K // 16 * 3) and got.dtype == torch.int32
    assert np.array_equal(got.cpu().numpy(), want)
    back = ops.w3s_unpack(got, N, K)
    assert np.array_equal(back.cpu().numpy(), P)                        # zero padding rows included
    with pytest.raises(ValueError):
        ops.w3s_pack(dev(P)[:-1], N, K)


# every position of a chunk, both slabs: k = 16 c + i over different chunks, plus the ends of the row
def _probe_ks(K):
    ks = [(16 * (3 * i + 1) + i) % K for i in range(16)] + [0, K - 1, K // 2 + 5]
    return sorted(set(ks))


@pytest.mark.parametrize("M", [1, 2, 3, 4])
@pytest.mark.parametrize("NK", [(512, 1024), (130, 192), (12, 128), (1002, 4096), (4096, 4096), (64, 11008), (256, 2048 + 768)])
def test_w3s_gemv_vs_oracle(ops, oracle, M, NK):
    """1..4 activation rows, fp16: the row-per-wave decode kernel on the stream layout (general four-op rebuild: one zero-point fails the meta check)"""
    N, K = NK
    U, s, z = _layer(N, K, seed=N + K + M)
    P, Wd = _oracle_W(oracle, U, s, z, N, K, 1)
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(1)).half()
    bias = torch.randn(N, generator=torch.Generator().manual_seed(2)).half() if M % 2 else None
    yo, _ = oracle.matmul(x.numpy(), Wd, None if bias is None else bias.numpy(), 1)
    W3, sd, zd = ops.w3s_pack(dev(P), N, K), s.cuda(), z.cuda()
    if N * K // 64 > 2:
        assert not ops.w3s_meta_scalable(sd, zd, N, K)
    y = ops.gemv(x.cuda(), W3, sd, zd, None if bias is None else bias.cuda(), N, K, 64, 3, opts=ops.OPT_W3S)
    torch.testing.assert_close(y.float().cpu(), torch.from_numpy(yo.astype(np.float32)), rtol=1e-3, atol=1e-3)
    assert torch.equal(y, ops.forward(x.cuda(), W3, sd, zd, None if bias is None else bias.cuda(), N, K, 64, 3, opts=ops.OPT_W3S))
    Wdev = ops.dequantize(dev(P), sd.reshape(-1), zd.reshape(-1), N, K, 64, 3)
    assert np.array_equal(Wdev.cpu().numpy().view(np.uint16), Wd.view(np.uint16))
    ks = _probe_ks(K)
    for i in range(0, len(ks), M):   # M one-hot rows per launch
        e = torch.zeros(M, K, dtype=torch.float16, device="cuda")
        for r, k in enumerate(ks[i:i + M]): e[r, k] = 1.0
        ye = ops.gemv(e, W3, sd, zd, None, N, K, 64, 3, opts=ops.OPT_W3S)
        for r, k in enumerate(ks[i:i + M]):
            assert torch.equal(ye[r], Wdev[:, k]), f"column {k} (chunk position {k % 16})"


@pytest.mark.parametrize("M", [1, 4])
@pytest.mark.parametrize("NK", [(512, 1024), (130, 192), (4096, 4096), (64, 11008)])
def test_w3s_three_op_rebuild_same_bits(ops, oracle, M, NK):
    """HQQ_OPT_META_SCALABLE on the stream layout: three field offsets per slab, three scalings of (zero, scale) — the same bits as the four-op form"""
    N, K = NK
    U, s, z = _layer(N, K, seed=N + K, sub_friendly=True)
    P, Wd = _oracle_W(oracle, U, s, z, N, K, 1)
    W3, sd, zd = ops.w3s_pack(dev(P), N, K), s.cuda(), z.cuda()
    assert ops.w3s_meta_scalable(sd, zd, N, K)
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(3)).half().cuda()
    y4 = ops.gemv(x, W3, sd, zd, None, N, K, 64, 3, opts=ops.OPT_W3S)
    y3 = ops.gemv(x, W3, sd, zd, None, N, K, 64, 3, opts=ops.OPT_W3S | ops.OPT_META_SCALABLE)
    assert torch.equal(y3, y4)
    yo, _ = oracle.matmul(x.cpu().numpy(), Wd, None, 1)
    torch.testing.assert_close(y3.float().cpu(), torch.from_numpy(yo.astype(np.float32)), rtol=1e-3, atol=1e-3)
    Wdev = ops.dequantize(dev(P), sd.reshape(-1), zd.reshape(-1), N, K, 64, 3)
    for k in _probe_ks(K):
        e = torch.zeros(1, K, dtype=torch.float16, device="cuda"); e[0, k] = 1.0
        assert torch.equal(ops.gemv(e, W3, sd, zd, None, N, K, 64, 3, opts=ops.OPT_W3S | ops.OPT_META_SCALABLE)[0], Wdev[:, k]), f"column {k}"
    if K % 256 == 0 and K >= 512:   # and in the skinny GEMM
        x32 = torch.randn(32, K, generator=torch.Generator().manual_seed(4)).half().cuda()
        assert torch.equal(ops.gemv(x32, W3, sd, zd, None, N, K, 64, 3, opts=ops.OPT_W3S | ops.OPT_META_SCALABLE), ops.gemv(x32, W3, sd, zd, None, N, K, 64, 3, opts=ops.OPT_W3S))


@pytest.mark.parametrize("M", [1, 3, 4])
@pytest.mark.parametrize("NK", [(512, 1024), (130, 192), (64, 11008)])
def test_w3s_gemv_bf16_vs_oracle(ops, oracle, M, NK):
    N, K = NK
    U, s, z = _layer(N, K, seed=N + K + 3, dt=torch.bfloat16)
    P, Wd = _oracle_W(oracle, U, s, z, N, K, 2)
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(1)).bfloat16()
    bias = torch.randn(N, generator=torch.Generator().manual_seed(2)).bfloat16() if M % 2 else None
    yo, _ = oracle.matmul(raw16(x), Wd, None if bias is None else raw16(bias), 2)
    want = torch.from_numpy(yo.view(np.int16).copy()).view(torch.bfloat16).float()
    W3, sd, zd = ops.w3s_pack(dev(P), N, K), s.cuda(), z.cuda()
    y = ops.gemv(x.cuda(), W3, sd, zd, None if bias is None else bias.cuda(), N, K, 64, 3, opts=ops.OPT_W3S)
    assert y.dtype == torch.bfloat16
    torch.testing.assert_close(y.float().cpu(), want, rtol=2.0 ** -7, atol=2e-3)
    Wdev = ops.dequantize(dev(P), sd.reshape(-1), zd.reshape(-1), N, K, 64, 3)
    assert np.array_equal(raw16(Wdev.cpu()), Wd)
    for k in _probe_ks(K):
        e = torch.zeros(1, K, dtype=torch.bfloat16, device="cuda"); e[0, k] = 1.0
        assert torch.equal(ops.gemv(e, W3, sd, zd, None, N, K, 64, 3, opts=ops.OPT_W3S)[0], Wdev[:, k]), f"column {k}"


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M", [5, 16, 17, 32, 33, 64])
@pytest.mark.parametrize("NK", [(512, 1024), (200, 2048 + 768), (64, 11008), (4096 + 8, 512), (334, 1024)])
def test_w3s_skinny_gemm_vs_oracle(ops, oracle, dt, M, NK):
    """5..64 activation rows: the weight-streaming skinny GEMM on the stream layout (both tiles, split-K, ragged last panels)"""
    N, K = NK
    code = 2 if dt == torch.bfloat16 else 1
    assert ops.skinny_covers(dt, M, N, K, 64, 3, w3s=True) and not ops.skinny_covers(dt, M, N, K, 64, 3)
    U, s, z = _layer(N, K, seed=N + K + 7, dt=dt)
    P, Wd = _oracle_W(oracle, U, s, z, N, K, code)
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(1)).to(dt)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(2)).to(dt) if M % 2 else None
    if code == 2:
        yo, _ = oracle.matmul(raw16(x), Wd, None if bias is None else raw16(bias), 2)
        want = torch.from_numpy(yo.view(np.int16).copy()).view(torch.bfloat16).float()
    else:
        yo, _ = oracle.matmul(x.numpy(), Wd, None if bias is None else bias.numpy(), 1)
        want = torch.from_numpy(yo.astype(np.float32))
    W3, sd, zd = ops.w3s_pack(dev(P), N, K), s.cuda(), z.cuda()
    args = (W3, sd, zd, None if bias is None else bias.cuda(), N, K, 64, 3)
    y = ops.forward(x.cuda(), *args, opts=ops.OPT_W3S)
    if code == 2:
        torch.testing.assert_close(y.float().cpu(), want, rtol=2.0 ** -7, atol=2e-3)
    else:
        torch.testing.assert_close(y.float().cpu(), want, rtol=1e-3, atol=1e-3)
    assert torch.equal(y, ops.gemv(x.cuda(), *args, opts=ops.OPT_W3S))
    assert torch.equal(y[:5], ops.gemv(x[:5].cuda(), *args, opts=ops.OPT_W3S))     # a row does not depend on the batch it is computed in
    Wdev = ops.dequantize(dev(P), sd.reshape(-1), zd.reshape(-1), N, K, 64, 3)
    ks = _probe_ks(K)[:M]
    e = torch.zeros(M, K, dtype=dt, device="cuda")
    for r, k in enumerate(ks): e[r, k] = 1.0
    ye = ops.gemv(e, W3, sd, zd, None, N, K, 64, 3, opts=ops.OPT_W3S)
    for r, k in enumerate(ks):
        assert torch.equal(ye[r], Wdev[:, k]), f"column {k}"


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(65, 256, 256), (200, 384, 512), (128, 4096, 4096), (1000, 512, 4096), (640, 128, 11008)])
def test_w3s_pipelined_gemm_vs_oracle(ops, oracle, dt, M, N, K):
    """beyond 64 rows: the pipelined split-K MFMA GEMM (gemm_pipe.hip) on the stream layout — LDS-DMA in 12-byte pieces, the same rebuild"""
    code = 2 if dt == torch.bfloat16 else 1
    U, s, z = _layer(N, K, seed=N + K + M, dt=dt, sub_friendly=(M % 2 == 0))
    P, Wd = _oracle_W(oracle, U, s, z, N, K, code)
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(1)).to(dt)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(2)).to(dt) if M % 2 else None
    if code == 2:
        yo, _ = oracle.matmul(raw16(x), Wd, None if bias is None else raw16(bias), 2)
        want = torch.from_numpy(yo.view(np.int16).copy()).view(torch.bfloat16).float()
    else:
        yo, _ = oracle.matmul(x.numpy(), Wd, None if bias is None else bias.numpy(), 1)
        want = torch.from_numpy(yo.astype(np.float32))
    W3, sd, zd = ops.w3s_pack(dev(P), N, K), s.cuda(), z.cuda()
    o = ops.OPT_W3S | (ops.OPT_META_SCALABLE if (code == 1 and ops.w3s_meta_scalable(sd, zd, N, K)) else 0)
    y = ops.forward(x.cuda(), W3, sd, zd, None if bias is None else bias.cuda(), N, K, 64, 3, fused=True, opts=o)
    if code == 2:
        torch.testing.assert_close(y.float().cpu(), want, rtol=2.0 ** -7, atol=2e-3)
    else:
        torch.testing.assert_close(y.float().cpu(), want, rtol=1e-3, atol=1e-3)
    if o & ops.OPT_META_SCALABLE:   # the three-op rebuild gives the four-op form's bits
        assert torch.equal(y, ops.forward(x.cuda(), W3, sd, zd, None if bias is None else bias.cuda(), N, K, 64, 3, fused=True, opts=ops.OPT_W3S))
    Wdev = ops.dequantize(dev(P), sd.reshape(-1), zd.reshape(-1), N, K, 64, 3)
    ks = _probe_ks(K)
    e = torch.zeros(M, K, dtype=dt, device="cuda")
    for r, k in enumerate(ks): e[r, k] = 1.0
    ye = ops.forward(e, W3, sd, zd, None, N, K, 64, 3, fused=True, opts=o)
    for r, k in enumerate(ks):
        assert torch.equal(ye[r], Wdev[:, k]), f"column {k}"
    assert torch.count_nonzero(ye[len(ks):]) == 0


def test_w3s_meta_check_reports_a_zero_point_below_the_grid(ops):
    N, K = 16, 128
    s = torch.full((N * K // 64, 1), 0.003, dtype=torch.float16, device="cuda")
    z = torch.full((N * K // 64, 1), 3.5, dtype=torch.float16, device="cuda")
    assert ops.w3s_meta_scalable(s, z, N, K)
    z[7] = 2.0 ** -6 * (1 + 2.0 ** -10)     # lowest bit 2^-16: z 2^-9 is not a multiple of 2^-24
    assert not ops.w3s_meta_scalable(s, z, N, K)
    z[7] = 2.0 ** -5 * (1 + 2.0 ** -10)     # lowest bit 2^-15: exact
    assert ops.w3s_meta_scalable(s, z, N, K)
    s[3] = 200.0                            # scale 2^9 overflows fp16
    assert not ops.w3s_meta_scalable(s, z, N, K)


def test_w3s_grouped_launch_equals_single_launches(ops, oracle):
    K = 1024
    Ns = (512, 128, 130)
    Ls = []
    for i, N in enumerate(Ns):
        U, s, z = _layer(N, K, seed=90 + i, sub_friendly=True)
        P = oracle.pack(3, U.numpy())
        Ls.append((ops.w3s_pack(dev(P), N, K), s.cuda(), z.cuda(), None, N))
    for M in (1, 4, 32):
        x = torch.randn(M, K, generator=torch.Generator().manual_seed(M)).half().cuda()
        for o in (ops.OPT_W3S, ops.OPT_W3S | ops.OPT_META_SCALABLE):
            outs = ops.gemv_grouped(x, Ls, K, 64, 3, opts=o)
            for (W3, s, z, _, N), y in zip(Ls, outs):
                assert torch.equal(y, ops.gemv(x, W3, s, z, None, N, K, 64, 3, opts=o))


def test_w3s_errors_are_loud(ops):
    W3 = torch.zeros(8, 12, dtype=torch.int32, device="cuda")
    s = torch.ones(16, 1, dtype=torch.float16, device="cuda")
    x = torch.zeros(1, 64, dtype=torch.float16, device="cuda")
    with pytest.raises(RuntimeError):   # the layout bit on another bit width
        ops.gemv(x, W3.view(torch.uint8), s, s, None, 16, 64, 64, 4, opts=ops.OPT_W3S)
    with pytest.raises(NotImplementedError):   # odd number of output rows
        ops.gemv(x, W3, s, s, None, 15, 64, 64, 3, opts=ops.OPT_W3S)
    with pytest.raises(RuntimeError):
        ops.w3s_pack(torch.zeros(2, 64, dtype=torch.int32, device="cuda"), 15, 64)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_patched_3bit_layer_keeps_the_reference_bytes(ops, oracle, dt):
    """prepare_for_inference(backend='hip') re-lays a 3-bit layer out; state_dict() still carries the reference's container byte for byte,
    load_state_dict takes it, dequantize() equals the oracle, forward agrees at 1 / 32 / 100 rows"""
    from hqq_amd.backends.hip import HQQLinearHIP, patch_hqq_to_hip
    from hqq_amd.core.quantize import BaseQuantizeConfig, HQQLinear
    N, K = 256, 512
    lin = torch.nn.Linear(K, N, bias=True)
    torch.manual_seed(0)
    lin.weight.data = torch.randn(N, K) * 0.02
    layer = HQQLinear(lin, BaseQuantizeConfig(nbits=3, group_size=64, axis=1), compute_dtype=dt, device="cuda")
    Wq_ref = layer.W_q.data.clone()
    meta_s, meta_z = layer.meta["scale"].clone(), layer.meta["zero"].clone()
    W_ref = layer.dequantize()
    ys = {M: layer(torch.randn(M, K, generator=torch.Generator().manual_seed(M)).to(dt).cuda()) for M in (1, 32, 100)}
    new = patch_hqq_to_hip(layer)
    assert isinstance(new, HQQLinearHIP) and new.w3s and new.W_q.shape == (N // 2, K // 16 * 3)
    sd = new.state_dict()
    assert sd["W_q"].dtype == torch.int32 and torch.equal(sd["W_q"], Wq_ref)          # the reference's bytes
    assert torch.equal(sd["scale"].reshape(-1), meta_s.reshape(-1)) and torch.equal(sd["zero"].reshape(-1), meta_z.reshape(-1))
    assert torch.equal(new.dequantize(), W_ref)
    P = Wq_ref.cpu().numpy()
    code = 2 if dt == torch.bfloat16 else 1
    Wd = oracle.dequantize(3, P, raw16(meta_s.cpu()) if code == 2 else meta_s.cpu().numpy(), raw16(meta_z.cpu()) if code == 2 else meta_z.cpu().numpy(), N, K, 64, code)
    assert np.array_equal(raw16(new.dequantize().cpu()), Wd.view(np.uint16))
    for M, y in ys.items():
        x = torch.randn(M, K, generator=torch.Generator().manual_seed(M)).to(dt).cuda()
        torch.testing.assert_close(new(x).float(), y.float(), rtol=2.0 ** -7 if code == 2 else 1e-3, atol=2e-3)
    # a second layer loads the reference-format state dict and computes the same bits
    W3_before = new.W_q.data.clone()
    new.W_q.data.zero_()
    new.load_state_dict(sd)
    assert torch.equal(new.W_q.data, W3_before)


def test_w3s_pack_kernel_matches_the_golden_fixture(ops):
    """hqq_hip_w3s_pack / _unpack against tests/golden/w3s_layout.npz (the REFERENCE's container re-laid out by the oracle): byte for byte"""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "w3s_layout.npz"))
    for name in ("a", "b", "c"):
        N, K = (int(v) for v in g[f"{name}_shape"])
        ref = torch.from_numpy(g[f"{name}_ref"].copy()).cuda()
        want = g[f"{name}_w3s"]
        if not ops.w3s_covers(N, K, 64):
            continue
        got = ops.w3s_pack(ref, N, K)
        assert np.array_equal(got.cpu().numpy().view(np.uint32).reshape(want.shape), want)
        back = ops.w3s_unpack(got, N, K)
        assert np.array_equal(back.cpu().numpy().view(np.int32), g[f"{name}_ref"])
