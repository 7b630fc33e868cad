"""GPU tests of the in-tree dense MFMA GEMM (hqq_amd/csrc/gemm_dense.hip, hqq_hip_gemm_dense): HQQLinear.matmul on dequantised weights
(hqq/core/quantize.py:880-882) without a library call — against the oracle's double-accumulated matmul and, bit for bit, against its own
properties (row independence, one-hot probes), at ragged and full sizes, fp16 and bf16."""
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


def raw16(t):
    return t.contiguous().view(torch.int16).numpy().view(np.uint16)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(1, 4, 64), (256, 256, 64), (300, 260, 128), (1000, 512, 4096), (2048, 4096, 1024), (257, 11008, 192), (8192, 256, 4096), (513, 1028, 2816), (130, 264, 64), (64, 24, 128)])
def test_dense_gemm_vs_oracle(ops, oracle, dt, M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(dt)
    W = (torch.randn(N, K, generator=g) * 0.05).to(dt)
    bias = torch.randn(N, generator=g).to(dt) if M % 2 else None
    code = 2 if dt == torch.bfloat16 else 1
    if code == 2:
        yo, _ = oracle.matmul(raw16(x), raw16(W), None if bias is None else raw16(bias), 2)
        want = torch.from_numpy(yo.view(np.int16).copy()).view(torch.bfloat16).float()
    else:
        yo, _ = oracle.matmul(x.numpy(), W.numpy(), None if bias is None else bias.numpy(), 1)
        want = torch.from_numpy(yo.astype(np.float32))
    y = ops.gemm_dense(x.cuda(), W.cuda(), None if bias is None else bias.cuda())
    assert y.dtype == dt and tuple(y.shape) == (M, N)
    # the forward tolerance + one ulp of the output dtype at the reference value AND at the matmul result before the bias add (both are stored
    # rounded — quantize.py:896-897 rounds the product, then the sum: a value at a rounding boundary lands one ulp apart under any other
    # summation order, and a cancelling bias leaves that ulp standing beside a small result)
    yf = y.float().cpu()
    pre = want if bias is None else want - bias.float()
    def ulp_of(v):
        return torch.pow(2.0, torch.floor(torch.log2(v.abs().clamp_min(2.0 ** -14))) - (7 if code == 2 else 10))
    bad = (yf - want).abs() > (1e-3 + 1e-3 * want.abs()) + ulp_of(want) + (0 if bias is None else 2 * ulp_of(pre))
    assert not bool(bad.any()), f"{int(bad.sum())} of {bad.numel()} outside tolerance"
    # reproducible, and a row does not depend on the batch it is computed in (a tile's rows are independent)
    assert torch.equal(y, ops.gemm_dense(x.cuda(), W.cuda(), None if bias is None else bias.cuda()))
    k = min(M, 37)
    assert torch.equal(y[:k], ops.gemm_dense(x[:k].cuda(), W.cuda(), None if bias is None else bias.cuda()))
    # one-hot probes: y[m, :] = W[:, k] exactly (one product, no rounding before the output's)
    ks = sorted({0, K - 1, (3 * K) // 7, 63, 64 % K, K // 2 + 5})
    e = torch.zeros(len(ks), K, dtype=dt, device="cuda")
    for r, kk in enumerate(ks): e[r, kk] = 1.0
    ye = ops.gemm_dense(e, W.cuda())
    for r, kk in enumerate(ks):
        assert torch.equal(ye[r], W[:, kk].cuda()), f"column {kk}"


@pytest.mark.parametrize("nbits", [4, 2, 8])
def test_long_prompt_forward_takes_the_in_tree_gemm_and_matches_the_oracle(ops, oracle, nbits):
    """ops.forward beyond the fused kernels' range: hqq_hip_dequantize + hqq_hip_gemm_dense (no torch.matmul) == the oracle on a row sample,
    and within the forward tolerance of the library composition"""
    N, K, M = 512, 1024, 3072
    g = torch.Generator().manual_seed(nbits)
    U = torch.randint(0, 2 ** nbits, (N * K // 64, 64), generator=g, dtype=torch.uint8)
    s = (torch.rand(N * K // 64, 1, generator=g) * 0.004 + 0.001).half()
    z = (torch.rand(N * K // 64, 1, generator=g) * (2 ** nbits - 1)).half()
    P = oracle.pack(nbits, U.numpy())
    Wd = oracle.dequantize(nbits, P, s.numpy(), z.numpy(), N, K, 64, 1)
    x = torch.randn(M, K, generator=g).half()
    args = (torch.from_numpy(P).cuda(), s.cuda(), z.cuda(), None, N, K, 64, nbits)
    real_matmul = torch.matmul
    calls = []
    torch.matmul = lambda *a, **k: (calls.append(1), real_matmul(*a, **k))[1]
    try:
        y = ops.forward(x.cuda(), *args)              # M = 3072: beyond gemm_pipe's preferred range -> dequantise + in-tree GEMM
    finally:
        torch.matmul = real_matmul
    assert not calls, "ops.forward reached torch.matmul"
    assert not ops._C.lib().hqq_hip_forward_prefers_fused(nbits, M, N, K, 64, 1)
    rows = [0, 1, 255, 256, 1000, 2047, 3071]
    yo, _ = oracle.matmul(x[rows].numpy(), Wd, None, 1)
    torch.testing.assert_close(y[rows].float().cpu(), torch.from_numpy(yo.astype(np.float32)), rtol=1e-3, atol=1e-3)
    yl = ops.forward(x.cuda(), *args, library_gemm=True)
    torch.testing.assert_close(y.float(), yl.float(), rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("nbits,gs,K,M", [(8, 32, 1024, 100), (8, 128, 1024, 17), (4, 64, 960, 300), (2, 64, 1088, 2560)])
def test_batches_of_layers_the_fused_kernels_do_not_cover_stay_in_tree(ops, oracle, nbits, gs, K, M):
    """17..2560 rows through a layer the fused GEMMs do not cover (8-bit with another group size; K % 128 != 0): dequantise kernel + the in-tree MFMA GEMM
    since round 6 — torch.matmul is never reached (missing 3) — and the result is the oracle's"""
    N = 384
    g = torch.Generator().manual_seed(nbits * 1000 + gs)
    R = N * K // gs
    U = torch.randint(0, 2 ** nbits, (R, gs), generator=g, dtype=torch.uint8)
    s = (torch.rand(R, 1, generator=g) * 0.004 + 0.001).half()
    z = (torch.rand(R, 1, generator=g) * (2 ** nbits - 1)).half()
    P = oracle.pack(nbits, U.numpy())
    Wd = oracle.dequantize(nbits, P, s.numpy(), z.numpy(), N, K, gs, 1)
    x = torch.randn(M, K, generator=g).half()
    bias = torch.randn(N, generator=g).half()
    assert not ops._C.lib().hqq_hip_forward_prefers_fused(nbits, M, N, K, gs, 1)
    real_matmul, calls = torch.matmul, []
    torch.matmul = lambda *a, **k: (calls.append(1), real_matmul(*a, **k))[1]
    try:
        y = ops.forward(x.cuda(), torch.from_numpy(P).cuda(), s.cuda(), z.cuda(), bias.cuda(), N, K, gs, nbits)
    finally:
        torch.matmul = real_matmul
    assert not calls, "ops.forward reached torch.matmul"
    rows = sorted({0, 1, M // 2, M - 1})
    yo, _ = oracle.matmul(x[rows].numpy(), Wd, bias.numpy(), 1)
    torch.testing.assert_close(y[rows].float().cpu(), torch.from_numpy(yo.astype(np.float32)), rtol=1e-3, atol=2e-3)


def test_configs2_prompt_of_65536_tokens_whole_and_in_chunks(ops, oracle):
    """BASELINE.json configs[2]: 65,536 prefill tokens through a 4-bit 4096 x 4096 layer — in one forward call (weights rebuilt once) and as
    8 chunks of 8192: identical bit for bit (rows are independent), and equal to the oracle on a row sample"""
    N, K, M, nbits = 4096, 4096, 65536, 4
    g = torch.Generator().manual_seed(65536)
    U = torch.randint(0, 16, (N * K // 64, 64), generator=g, dtype=torch.uint8)
    s = (torch.rand(N * K // 64, 1, generator=g) * 0.004 + 0.001).half()
    z = (torch.rand(N * K // 64, 1, generator=g) * 15).half()
    P = oracle.pack(nbits, U.numpy())
    Wd = oracle.dequantize(nbits, P, s.numpy(), z.numpy(), N, K, 64, 1)
    x = torch.randn(M, K, generator=g).half().cuda()
    args = (torch.from_numpy(P).cuda(), s.cuda(), z.cuda(), None, N, K, 64, nbits)
    y = ops.forward(x, *args)
    yc = torch.empty_like(y)
    for c0 in range(0, M, 8192):
        ops.forward(x[c0:c0 + 8192], *args, out=yc[c0:c0 + 8192])
    assert torch.equal(y, yc)
    rows = [0, 255, 256, 8191, 8192, 30000, 32767, 32768, 50001, 65279, 65280, 65535]
    yo, _ = oracle.matmul(x[rows].cpu().numpy(), Wd, None, 1)
    torch.testing.assert_close(y[rows].float().cpu(), torch.from_numpy(yo.astype(np.float32)), rtol=1e-3, atol=1e-3)
