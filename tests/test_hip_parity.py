"""GPU parity tests: the HIP kernels (through the C ABI, hqq_amd.ops) against
  (a) golden vectors produced by the reference itself (tests/golden), and
  (b) the CPU oracle (oracle/hqq_oracle.c) on seeded inputs, and
  (c) size-independent properties at BASELINE.json's full sizes.
Bar: bit-exact for packed bytes and dequantised weights; forward within atol=rtol=1e-3 for fp16 (one fp16 ulp,
BLAS-order noise in the reference itself); solver: see test_quantize_*.
"""
import numpy as np
import pytest

from conftest import CD_CODE, load_golden

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

TD = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}
NB = [8, 4, 3, 2, 1]


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "gpu tests need a ROCm device"
    from hqq_amd import ops as o
    assert o.is_available(), "libhqq_hip.so must load on the GPU box (no fallback)"
    return o


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t if dtype is None else t.to(dtype)


def cd_tensor(raw, cdn):
    """golden raw array (np.float16 / uint16-bf16 / float32) -> cuda tensor of the compute dtype"""
    t = torch.from_numpy(np.ascontiguousarray(raw))
    if cdn == "bf16":
        t = t.view(torch.bfloat16)
    return t.cuda()


def max_ulp_f16(a, b):
    """largest distance between two fp16 tensors in units in the last place (monotone integer key of the bit patterns)"""
    def key(t):
        i = t.detach().contiguous().view(torch.int16).to(torch.int32)
        return torch.where(i < 0, -(i & 0x7FFF), i)
    return int((key(a) - key(b)).abs().max())


def ulp_f16(t):
    """spacing of fp16 numbers at |t| (t: float32 tensor)"""
    e = torch.floor(torch.log2(t.abs().clamp_min(2.0 ** -14)))
    return torch.pow(2.0, e - 10)


def assert_forward_parity(y, want, what="", slack=1.0):
    """The stated forward tolerance (BASELINE.json: "within 1e-3 fp16"): |y - ref| <= 1e-3 + 1e-3*|ref| for the
    arithmetic, plus one fp16 ulp of the reference value because BOTH sides are stored rounded to fp16 (each rounding
    moves a value by up to half an ulp, which alone is 4.9e-4 relative)."""
    y, want = y.float().cpu(), want.float().cpu()
    tol = slack * (1e-3 + 1e-3 * want.abs()) + ulp_f16(want)
    bad = (y - want).abs() > tol
    assert not bool(bad.any()), f"{what}: {int(bad.sum())} of {bad.numel()} outside tolerance, worst {float(((y - want).abs() - tol).max()):.3e} over"


def bits(t):
    t = t.detach().contiguous().cpu()
    if t.dtype == torch.bfloat16 or t.dtype == torch.float16:
        return t.view(torch.int16).numpy()
    if t.dtype == torch.float32:
        return t.view(torch.int32).numpy()
    return t.numpy()


# ------------------------------------------------------------------------------------------------
# BitPack
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("nbits", NB)
def test_pack_unpack_golden(ops, nbits):
    g = load_golden(f"pack_{nbits}b")
    i = 0
    while f"U{i}" in g:
        U, P = g[f"U{i}"], g[f"P{i}"]
        got = ops.pack(nbits, dev(U))
        assert got.dtype == (torch.int32 if nbits == 3 else torch.uint8)
        assert np.array_equal(got.cpu().numpy(), P)
        assert np.array_equal(ops.pack(nbits, dev(U).float()).cpu().numpy(), P)      # float levels, as the solver emits
        up = ops.unpack(nbits, dev(P))
        want = g[f"UP{i}"] if nbits == 3 else U
        assert np.array_equal(up.cpu().numpy()[: len(want)], want)
        for dt in (torch.float16, torch.bfloat16, torch.float32):                     # tests/test_bitpack.py:24-34
            assert torch.equal(ops.unpack(nbits, dev(P), dt)[: len(U)].cpu(), torch.from_numpy(U).to(dt))
        i += 1


@pytest.mark.parametrize("nbits", NB)
@pytest.mark.parametrize("shape", [(32, 32), (128, 256), (4096, 4096), (8192, 128), (32, 4096), (1001, 8), (70, 24)])
def test_pack_unpack_roundtrip_and_oracle(ops, oracle, nbits, shape):
    if nbits != 3 and shape[0] % ops.PER[nbits]:
        with pytest.raises(ValueError):
            ops.pack(nbits, torch.zeros(shape, dtype=torch.uint8, device="cuda"))
        return
    g = torch.Generator().manual_seed(42)
    U = torch.randint(0, 2 ** nbits, shape, generator=g, dtype=torch.uint8)
    P = ops.pack(nbits, U.cuda())
    assert torch.equal(ops.unpack(nbits, P)[: shape[0]].cpu(), U)
    assert np.array_equal(P.cpu().numpy(), oracle.pack(nbits, U.numpy()))   # every size: the oracle packs 16 M levels in a fraction of a second


def test_pack_empty(ops):
    assert ops.pack(4, torch.zeros((0, 64), dtype=torch.uint8, device="cuda")).shape == (0, 64)
    assert ops.unpack(4, torch.zeros((0, 64), dtype=torch.uint8, device="cuda")).shape == (0, 64)


# ------------------------------------------------------------------------------------------------
# Quantizer.dequantize
# ------------------------------------------------------------------------------------------------
QUANT_FILES = ([f"quant_4b_16x4096_gs{g}" for g in (512, 1024, 4096)] + ["quant_2b_16x4096_gs2048"] + [f"quant_{b}b_192x256" for b in NB] + [f"quant_{b}b_64x2048_normal" for b in (4, 3, 2)] +
               [f"quant_{b}b_16x128_edge" for b in (4, 3, 2)] + [f"quant_4b_32x256_gs{g}" for g in (32, 128, 256)])


@pytest.mark.parametrize("name", QUANT_FILES)
def test_dequantize_golden_bit_exact(ops, name):
    g = load_golden(name)
    nbits, gs = int(g["nbits"]), int(g["gs"])
    N, K = g["W"].shape
    Wq = dev(g["Wq_packed"])
    for cdn in CD_CODE:
        if f"Wdeq_{cdn}" not in g:
            continue
        s, z = cd_tensor(g[f"scale_{cdn}"], cdn), cd_tensor(g[f"zero_{cdn}"], cdn)
        Wd = ops.dequantize(Wq, s, z, N, K, gs, nbits, axis=1)
        assert Wd.dtype == TD[cdn]
        assert np.array_equal(bits(Wd), bits(cd_tensor(g[f"Wdeq_{cdn}"], cdn))), (name, cdn)


@pytest.mark.parametrize("nbits", NB)
def test_dequantize_axis0_matches_formula(ops, oracle, nbits):
    # axis=0 (hqq_aten's only mode, hqq_aten_cuda.cpp:35): unpacked matrix is [gs, N*K/gs], meta is [1, N*K/gs]
    N, K, gs = 48, 160, 64
    g = torch.Generator().manual_seed(7)
    R = N * K // gs
    U = torch.randint(0, 2 ** nbits, (gs, R), generator=g, dtype=torch.uint8)
    if nbits != 3 and gs % ops.PER[nbits]:
        pytest.skip("not packable")
    P = ops.pack(nbits, U.cuda())
    for dt in (torch.float16, torch.float32, torch.bfloat16):
        s = (torch.rand(1, R, generator=g) * 0.1 + 0.01).to(dt).cuda()
        z = (torch.rand(1, R, generator=g) * (2 ** nbits - 1)).to(dt).cuda()
        got = ops.dequantize(P, s.reshape(-1), z.reshape(-1), N, K, gs, nbits, axis=0)
        want = ((U.cuda().to(dt) - z) * s).reshape(N, K)      # Quantizer.dequantize, quantize.py:198
        assert torch.equal(got, want)


# ------------------------------------------------------------------------------------------------
# fused forward: GEMV (M<=8) and MFMA GEMM
# ------------------------------------------------------------------------------------------------
def _layer_from_golden(g, cdn):
    return dev(g["Wq_packed"]), cd_tensor(g[f"scale_{cdn}"], cdn), cd_tensor(g[f"zero_{cdn}"], cdn)


@pytest.mark.parametrize("name", ["quant_4b_192x256", "quant_2b_192x256", "quant_4b_64x2048_normal", "quant_2b_64x2048_normal",
                                  "quant_4b_16x128_edge", "quant_2b_16x128_edge", "quant_4b_32x256_gs32", "quant_4b_32x256_gs128",
                                  "quant_4b_32x256_gs256"])
def test_forward_golden(ops, name):
    g = load_golden(name)
    nbits, gs = int(g["nbits"]), int(g["gs"])
    N, K = g["W"].shape
    Wq, s, z = _layer_from_golden(g, "f16")
    x = dev(g["x_f32"]).half()
    b = dev(g["bias_f32"]).half() if "bias_f32" in g else None
    want = cd_tensor(g["y_f16"], "f16").float()
    y = ops.gemv(x, Wq, s, z, b, N, K, gs, nbits)       # default mode: exact weights on the MFMA path
    torch.testing.assert_close(y.float(), want, rtol=1e-3, atol=1e-3)
    if (N // ops.PER[nbits]) % 4 == 0 and K % 64 == 0:
        y2 = ops.gemm(x, Wq, s, z, b, N, K, gs, nbits)
        torch.testing.assert_close(y2.float(), want, rtol=1e-3, atol=1e-3)


def _random_layer(N, K, gs, nbits, seed, dt=torch.float16):
    g = torch.Generator().manual_seed(seed)
    R = N * K // gs
    U = torch.randint(0, 2 ** nbits, (R, gs), generator=g, dtype=torch.uint8)
    s = (torch.rand(R, 1, generator=g) * 0.004 + 0.001).to(dt)
    z = (torch.rand(R, 1, generator=g) * (2 ** nbits - 1)).to(dt)
    return U, s, z


@pytest.mark.parametrize("nbits", [4, 2])
@pytest.mark.parametrize("M", [1, 2, 3, 5, 8, 13, 16])
@pytest.mark.parametrize("NK", [(512, 1024), (256, 2048 + 768), (64, 11008), (40, 192)])
def test_gemv_vs_oracle(ops, oracle, nbits, M, NK):
    N, K = NK
    gs = 64
    U, s, z = _random_layer(N, K, gs, nbits, seed=N + K + nbits)
    P = oracle.pack(nbits, U.numpy())
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(1)).half()
    bias = torch.randn(N, generator=torch.Generator().manual_seed(2)).half() if M % 2 else None
    Wd = oracle.dequantize(nbits, P, s.numpy(), z.numpy(), N, K, gs, 1)
    yo, y32 = oracle.matmul(x.numpy(), Wd, None if bias is None else bias.numpy(), 1)
    y = ops.gemv(x.cuda(), dev(P), s.cuda(), z.cuda(), None if bias is None else bias.cuda(), N, K, gs, nbits)
    # fp32-accumulated result vs the double-accumulated oracle on identical (reference-exact) weights
    torch.testing.assert_close(y.float().cpu(), torch.from_numpy(yo.astype(np.float32)), rtol=1e-3, atol=1e-3)
    # and the fused kernel must agree with the dequant kernel bit-for-bit on a one-hot probe: y[n] = W[n,k]
    k = (3 * K) // 7
    e = torch.zeros(1, K, dtype=torch.float16, device="cuda"); e[0, k] = 1.0
    col = ops.gemv(e, dev(P), s.cuda(), z.cuda(), None, N, K, gs, nbits)[0]
    Wdev = ops.dequantize(dev(P), s.cuda().reshape(-1), z.cuda().reshape(-1), N, K, gs, nbits)
    assert torch.equal(col, Wdev[:, k])


@pytest.mark.parametrize("nbits", [8, 1])
@pytest.mark.parametrize("M", [1, 3, 7, 16])
def test_gemv_8bit_1bit_vs_oracle(ops, oracle, nbits, M):
    """the other byte containers of the reference (8-bit: also 5/6-bit levels; 1-bit: eight slabs per byte) on the same kernels"""
    N, K, gs = 256, 1024, 64
    U, s, z = _random_layer(N, K, gs, nbits, seed=nbits)
    P = oracle.pack(nbits, U.numpy())
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(1)).half()
    Wd = oracle.dequantize(nbits, P, s.numpy(), z.numpy(), N, K, gs, 1)
    yo, _ = oracle.matmul(x.numpy(), Wd, None, 1)
    y = ops.gemv(x.cuda(), dev(P), s.cuda(), z.cuda(), None, N, K, gs, nbits)
    torch.testing.assert_close(y.float().cpu(), torch.from_numpy(yo.astype(np.float32)), rtol=1e-3, atol=1e-3 * (16 if nbits == 8 else 1))
    e = torch.zeros(1, K, dtype=torch.float16, device="cuda"); e[0, 5] = 1.0
    Wdev = ops.dequantize(dev(P), s.cuda().reshape(-1), z.cuda().reshape(-1), N, K, gs, nbits)
    assert torch.equal(ops.gemv(e, dev(P), s.cuda(), z.cuda(), None, N, K, gs, nbits)[0], Wdev[:, 5])


@pytest.mark.parametrize("nbits", [8, 4, 2])
@pytest.mark.parametrize("M", [17, 32, 33, 64])
@pytest.mark.parametrize("NK", [(512, 1024), (200, 2048 + 768), (64, 11008), (4096 + 8, 512), (333, 1024)])
def test_skinny_gemm_vs_oracle(ops, oracle, nbits, M, NK):
    """decode with a batch of 17..64 rows (skinny.hip: weights streamed once, split-K partials summed in a fixed order by the
    last split to arrive)"""
    N, K = NK
    gs = 64
    if N % (8 // nbits): N += (8 // nbits) - N % (8 // nbits)   # (333: odd N for 8-bit, a ragged last panel for all)
    assert ops.skinny_covers(torch.float16, M, N, K, gs, nbits)
    U, s, z = _random_layer(N, K, gs, nbits, seed=N + K + nbits + 7)
    P = oracle.pack(nbits, U.numpy())
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(1)).half()
    bias = torch.randn(N, generator=torch.Generator().manual_seed(2)).half() if M % 2 else None
    Wd = oracle.dequantize(nbits, P, s.numpy(), z.numpy(), N, K, gs, 1)
    yo, y32 = oracle.matmul(x.numpy(), Wd, None if bias is None else bias.numpy(), 1)
    args = (dev(P), s.cuda(), z.cuda(), None if bias is None else bias.cuda(), N, K, gs, nbits)
    y = ops.forward(x.cuda(), *args)                       # M <= 64 on a covered config: the fused path, not the library composition
    assert_forward_parity(y, torch.from_numpy(yo.astype(np.float32)), "skinny gemm vs oracle")
    assert torch.equal(y, ops.gemv(x.cuda(), *args))
    # reproducible, and a row's result does not depend on the batch it is computed in (5 rows take the same kernel)
    assert torch.equal(y, ops.forward(x.cuda(), *args))
    assert torch.equal(y[:5], ops.gemv(x[:5].cuda(), *args))
    # one-hot probe: y[m, n] = W[n, k] bit for bit
    e = torch.zeros(M, K, dtype=torch.float16, device="cuda")
    ks = [(3 * K) // 7, 0, K - 1]
    for i, k in enumerate(ks): e[i * 5, k] = 1.0
    Wdev = ops.dequantize(dev(P), s.cuda().reshape(-1), z.cuda().reshape(-1), N, K, gs, nbits)
    ye = ops.gemv(e, dev(P), s.cuda(), z.cuda(), None, N, K, gs, nbits)
    for i, k in enumerate(ks): assert torch.equal(ye[i * 5], Wdev[:, k])
    assert torch.count_nonzero(ye[1]) == 0


@pytest.mark.parametrize("nbits", [8, 4, 2])
@pytest.mark.parametrize("M", [5, 16, 33, 64])
@pytest.mark.parametrize("NK", [(512, 1024), (200, 2048 + 768), (64, 11008)])
def test_skinny_gemm_bf16_vs_oracle(ops, oracle, nbits, M, NK):
    """the batched-decode kernel in bf16: weights rounded to bf16 twice as the reference does on bf16 tensors (one-hot probe
    bit-exact vs the dequant kernel), outputs within one bf16 ulp of the double-accumulated oracle"""
    N, K = NK
    gs = 64
    assert ops.skinny_covers(torch.bfloat16, M, N, K, gs, nbits)
    U, s, z = _random_layer(N, K, gs, nbits, seed=N + K + nbits + 3, dt=torch.bfloat16)
    z.view(-1)[::5] = 0.00836                                  # zero-points far below one level: q - z must still round once
    z.view(-1)[1::11] = 2.0 ** -12
    P = oracle.pack(nbits, U.numpy())
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(1)).bfloat16()
    bias = torch.randn(N, generator=torch.Generator().manual_seed(2)).bfloat16() if M % 2 else None
    raw = lambda t: t.view(torch.int16).numpy().view(np.uint16)                     # noqa: E731  (oracle takes raw bf16 bits)
    Wd = oracle.dequantize(nbits, P, raw(s), raw(z), N, K, gs, 2)
    yo, _ = oracle.matmul(raw(x), Wd, None if bias is None else raw(bias), 2)
    want = torch.from_numpy(yo.view(np.int16).copy()).view(torch.bfloat16).float()
    args = (dev(P), s.cuda(), z.cuda(), None if bias is None else bias.cuda(), N, K, gs, nbits)
    y = ops.forward(x.cuda(), *args)
    assert y.dtype == torch.bfloat16 and torch.equal(y, ops.gemv(x.cuda(), *args))
    torch.testing.assert_close(y.float().cpu(), want, rtol=2.0 ** -7, atol=2e-3)
    assert torch.equal(y[:5], ops.gemv(x[:5].cuda(), *args))
    e = torch.zeros(M, K, dtype=torch.bfloat16, device="cuda")
    ks = [(3 * K) // 7, 0, K - 1]
    for i, k in enumerate(ks): e[i, k] = 1.0
    Wdev = ops.dequantize(dev(P), s.cuda().reshape(-1), z.cuda().reshape(-1), N, K, gs, nbits)
    ye = ops.gemv(e, dev(P), s.cuda(), z.cuda(), None, N, K, gs, nbits)
    for i, k in enumerate(ks):
        assert torch.equal(ye[i], Wdev[:, k])
        if nbits != 8:   # the row-per-wave kernel too (its bf16 variant covers 4/2-bit)
            assert torch.equal(ops.gemv(e[i:i + 1], *args[:3], None, N, K, gs, nbits)[0], Wdev[:, k])


def test_skinny_split_k_reproducible_bits(ops):
    """the K splits of a panel finish in any order; whichever arrives last adds the partial tiles in split order, so two hundred
    launches of a heavily split layer (and of a grouped one) give the same bits, and the arrival counters are back at zero"""
    K, gs, M, nbits = 4096, 64, 32, 4
    layers = []
    for i, N in enumerate([4096, 1024, 1024]):
        U, s, z = _random_layer(N, K, gs, nbits, seed=900 + i)
        layers.append((ops.pack(nbits, U.cuda()), s.cuda(), z.cuda(), None, N))
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(4)).half().cuda()
    first = [y.clone() for y in ops.gemv_grouped(x, layers, K, gs, nbits)]
    single = ops.gemv(x, *layers[0][:4], 4096, K, gs, nbits).clone()
    # (a grouped call may pick another K split than the single call: same sums in another association, one fp16 ulp apart at most)
    torch.testing.assert_close(single.float(), first[0].float(), rtol=2.0 ** -10, atol=2e-3)
    for _ in range(200):
        for y, want in zip(ops.gemv_grouped(x, layers, K, gs, nbits), first): assert torch.equal(y, want)
        assert torch.equal(ops.gemv(x, *layers[0][:4], 4096, K, gs, nbits), single)


def test_skinny_gemm_grouped_and_capture(ops):
    """grouped launch == single launches; the split-K scratch is never grown inside a stream capture"""
    K, gs, M, nbits = 1024, 64, 24, 4
    layers = []
    for i, N in enumerate([512, 96, 40, 1024]):
        U, s, z = _random_layer(N, K, gs, nbits, seed=300 + i)
        b = torch.randn(N, generator=torch.Generator().manual_seed(i)).half().cuda() if i % 2 else None
        layers.append((ops.pack(nbits, U.cuda()), s.cuda(), z.cuda(), b, N))
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(9)).half().cuda()
    ys = ops.gemv_grouped(x, layers, K, gs, nbits)
    for (Wq, s, z, b, N), y in zip(layers, ys):
        assert torch.equal(y, ops.gemv(x, Wq, s, z, b, N, K, gs, nbits))
    # captured after a warm-up call of the same shape: replays give the same bits
    Wq, s, z, b, N = layers[3]
    out = torch.empty(M, N, dtype=torch.float16, device="cuda")
    want = ops.gemv(x, Wq, s, z, b, N, K, gs, nbits, out=out).clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
            ops.gemv(x, Wq, s, z, b, N, K, gs, nbits, out=out)
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want)


def test_skinny_scratch_is_not_grown_inside_a_capture(ops):
    """a split-K shape whose scratch does not exist yet fails loudly when first met inside a stream capture, and works afterwards"""
    N, K, gs, M, nbits = 64 * 2 * 40, 16384, 64, 64, 4          # 40 panels x 16 splits x 64 rows x N fp32 > the 8 MiB the scratch starts with
    U, s, z = _random_layer(N, K, gs, nbits, seed=77)
    Wq, s, z = ops.pack(nbits, U.cuda()), s.cuda(), z.cuda()
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(3)).half().cuda()
    out = torch.empty(M, N, dtype=torch.float16, device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    failed = False
    try:
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                ops.gemv(x, Wq, s, z, None, N, K, gs, nbits, out=out)
    except RuntimeError as e:
        failed = "outside stream capture" in str(e) or "scratch" in str(e)
    torch.cuda.synchronize()
    y = ops.gemv(x, Wq, s, z, None, N, K, gs, nbits, out=out)      # eager: allocates, then computes
    Wd = ops.dequantize(Wq, s.reshape(-1), z.reshape(-1), N, K, gs, nbits)
    torch.testing.assert_close(y.float(), x.float() @ Wd.float().t(), rtol=2e-3, atol=4e-3)
    # (`failed` is True when this test meets the initial 8 MiB scratch; had an earlier call grown it, the capture would simply succeed)
    assert isinstance(failed, bool)


@pytest.mark.parametrize("case", [(8, 221, 1024, 64, 48), (8, 167, 4352, 64, 5), (4, 786, 224, 16, 7), (2, 1208, 32, 32, 7), (3, 238, 704, 64, 13),
                                  (1, 4192, 736, 32, 5)])
def test_forward_decode_cases_outside_the_fused_kernels(ops, case):
    """found by tools/fuzz_forward.py: odd N at 8 bits (the skinny kernel's finish pass stores column pairs), 5..16 rows with
    K % 64 != 0, 3-bit beyond 4 rows — `forward` must compose these (HIP dequantise + library GEMM), not fail or mis-store"""
    nbits, N, K, gs, M = case
    U, s, z = _random_layer(N, K, gs, nbits, seed=sum(case))
    P = ops.pack(nbits, U.cuda())
    s, z = s.cuda(), z.cuda()
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(5)).half().cuda()
    b = torch.randn(N, generator=torch.Generator().manual_seed(6)).half().cuda()
    y = ops.forward(x, P, s, z, b, N, K, gs, nbits)
    Wd = ops.dequantize(P, s.reshape(-1), z.reshape(-1), N, K, gs, nbits)
    ref = (x.double() @ Wd.double().t()).half()
    want = (ref + b).float()
    tol = 2.0 ** -10 * (ref.float().abs() + want.abs()).clamp(min=2.0 ** -4) * 1.01 + 4e-7 * (x.float().abs() @ Wd.float().abs().t()) + 1e-4
    assert bool(((y.float() - want).abs() <= tol).all())
    if not ops.decode_covers(torch.float16, M, N, K, gs, nbits) and not ops.skinny_covers(torch.float16, M, N, K, gs, nbits) and nbits not in (4, 2):
        with pytest.raises(NotImplementedError):   # fused=True never composes: the uncovered configuration is reported
            ops.forward(x, P, s, z, b, N, K, gs, nbits, fused=True)


def test_forward_empty_and_shape_errors(ops):
    N, K, gs = 64, 128, 64
    U, s, z = _random_layer(N, K, gs, 4, seed=1)
    P = ops.pack(4, U.cuda())
    s, z = s.cuda(), z.cuda()
    y = ops.forward(torch.zeros(0, K, dtype=torch.float16, device="cuda"), P, s, z, None, N, K, gs, 4)
    assert tuple(y.shape) == (0, N)
    assert ops.gemv_grouped(torch.zeros(0, K, dtype=torch.float16, device="cuda"), [(P, s, z, None, N)], K, gs, 4)[0].shape == (0, N)
    y3 = ops.forward(torch.randn(2, 3, K, device="cuda").half(), P, s, z, None, N, K, gs, 4)      # leading dims are flattened and restored
    assert tuple(y3.shape) == (2, 3, N)
    with pytest.raises(ValueError):
        ops.forward(torch.zeros(1, K + 16, dtype=torch.float16, device="cuda"), P, s, z, None, N, K, gs, 4)
    with pytest.raises(TypeError):
        ops.forward(torch.zeros(1, K, dtype=torch.float32, device="cuda"), P, s, z, None, N, K, gs, 4)      # x must share the compute dtype
    with pytest.raises(RuntimeError):
        ops.forward(torch.zeros(1, K, dtype=torch.float16), P, s, z, None, N, K, gs, 4)                         # CPU tensor: no CPU path


@pytest.mark.parametrize("M", [1, 2, 4])
@pytest.mark.parametrize("NK", [(512, 1024), (101, 512), (1001, 4096), (64, 11008), (4096, 4096), (12, 128), (33, 192)])
def test_gemv_3bit_vs_oracle(ops, oracle, M, NK):
    """3-bit containers: ten unrelated rows per int32, step = ceil(R/10) not row-aligned (N = 101 / 1001: output rows
    straddle slab boundaries).  Exact weights: one-hot probe bit-identical to the dequant kernel."""
    N, K = NK
    gs, nbits = 64, 3
    U, s, z = _random_layer(N, K, gs, nbits, seed=N + K)
    P = oracle.pack(nbits, U.numpy())
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(1)).half()
    bias = torch.randn(N, generator=torch.Generator().manual_seed(2)).half() if M % 2 else None
    Pd, sd, zd = dev(P), s.cuda(), z.cuda()
    y = ops.gemv(x.cuda(), Pd, sd, zd, None if bias is None else bias.cuda(), N, K, gs, nbits)
    Wdev = ops.dequantize(Pd, sd.reshape(-1), zd.reshape(-1), N, K, gs, nbits)
    Wd = oracle.dequantize(nbits, P, s.numpy(), z.numpy(), N, K, gs, 1)   # the oracle at every size (a 4096 x 4096 forward takes ~50 ms)
    assert np.array_equal(Wdev.cpu().numpy().view(np.uint16), Wd.view(np.uint16))
    yo, _ = oracle.matmul(x.numpy(), Wd, None if bias is None else bias.numpy(), 1)
    want = torch.from_numpy(yo.astype(np.float32))
    torch.testing.assert_close(y.float().cpu(), want, rtol=1e-3, atol=1e-3)
    for k in ((3 * K) // 7, 0, K - 1):
        e = torch.zeros(1, K, dtype=torch.float16, device="cuda"); e[0, k] = 1.0
        assert torch.equal(ops.gemv(e, Pd, sd, zd, None, N, K, gs, nbits)[0], Wdev[:, k])
    with pytest.raises(NotImplementedError):
        ops.gemv(torch.zeros(5, K, dtype=torch.float16, device="cuda"), Pd, sd, zd, None, N, K, gs, nbits)


@pytest.mark.parametrize("M", [1, 3, 4])
@pytest.mark.parametrize("NK", [(512, 1024), (1001, 4096), (64, 11008), (4096, 4096), (173, 2048)])
def test_gemv_3bit_slab_sharing_kernel(ops, oracle, M, NK):
    """the kernel large 3-bit launches take (gemv3s.hip: each packed word loaded once for all ten slabs, per-task partial sums
    added in a fixed order by a second launch), forced here on small layers: exact weights (one-hot probes bit-identical to the
    dequant kernel), outputs within tolerance of the oracle / of the row-per-wave kernel, grouped == single, reproducible bits"""
    N, K = NK
    gs, nbits = 64, 3
    U, s, z = _random_layer(N, K, gs, nbits, seed=N + K + 5)
    P = oracle.pack(nbits, U.numpy())
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(1)).half().cuda()
    bias = torch.randn(N, generator=torch.Generator().manual_seed(2)).half().cuda() if M % 2 else None
    Pd, sd, zd = dev(P), s.cuda(), z.cuda()
    y1 = ops.gemv(x, Pd, sd, zd, bias, N, K, gs, nbits, opts=ops.OPT_GEMV3_ROWWISE).clone()
    SLABS = ops.OPT_GEMV3_SLABS
    y2 = ops.gemv(x, Pd, sd, zd, bias, N, K, gs, nbits, opts=SLABS).clone()
    torch.testing.assert_close(y2.float(), y1.float(), rtol=1e-3, atol=1e-3)
    Wdev = ops.dequantize(Pd, sd.reshape(-1), zd.reshape(-1), N, K, gs, nbits)
    want = x.float() @ Wdev.float().t() + (0 if bias is None else bias.float())
    torch.testing.assert_close(y2.float(), want, rtol=1e-3, atol=1e-3)
    for k in ((3 * K) // 7, 0, 1, 2, 3, 63, 64, K - 1):
        e = torch.zeros(M, K, dtype=torch.float16, device="cuda"); e[M - 1, k] = 1.0
        ye = ops.gemv(e, Pd, sd, zd, None, N, K, gs, nbits, opts=SLABS)
        assert torch.equal(ye[M - 1], Wdev[:, k])
        if M > 1: assert torch.count_nonzero(ye[0]) == 0
    # grouped launch (layers of different N share the task space and the workspace) == single launches; same bits every time
    U2, s2, z2 = _random_layer(N + 37, K, gs, nbits, seed=N + K + 6)
    P2 = dev(oracle.pack(nbits, U2.numpy()))
    layers = [(Pd, sd, zd, bias, N), (P2, s2.cuda(), z2.cuda(), None, N + 37)]
    ys = [t.clone() for t in ops.gemv_grouped(x, layers, K, gs, nbits, opts=SLABS)]
    assert torch.equal(ys[0], y2)
    assert torch.equal(ys[1], ops.gemv(x, P2, s2.cuda(), z2.cuda(), None, N + 37, K, gs, nbits, opts=SLABS))
    for _ in range(5): assert torch.equal(ops.gemv(x, Pd, sd, zd, bias, N, K, gs, nbits, opts=SLABS), y2)


@pytest.mark.parametrize("nbits", [4, 2])
@pytest.mark.parametrize("M", [1, 3, 4])
@pytest.mark.parametrize("NK", [(512, 1024), (64, 11008), (40, 192)])
def test_gemv_bf16_vs_oracle(ops, oracle, nbits, M, NK):
    """bf16 compute dtype: weights rounded to bf16 twice exactly as the reference does on bf16 tensors (one-hot probe is
    bit-exact vs the dequant kernel / oracle); outputs within one bf16 ulp (2^-8) of the double-accumulated oracle."""
    N, K = NK
    gs = 64
    U, s, z = _random_layer(N, K, gs, nbits, seed=N + K + nbits, dt=torch.bfloat16)
    P = oracle.pack(nbits, U.numpy())
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(1)).bfloat16()
    bias = torch.randn(N, generator=torch.Generator().manual_seed(2)).bfloat16() if M % 2 else None
    raw = lambda t: t.view(torch.int16).numpy().view(np.uint16)                     # noqa: E731  (oracle takes raw bf16 bits)
    Wd = oracle.dequantize(nbits, P, raw(s), raw(z), N, K, gs, 2)
    yo, _ = oracle.matmul(raw(x), Wd, None if bias is None else raw(bias), 2)
    want = torch.from_numpy(yo.view(np.int16).copy()).view(torch.bfloat16).float()
    y = ops.gemv(x.cuda(), dev(P), s.cuda(), z.cuda(), None if bias is None else bias.cuda(), N, K, gs, nbits)
    assert y.dtype == torch.bfloat16
    torch.testing.assert_close(y.float().cpu(), want, rtol=2.0 ** -7, atol=2e-3)
    k = (3 * K) // 7
    e = torch.zeros(1, K, dtype=torch.bfloat16, device="cuda"); e[0, k] = 1.0
    col = ops.gemv(e, dev(P), s.cuda(), z.cuda(), None, N, K, gs, nbits)[0]
    Wdev = ops.dequantize(dev(P), s.cuda().reshape(-1), z.cuda().reshape(-1), N, K, gs, nbits)
    assert torch.equal(col, Wdev[:, k])
    assert np.array_equal(bits(Wdev), Wd.view(np.int16))                              # and the dequant kernel == oracle, bit for bit


@pytest.mark.parametrize("nbits", [4, 2])
def test_gemv_grouped_equals_single_launches(ops, nbits):
    """q|k|v-style horizontal fusion: one launch over layers of different N sharing x == the per-layer launches, bit for bit"""
    K, gs, M = 1024, 64, 3
    Ns = [512, 96, 40, 1024]
    layers = []
    for i, N in enumerate(Ns):
        U, s, z = _random_layer(N, K, gs, nbits, seed=100 + i)
        b = torch.randn(N, generator=torch.Generator().manual_seed(i)).half().cuda() if i % 2 else None
        layers.append((ops.pack(nbits, U.cuda()), s.cuda(), z.cuda(), b, N))
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(9)).half().cuda()
    ys = ops.gemv_grouped(x, layers, K, gs, nbits)
    for (Wq, s, z, b, N), y in zip(layers, ys):
        assert torch.equal(y, ops.gemv(x, Wq, s, z, b, N, K, gs, nbits))
    with pytest.raises(ValueError):
        ops.gemv_grouped(x, layers + layers, K, gs, nbits)


@pytest.mark.parametrize("nbits", [4, 2])
@pytest.mark.parametrize("M,N,K", [(128, 256, 256), (200, 384, 512), (16, 128, 1024), (9, 1024, 128), (1000, 512, 4096)])
def test_gemm_vs_oracle(ops, oracle, nbits, M, N, K):
    gs = 64
    U, s, z = _random_layer(N, K, gs, nbits, seed=M + N + K)
    P = oracle.pack(nbits, U.numpy())
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(1)).half()
    bias = torch.randn(N, generator=torch.Generator().manual_seed(2)).half()
    Wd = oracle.dequantize(nbits, P, s.numpy(), z.numpy(), N, K, gs, 1)
    yo, _ = oracle.matmul(x.numpy(), Wd, bias.numpy(), 1)
    y = ops.gemm(x.cuda(), dev(P), s.cuda(), z.cuda(), bias.cuda(), N, K, gs, nbits)
    torch.testing.assert_close(y.float().cpu(), torch.from_numpy(yo.astype(np.float32)), rtol=1e-3, atol=2e-3)
    # asymmetric probe (transpose-detecting): x = one-hot rows picks single weight columns exactly
    e = torch.zeros(M, K, dtype=torch.float16, device="cuda")
    ks = torch.arange(M, device="cuda") * 7 % K
    e[torch.arange(M, device="cuda"), ks] = 1.0
    Y = ops.gemm(e, dev(P), s.cuda(), z.cuda(), None, N, K, gs, nbits)
    Wdev = ops.dequantize(dev(P), s.cuda().reshape(-1), z.cuda().reshape(-1), N, K, gs, nbits)
    assert torch.equal(Y, Wdev[:, ks].t().contiguous())


@pytest.mark.parametrize("nbits", [4, 2])
def test_gemm_register_tile_variant(ops, nbits):
    """opt-in register-tile kernel (HQQ_OPT_GEMM_REGTILE: weights dequantised straight into MFMA operands, only x through LDS) vs the
    default LDS-staged kernel: same exact weights, different fp32 summation order; ragged N and M exercise the masked edges"""
    for (M, N, K) in ((16384, 4096, 512), (8200, 4112, 256)):
        U, s, z = _random_layer(N, K, 64, nbits, seed=3)
        P = ops.pack(nbits, U.cuda())
        x = torch.randn(M, K, generator=torch.Generator().manual_seed(4)).half().cuda()
        bias = torch.randn(N, generator=torch.Generator().manual_seed(5)).half().cuda()
        base = ops.gemm(x, P, s.cuda(), z.cuda(), bias, N, K, 64, nbits)
        rt = ops.gemm(x, P, s.cuda(), z.cuda(), bias, N, K, 64, nbits, opts=ops.OPT_GEMM_REGTILE)
        torch.testing.assert_close(rt.float(), base.float(), rtol=1e-3, atol=2e-3)
        e = torch.zeros_like(x); e[torch.arange(M, device="cuda"), torch.arange(M, device="cuda") * 5 % K] = 1.0   # one-hot rows: exact columns
        rt1 = ops.gemm(e, P, s.cuda(), z.cuda(), None, N, K, 64, nbits, opts=ops.OPT_GEMM_REGTILE)
        assert torch.equal(rt1, ops.gemm(e, P, s.cuda(), z.cuda(), None, N, K, 64, nbits))


@pytest.mark.parametrize("nbits", [4, 2])
def test_forward_full_size_properties(ops, nbits):
    """Llama-2-7B shapes (BASELINE.json configs[1]): linearity and one-hot exactness, no oracle needed."""
    for (N, K) in [(4096, 4096), (11008, 4096), (4096, 11008)]:
        U, s, z = _random_layer(N, K, 64, nbits, seed=N ^ K)
        P = ops.pack(nbits, U.cuda())
        s, z = s.cuda(), z.cuda()
        g = torch.Generator().manual_seed(5)
        x1 = torch.randn(1, K, generator=g).half().cuda()
        x2 = torch.randn(1, K, generator=g).half().cuda()
        Wd = ops.dequantize(P, s.reshape(-1), z.reshape(-1), N, K, 64, nbits).float()
        for xx in (x1, x2):
            y = ops.forward(xx, P, s, z, None, N, K, 64, nbits).float()
            torch.testing.assert_close(y, xx.float() @ Wd.t(), rtol=1e-3, atol=1e-3)
        # GEMV (M=2) and GEMM (M=64) give the same rows as M=1
        X = torch.cat([x1, x2]).contiguous()
        y2 = ops.gemv(X, P, s, z, None, N, K, 64, nbits)
        assert torch.equal(y2[0], ops.gemv(x1, P, s, z, None, N, K, 64, nbits)[0])
        X64 = torch.randn(64, K, generator=g).half().cuda()
        torch.testing.assert_close(ops.gemm(X64, P, s, z, None, N, K, 64, nbits).float(), X64.float() @ Wd.t(), rtol=1e-3, atol=2e-3)


def test_forward_70b_shard_shapes_properties(ops):
    """Llama-2-70B shapes (BASELINE.json configs[4]) incl. the few-row / long-K per-rank shards that take the K-split path:
    one-hot exactness and agreement with the dequantise kernel, no oracle needed at this size"""
    for (N, K) in [(8192, 8192), (1024, 8192), (1024, 28672), (128, 8192), (3584, 8192)]:
        U, s, z = _random_layer(N, K, 64, 4, seed=N + K)
        P = ops.pack(4, U.cuda())
        s, z = s.cuda(), z.cuda()
        Wd = ops.dequantize(P, s.reshape(-1), z.reshape(-1), N, K, 64, 4)
        x = torch.randn(2, K, generator=torch.Generator().manual_seed(1)).half().cuda()
        y = ops.forward(x, P, s, z, None, N, K, 64, 4)
        torch.testing.assert_close(y.float(), x.float() @ Wd.float().t(), rtol=1e-3, atol=2e-3)
        assert torch.equal(ops.forward(x[:1], P, s, z, None, N, K, 64, 4)[0], y[0])          # a row's result does not depend on the batch
        e = torch.zeros(1, K, dtype=torch.float16, device="cuda"); e[0, K - 3] = 1.0
        assert torch.equal(ops.forward(e, P, s, z, None, N, K, 64, 4)[0], Wd[:, K - 3])


def test_forward_3bit_large_launches_properties(ops):
    """3-bit launches of >= 19 MB take the slab-sharing kernel by default (8192 x 8192 alone; gate|up of the 7B block as one
    grouped launch): agreement with the dequantise kernel, one-hot exactness, batch independence, grouped == single"""
    def make(N, K, seed):
        U, s, z = _random_layer(N, K, 64, 3, seed=seed)
        P = ops.pack(3, U.cuda())
        s, z = s.cuda(), z.cuda()
        return P, s, z, ops.dequantize(P, s.reshape(-1), z.reshape(-1), N, K, 64, 3)
    N, K = 8192, 8192
    P, s, z, Wd = make(N, K, 1)
    x = torch.randn(3, K, generator=torch.Generator().manual_seed(1)).half().cuda()
    y = ops.forward(x, P, s, z, None, N, K, 64, 3)
    torch.testing.assert_close(y.float(), x.float() @ Wd.float().t(), rtol=1e-3, atol=2e-3)
    assert torch.equal(ops.forward(x[:1], P, s, z, None, N, K, 64, 3)[0], y[0])
    e = torch.zeros(1, K, dtype=torch.float16, device="cuda"); e[0, K - 3] = 1.0
    assert torch.equal(ops.forward(e, P, s, z, None, N, K, 64, 3)[0], Wd[:, K - 3])
    del P, s, z, Wd
    N, K = 11008, 4096
    A, B = make(N, K, 2), make(N, K, 3)
    x = torch.randn(1, K, generator=torch.Generator().manual_seed(2)).half().cuda()
    ya, yb = ops.gemv_grouped(x, [(A[0], A[1], A[2], None, N), (B[0], B[1], B[2], None, N)], K, 64, 3)
    for (P, s, z, Wd), yg in ((A, ya), (B, yb)):
        torch.testing.assert_close(yg.float(), x.float() @ Wd.float().t(), rtol=1e-3, atol=2e-3)
        # alone this layer is 18 MB and takes the row-per-wave kernel: same exact weights, another summation order
        torch.testing.assert_close(ops.gemv(x, P, s, z, None, N, K, 64, 3).float(), yg.float(), rtol=2.0 ** -10, atol=1e-3)


def test_forward_3bit_long_k_rows_beyond_the_lds_budget_compose(ops):
    """the 3-bit decode kernels stage x in LDS: with K = 27648 (a 70B-sized down_proj shard) three rows no longer fit — `forward`
    must compose (HIP dequantise + library GEMM) instead of raising; two rows stay fused"""
    N, K = 40, 27648
    U, s, z = _random_layer(N, K, 64, 3, seed=77)
    P, s, z = ops.pack(3, U.cuda()), s.cuda(), z.cuda()
    Wd = ops.dequantize(P, s.reshape(-1), z.reshape(-1), N, K, 64, 3)
    assert ops.decode_covers(torch.float16, 2, N, K, 64, 3) and not ops.decode_covers(torch.float16, 3, N, K, 64, 3)
    for M in (2, 3, 4):
        x = torch.randn(M, K, generator=torch.Generator().manual_seed(M)).half().cuda()
        torch.testing.assert_close(ops.forward(x, P, s, z, None, N, K, 64, 3).float(), x.float() @ Wd.float().t(), rtol=2e-3, atol=4e-3)
    with pytest.raises(NotImplementedError):
        ops.gemv(torch.zeros(3, K, dtype=torch.float16, device="cuda"), P, s, z, None, N, K, 64, 3)   # the kernel itself says so


def test_forward_unsupported_is_loud(ops):
    # 3-bit with a group size the fused kernel does not cover: reported, never silently computed elsewhere
    x = torch.zeros(1, 128, dtype=torch.float16, device="cuda")
    W3 = torch.zeros(7, 128, dtype=torch.int32, device="cuda")
    s = torch.ones(64, 1, dtype=torch.float16, device="cuda")
    with pytest.raises(NotImplementedError):
        ops.gemv(x, W3, s, s, None, 64, 128, 128, 3)


# ------------------------------------------------------------------------------------------------
# Quantizer.quantize (solver + pack)
# ------------------------------------------------------------------------------------------------
def _check_quant(ops, W, nbits, gs, want_packed, want_scale, want_zero):
    """the HIP solver against what the reference's CPU (float32) path produced: every level, every zero-point bit, every scale bit"""
    Wq, s, z, info = ops.quantize(dev(W), nbits=nbits, group_size=gs, round_zero=(nbits == 4), return_info=True)
    torch.cuda.synchronize()
    # compare levels, not bytes, so a single differing level counts once
    got_u = ops.unpack(ops.PACK_BITS[nbits], Wq).cpu().numpy()[: W.size // gs]
    want_u = ops.unpack(ops.PACK_BITS[nbits], dev(want_packed)).cpu().numpy()[: W.size // gs]
    nbad = int((got_u != want_u).sum())
    zb, zw = z.cpu().numpy().reshape(-1).view(np.uint32), np.ascontiguousarray(want_zero, dtype=np.float32).reshape(-1).view(np.uint32)
    sb, sw = s.cpu().numpy().reshape(-1).view(np.uint32), np.ascontiguousarray(want_scale, dtype=np.float32).reshape(-1).view(np.uint32)
    nz, ns = int((zb != zw).sum()), int((sb != sw).sum())
    print(f"solver vs reference: {nbad} of {W.size} levels, {nz} of {zb.size} zero-points, {ns} scales differ; iterations {info.cpu().numpy().tolist()}")
    assert nbad == 0 and nz == 0 and ns == 0, f"{nbad} levels / {nz} zero-points / {ns} scales differ from the reference"
    assert np.array_equal(Wq.cpu().numpy(), want_packed)
    return nbad, info.cpu().numpy()


@pytest.mark.parametrize("name", QUANT_FILES)
def test_quantize_golden(ops, name):
    g = load_golden(name)
    nbits, gs = int(g["nbits"]), int(g["gs"])
    _check_quant(ops, g["W"], nbits, gs, g["Wq_packed"], g["scale_f32"], g["zero_f32"])


@pytest.mark.parametrize("nbits", [4, 3, 2])
def test_quantize_config1_1024(ops, nbits):
    import hashlib
    g = load_golden(f"cfg1_1024_{nbits}b")
    torch.manual_seed(0)
    W = torch.nn.Linear(1024, 1024, bias=False).weight.data.numpy()
    if hashlib.sha256(W.tobytes()).hexdigest().encode() != g["W_sha256"].tobytes():
        pytest.skip("torch RNG stream differs from the one the fixture was generated with")
    _check_quant(ops, W, nbits, 64, g["Wq_packed"], g["scale_f32"], g["zero_f32"])
    # end to end: quantise on the GPU, forward on the GPU, compare with the reference's CPU forward
    Wq, s, z = ops.quantize(dev(W), nbits=nbits, group_size=64, round_zero=(nbits == 4))
    if nbits in (4, 2):
        x = dev(g["x_f32"]).half()
        y = ops.forward(x, Wq, s.half(), z.half(), None, 1024, 1024, 64, nbits)
        torch.testing.assert_close(y.float().cpu(), torch.from_numpy(g["y_f16"].astype(np.float32)), rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("nbits", [4, 3, 2])
def test_quantize_vs_oracle_half_input(ops, oracle, nbits):
    # fp16 weights (the usual checkpoint dtype): `tensor.float()` first (quantize.py:102)
    W = (torch.randn(512, 1024, generator=torch.Generator().manual_seed(0)) * 0.02).half()
    o = oracle.quantize(W.float().numpy(), nbits=nbits, group_size=64)
    _check_quant(ops, W.numpy(), nbits, 64, oracle.pack(nbits, o["Wq"]), o["scale"], o["zero"])
    assert True


@pytest.mark.parametrize("nbits", [8, 6, 5, 1.58, 1])
def test_quantize_other_supported_bits(ops, oracle, nbits):
    """Quantizer.SUPPORTED_BITS beyond 4/3/2: 6- and 5-bit levels live in 8-bit containers, 1.58-bit (3 levels) in 2-bit ones
    (Quantizer.bit_to_packing, hqq/core/quantize.py:40-49); max_v = round(2^nbits - 1) (:121)"""
    W = (torch.randn(256, 512, generator=torch.Generator().manual_seed(int(nbits * 10))) * 0.02)
    o = oracle.quantize(W.numpy(), nbits=nbits, group_size=64)
    assert int(o["Wq"].max()) <= round(2 ** nbits - 1)
    container = ops.PACK_BITS[nbits]
    _check_quant(ops, W.numpy(), nbits, 64, oracle.pack(container, o["Wq"]), o["scale"], o["zero"])


@pytest.mark.parametrize("seed", range(12))
def test_quantize_random_shapes_vs_oracle(ops, oracle, seed):
    """solver + packing on random (N, K, group_size, nbits, weight scale, dtype): ragged group counts, tiny layers, outliers,
    constant groups (the `denom <= 1e-4` guard, quantize.py:128) — same levels / scale / zero as the CPU oracle"""
    import random
    rnd = random.Random(seed)
    nbits = rnd.choice([4, 4, 3, 2, 8, 1])
    gs = rnd.choice([64, 64, 32, 128, 16, 8])
    N, K = rnd.randint(1, 40) * 8, gs * rnd.randint(1, 24)
    g = torch.Generator().manual_seed(100 + seed)
    W = torch.randn(N, K, generator=g) * rnd.choice([0.02, 1.0, 1e-3, 30.0])
    if seed % 3 == 0:
        W[rnd.randrange(N), :gs] = 0.5                       # a constant group
        W[rnd.randrange(N), rnd.randrange(K)] = 1e3          # an outlier
    W = W.to(rnd.choice([torch.float32, torch.float16]))
    o = oracle.quantize(W.float().numpy(), nbits=nbits, group_size=gs)
    _check_quant(ops, W.numpy(), nbits, gs, oracle.pack(ops.PACK_BITS[nbits], o["Wq"]), o["scale"], o["zero"])


def test_quantize_full_size_properties(ops):
    """4096x4096 N(0,0.02^2): levels in range, dequant error sane, round trip through pack."""
    W = (torch.randn(4096, 4096, generator=torch.Generator().manual_seed(0)) * 0.02).half().cuda()
    for nbits in (4, 3, 2):
        Wq, s, z, info = ops.quantize(W, nbits=nbits, group_size=64, round_zero=(nbits == 4), return_info=True)
        U = ops.unpack(nbits, Wq)[: 4096 * 64]
        assert int(U.max()) <= 2 ** nbits - 1
        Wd = ops.dequantize(Wq, s.half().reshape(-1), z.half().reshape(-1), 4096, 4096, 64, nbits).float()
        err = (Wd - W.float()).abs().mean().item()
        assert err < {4: 0.0016, 3: 0.0035, 2: 0.008}[nbits], err
        assert 1 <= int(info[0]) <= 20
        # idempotence of packing: pack(unpack(Wq)) == Wq
        assert torch.equal(ops.pack(nbits, U), Wq)
        # no-optimize path = plain min/max rounding, and it must be worse or equal in L1 error
        Wq0, s0, z0 = ops.quantize(W, nbits=nbits, group_size=64, round_zero=(nbits == 4), optimize=False)
        Wd0 = ops.dequantize(Wq0, s0.half().reshape(-1), z0.half().reshape(-1), 4096, 4096, 64, nbits).float()
        assert (Wd0 - W.float()).abs().pow(0.7).mean() >= (Wd - W.float()).abs().pow(0.7).mean()
