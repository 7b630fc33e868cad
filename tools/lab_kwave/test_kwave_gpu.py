"""LAB (run with HQQ_AMD_LIB=tools/libhqq_hip_kwave.so python -m pytest tools/lab_kwave/test_kwave_gpu.py; needs tests/conftest.py's `oracle` fixture:
`-p conftest --rootdir tests` or copy it next to this file).  GPU parity tests of the batched-decode kernel WITHOUT a cross-workgroup K split (hqq_amd/csrc/kwave.hip, round 5): 5..64 activation
rows, fp16 / bf16, 8- / 4- / 3-bit (stream layout) / 2-bit, group_size 64 — through the C ABI, against the CPU oracle (oracle/hqq_oracle.c).

What the reference computes there: HQQLinear.forward = torch.matmul(x, dequantize().t()) (+ bias) — hqq/core/quantize.py:880-898.
Bar: forward within rtol = atol = 1e-3 (fp16; bf16: one bf16 ulp) of the double-accumulated oracle on reference-exact weights; a one-hot
activation row reads a column of the dequantised matrix BIT FOR BIT; and the properties the design promises by construction:
  * the number of 16-row tiles a workgroup owns (a host speed choice) never changes a bit,
  * a row of y does not depend on the batch it was computed in (every M of 5..64 takes the same kernel, the same K slices per wave),
  * the three-op rebuild (HQQ_OPT_META_SCALABLE) gives the four-op one's bits,
  * no workspace: capturable without any reservation.
The split-K kernel it replaces (skinny.hip, HQQ_OPT_BATCH_SPLITK) stays tested beside it: same oracle, same tolerance.
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


def _layer(N, K, nbits, seed, dt, sub_friendly):
    g = torch.Generator().manual_seed(seed)
    R = N * K // 64
    U = torch.randint(0, 2 ** nbits, (R, 64), generator=g, dtype=torch.uint8)
    s = (torch.rand(R, 1, generator=g) * 0.004 + 0.001).to(dt)
    z = (torch.rand(R, 1, generator=g) * (2 ** nbits - 1)).to(dt)
    if sub_friendly:
        z = z.float().clamp_min(0.0625).to(dt)   # lowest bit of every zero-point >= 2^-15: z 2^-9 exact in fp16
    else:   # zero-points far below one level: q - z must still round once
        z.view(-1)[::5] = 0.00836
        z.view(-1)[1::11] = 2.0 ** -12
    return U, s, z


def _prepare(ops, oracle, N, K, nbits, dt, seed, sub_friendly=False):
    """-> (device args tuple for ops.gemv, base option bits, dequantised matrix on the device, oracle weights, code)"""
    code = 2 if dt == torch.bfloat16 else 1
    U, s, z = _layer(N, K, nbits, seed, dt, sub_friendly)
    P = oracle.pack(nbits, U.numpy())
    Wd = oracle.dequantize(nbits, P, raw16(s) if code == 2 else s.numpy(), raw16(z) if code == 2 else z.numpy(), N, K, 64, code)
    sd, zd = s.cuda(), z.cuda()
    Wdev = ops.dequantize(dev(P), sd.reshape(-1), zd.reshape(-1), N, K, 64, nbits)
    if nbits == 3:
        Wq, base = ops.w3s_pack(dev(P), N, K), ops.OPT_W3S
        if code == 1 and sub_friendly and ops.w3s_meta_scalable(sd, zd, N, K):
            base |= ops.OPT_META_SCALABLE
    else:
        Wq, base = dev(P), 0
        if code == 1 and sub_friendly and ops.meta_scalable(sd, zd, N, K, 64, nbits):
            base |= ops.OPT_META_SCALABLE
    return (Wq, sd, zd), base, Wdev, Wd, code


def _want(oracle, x, Wd, bias, code):
    if code == 2:
        yo, _ = oracle.matmul(raw16(x), Wd, None if bias is None else raw16(bias), 2)
        return torch.from_numpy(yo.view(np.int16).copy()).view(torch.bfloat16).float()
    yo, _ = oracle.matmul(x.numpy(), Wd, None if bias is None else bias.numpy(), 1)
    return torch.from_numpy(yo.astype(np.float32))


def _close(y, want, code, nbits=4):
    if code == 2:
        torch.testing.assert_close(y.float().cpu(), want, rtol=2.0 ** -7, atol=2e-3)
    else:
        # the stated tolerance, plus one fp16 ulp of the reference value because both sides are stored rounded to fp16 (tests/test_hip_parity.py)
        y = y.float().cpu()
        ulp = torch.pow(2.0, torch.floor(torch.log2(want.abs().clamp_min(2.0 ** -14))) - 10)
        bad = (y - want).abs() > (16 if nbits == 8 else 1) * 1e-3 + 1e-3 * want.abs() + ulp   # (8-bit levels: weights 16x larger, as tests/test_hip_parity.py allows)
        assert not bool(bad.any()), f"{int(bad.sum())} of {bad.numel()} outside tolerance"


SHAPES = [(512, 1024), (208, 2048 + 768), (72, 11008), (344, 1152), (4096 + 16, 1280)]   # (344 / 72: ragged last tiles; 2816, 1152, 1280, 11008: uneven K slices per wave; 1152: K % 256 != 0)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("nbits", [8, 4, 3, 2])
@pytest.mark.parametrize("NK", SHAPES)
def test_kwave_vs_oracle_every_tile_count_same_bits(ops, oracle, dt, nbits, NK):
    N, K = NK
    args, base, Wdev, Wd, code = _prepare(ops, oracle, N, K, nbits, dt, seed=N + K + nbits)
    for M in (5, 16, 17, 32, 33, 48, 64):
        x = torch.randn(M, K, generator=torch.Generator().manual_seed(M)).to(dt)
        bias = torch.randn(N, generator=torch.Generator().manual_seed(2)).to(dt) if M % 2 else None
        full = args + (None if bias is None else bias.cuda(), N, K, 64, nbits)
        want = _want(oracle, x, Wd, bias, code)
        y = ops.gemv(x.cuda(), *full, opts=base)
        _close(y, want, code, nbits)
        if K % 256 == 0:   # (forward() routes the other K elsewhere)
            assert torch.equal(y, ops.forward(x.cuda(), *full, opts=base))
        # the split-K kernel it replaces: same oracle, same tolerance (K % 256 == 0 only)
        if K % 256 == 0:
            _close(ops.gemv(x.cuda(), *full, opts=base | ops.OPT_BATCH_SPLITK), want, code, nbits)
        # tiles per workgroup: a speed choice, never a bit
        for rt in (1, 2, 3, 4, 6):
            assert torch.equal(y, ops.gemv(x.cuda(), *full, opts=base | ops.OPT_SKINNY_KS(rt))), f"M={M} rt={rt}"
        # a row does not depend on the batch it is computed in
        assert torch.equal(y[:5], ops.gemv(x[:5].cuda(), *full, opts=base))
        if M >= 33:
            assert torch.equal(y[:20], ops.gemv(x[:20].cuda(), *full, opts=base))


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("nbits", [8, 4, 3, 2])
def test_kwave_one_hot_rows_read_the_dequantised_matrix(ops, oracle, dt, nbits):
    """y[m, :] = W[:, k] bit for bit for one-hot rows: every k residue of a step (128 k), first / last k, rows in every m-tile"""
    N, K = 208, 2048 + 768
    args, base, Wdev, _, _ = _prepare(ops, oracle, N, K, nbits, dt, seed=5 + nbits)
    M = 64
    ks = [0, K - 1, (3 * K) // 7] + [128 * (i % 22) + (i * 37) % 128 for i in range(61)]
    e = torch.zeros(M, K, dtype=dt, device="cuda")
    for r, k in enumerate(ks): e[r, k] = 1.0
    for rt in (0, 1, 3):
        ye = ops.gemv(e, *args, None, N, K, 64, nbits, opts=base | ops.OPT_SKINNY_KS(rt))
        for r, k in enumerate(ks):
            assert torch.equal(ye[r], Wdev[:, k]), f"column {k} (row {r}, rt {rt})"


@pytest.mark.parametrize("nbits", [8, 4, 3, 2])
def test_kwave_three_op_rebuild_same_bits(ops, oracle, nbits):
    N, K = 344, 2048
    args, base, Wdev, Wd, code = _prepare(ops, oracle, N, K, nbits, torch.float16, seed=40 + nbits, sub_friendly=True)
    assert base & ops.OPT_META_SCALABLE, "the fixture's group constants should pass the meta check"
    for M in (7, 32, 64):
        x = torch.randn(M, K, generator=torch.Generator().manual_seed(M)).half()
        y3 = ops.gemv(x.cuda(), *args, None, N, K, 64, nbits, opts=base)
        y4 = ops.gemv(x.cuda(), *args, None, N, K, 64, nbits, opts=base & ~ops.OPT_META_SCALABLE)
        assert torch.equal(y3, y4)
        _close(y3, _want(oracle, x, Wd, None, code), code, nbits)


@pytest.mark.parametrize("nbits", [4, 3, 2])
def test_kwave_grouped_equals_single_and_captures_without_workspace(ops, oracle, nbits):
    K, M = 1024, 24
    layers, singles = [], []
    base = ops.OPT_W3S if nbits == 3 else 0
    for i, N in enumerate([512, 96, 40, 1024]):
        args, _, _, _, _ = _prepare(ops, oracle, N, K, nbits, torch.float16, seed=300 + i)
        b = torch.randn(N, generator=torch.Generator().manual_seed(i)).half().cuda() if i % 2 else None
        layers.append(args + (b, N))
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(9)).half().cuda()
    assert ops._workspace(x, nbits, [L[4] for L in layers], M, K, 64, base) == (None, 0)
    ys = ops.gemv_grouped(x, layers, K, 64, nbits, opts=base)
    for (Wq, s, z, b, N), y in zip(layers, ys):
        assert torch.equal(y, ops.gemv(x, Wq, s, z, b, N, K, 64, nbits, opts=base))
    Wq, s, z, b, N = layers[3]
    out = torch.empty(M, N, dtype=torch.float16, device="cuda")
    want = ops.gemv(x, Wq, s, z, b, N, K, 64, nbits, out=out, opts=base).clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
            ops.gemv(x, Wq, s, z, b, N, K, 64, nbits, out=out, opts=base)
    out.zero_()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want)


def test_kwave_full_size_7b_shapes_vs_oracle_rows(ops, oracle):
    """configs[1]'s shapes at 32 rows: a sample of output rows against the oracle, every column one-hot exact on a k sample"""
    for N, K in ((4096, 4096), (11008, 4096), (4096, 11008)):
        g = torch.Generator().manual_seed(N + K)
        R = N * K // 64
        U = torch.randint(0, 16, (R, 64), generator=g, dtype=torch.uint8)
        s = (torch.rand(R, 1, generator=g) * 0.004 + 0.001).half()
        z = (torch.rand(R, 1, generator=g) * 15).half()
        Wq, sd, zd = ops.pack(4, U.cuda()), s.cuda(), z.cuda()
        Wdev = ops.dequantize(Wq, sd.reshape(-1), zd.reshape(-1), N, K, 64, 4)
        x = torch.randn(32, K, generator=g).half()
        y = ops.gemv(x.cuda(), Wq, sd, zd, None, N, K, 64, 4)
        want = (x.double() @ Wdev.cpu().double().t()).float()          # reference-exact weights (bit-identical to the oracle's: test_dequantize_*), double accumulation
        torch.testing.assert_close(y.float().cpu(), want, rtol=1e-3, atol=1e-3)
        assert torch.equal(y[:7], ops.gemv(x[:7].cuda(), Wq, sd, zd, None, N, K, 64, 4))
        e = torch.zeros(8, K, dtype=torch.float16, device="cuda")
        ks = [0, K - 1, K // 2 + 5, 127, 128, 1023, 1024, K - 129]
        for r, k in enumerate(ks): e[r, k] = 1.0
        ye = ops.gemv(e, Wq, sd, zd, None, N, K, 64, 4)
        for r, k in enumerate(ks): assert torch.equal(ye[r], Wdev[:, k])
