"""GPU parity tests of the no-K-split batched-decode kernel (hqq_amd/csrc/batch.hip, 5..64 activation rows): through the C ABI against the
CPU oracle, forced with OPT_BATCH_NEW so that small test layers reach it (by default only launches of at least half a chip of units do).

Bar: a one-hot activation row reads a column of the dequantised matrix BIT FOR BIT (the weights multiplied are the reference's,
hqq/core/quantize.py:198); forward within rtol = atol = 1e-3 (fp16) / one bf16 ulp of the double-accumulated oracle; a row's bits do not
depend on the batch (5 rows vs 64: one token half vs two), on the units per workgroup or on what else was in the grouped launch.
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


def _layer(N, K, nbits, seed, dt):
    g = torch.Generator().manual_seed(seed)
    R = N * K // 64
    U = torch.randint(0, 2 ** nbits, (R, 64), generator=g, dtype=torch.uint8)
    s = (torch.rand(R, 1, generator=g) * 0.004 + 0.001).to(dt)
    z = (torch.rand(R, 1, generator=g) * (2 ** nbits - 1)).to(dt)
    z.view(-1)[::5] = 0.00836          # zero-points far below one level: q - z must still round once
    z.view(-1)[1::11] = 2.0 ** -12
    return U, s, z


def _want(oracle, x, Wd, bias, code):
    if code == 2:
        yo, _ = oracle.matmul(raw16(x), Wd, None if bias is None else raw16(bias), 2)
        return torch.from_numpy(yo.view(np.int16).copy()).view(torch.bfloat16).float()
    yo, _ = oracle.matmul(x.numpy(), Wd, None if bias is None else bias.numpy(), 1)
    return torch.from_numpy(yo.astype(np.float32))


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("nbits", [4, 2])
@pytest.mark.parametrize("M", [5, 16, 17, 32, 33, 48, 64])
@pytest.mark.parametrize("NK", [(512, 1024), (4096 + 16, 512), (208, 2048 + 768), (64, 11008)])
def test_batch_kernel_vs_oracle(ops, oracle, dt, nbits, M, NK):
    N, K = NK
    code = 2 if dt == torch.bfloat16 else 1
    U, s, z = _layer(N, K, nbits, seed=N + K + nbits + M, dt=dt)
    P = oracle.pack(nbits, U.numpy())
    Wd = oracle.dequantize(nbits, P, raw16(s) if code == 2 else s.numpy(), raw16(z) if code == 2 else z.numpy(), N, K, 64, code)
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(1)).to(dt)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(2)).to(dt) if M % 2 else None
    want = _want(oracle, x, Wd, bias, code)
    args = (dev(P), s.cuda(), z.cuda(), None if bias is None else bias.cuda(), N, K, 64, nbits)
    y = ops.gemv(x.cuda(), *args, opts=ops.OPT_BATCH_NEW)
    if code == 2:
        torch.testing.assert_close(y.float().cpu(), want, rtol=2.0 ** -7, atol=2e-3)
    else:
        torch.testing.assert_close(y.float().cpu(), want, rtol=1e-3, atol=1e-3)
    # reproducible; a row does not depend on the batch it is computed in (5 rows: one token half, one token tile)
    assert torch.equal(y, ops.gemv(x.cuda(), *args, opts=ops.OPT_BATCH_NEW))
    assert torch.equal(y[:5], ops.gemv(x[:5].cuda(), *args, opts=ops.OPT_BATCH_NEW))
    # the split-K kernel it replaces sums in another association: same weights, results within the forward tolerance of each other
    yo = ops.gemv(x.cuda(), *args, opts=ops.OPT_BATCH_OLD)
    torch.testing.assert_close(y.float(), yo.float(), rtol=2.0 ** -7 if code == 2 else 1e-3, atol=2e-3)
    # one-hot probes: y[m, n] = W[n, k] bit for bit — every position of a 16-k lane chunk, the ends of the row, every k-block class
    Wdev = ops.dequantize(args[0], args[1].reshape(-1), args[2].reshape(-1), N, K, 64, nbits)
    ks = sorted(set([(64 * (5 * i + 1) + i) % K for i in range(16)] + [0, K - 1, K // 2 + 5, K - 64, 63]))[:M]
    e = torch.zeros(M, K, dtype=dt, device="cuda")
    for r, k in enumerate(ks): e[r, k] = 1.0
    ye = ops.gemv(e, *args[:3], None, N, K, 64, nbits, opts=ops.OPT_BATCH_NEW)
    for r, k in enumerate(ks):
        assert torch.equal(ye[r], Wdev[:, k]), f"column {k}"
    assert torch.count_nonzero(ye[len(ks):]) == 0


@pytest.mark.parametrize("nbits", [4, 2])
def test_batch_kernel_full_size_layers(ops, oracle, nbits):
    """the shapes it is for: 4096 x 4096 (one unit per workgroup at 4 bits) and a grouped q|k|v (three), 32 and 64 rows, against the oracle"""
    K = 4096
    Ls, Wds = [], []
    for i in range(3):
        U, s, z = _layer(4096, K, nbits, seed=50 + i, dt=torch.float16)
        P = oracle.pack(nbits, U.numpy())
        Wds.append(oracle.dequantize(nbits, P, s.numpy(), z.numpy(), 4096, K, 64, 1))
        Ls.append((dev(P), s.cuda(), z.cuda(), None, 4096))
    for M in (32, 64):
        x = torch.randn(M, K, generator=torch.Generator().manual_seed(M)).half()
        outs = ops.gemv_grouped(x.cuda(), Ls, K, 64, nbits)          # default routing: a launch this size takes the new kernel
        for (W, s, z, _, N), Wd, y in zip(Ls, Wds, outs):
            torch.testing.assert_close(y.float().cpu(), _want(oracle, x, Wd, None, 1), rtol=1e-3, atol=1e-3)
            assert torch.equal(y, ops.gemv(x.cuda(), W, s, z, None, N, K, 64, nbits, opts=ops.OPT_BATCH_NEW))   # grouped or alone: the same bits


def test_batch_kernel_grouped_layers_and_graph_capture(ops, oracle):
    K, nbits = 1024, 4
    Ls = []
    for i, N in enumerate((512, 128, 272)):
        U, s, z = _layer(N, K, nbits, seed=90 + i, dt=torch.float16)
        bias = torch.randn(N, generator=torch.Generator().manual_seed(i)).half().cuda() if i == 1 else None
        Ls.append((dev(oracle.pack(nbits, U.numpy())), s.cuda(), z.cuda(), bias, N))
    x = torch.randn(24, K, generator=torch.Generator().manual_seed(3)).half().cuda()
    outs = ops.gemv_grouped(x, Ls, K, 64, nbits, opts=ops.OPT_BATCH_NEW)
    for (W, s, z, b, N), y in zip(Ls, outs):
        assert torch.equal(y, ops.gemv(x, W, s, z, b, N, K, 64, nbits, opts=ops.OPT_BATCH_NEW))
    g = torch.cuda.CUDAGraph()
    bufs = [torch.empty_like(o) for o in outs]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.gemv_grouped(x, Ls, K, 64, nbits, outs=bufs, opts=ops.OPT_BATCH_NEW)
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(g):
        ops.gemv_grouped(x, Ls, K, 64, nbits, outs=bufs, opts=ops.OPT_BATCH_NEW)
    for b in bufs: b.zero_()
    g.replay()
    torch.cuda.synchronize()
    for o, b in zip(outs, bufs):
        assert torch.equal(o, b)
