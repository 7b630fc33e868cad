"""GPU tests of the chained decode launches (csrc/gemv_chain.hip, hqq_hip_gemv_chained; round 3): a decode step as overlapped launches on
two streams, each waiting IN the kernel for its predecessor's outputs.  Every stage reads the buffer the previous stage wrote, so a
wrong, late or stale hand-off shows up as NaN or as a value that differs from the stream-ordered launches — and the two must agree
bit for bit (same arithmetic, same summation order)."""
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "gpu tests need a ROCm device"
    from hqq_amd import ops as o
    assert o.is_available(), "libhqq_hip.so must load on the GPU box (no fallback)"
    return o


def _qlayer(ops, N, K, nbits, seed, std):
    W = (torch.randn(N, K, generator=torch.Generator().manual_seed(seed)) * std).half().cuda()
    Wq, s, z = ops.quantize(W, nbits=nbits, group_size=64, round_zero=(nbits == 4))
    return Wq, s.half(), z.half()


def _build(ops, nbits, dims, M, bias, sub, seed=0):
    """stages whose x IS the previous stage's last output buffer; returns (x0, stages, per-stage opts)"""
    torch.manual_seed(seed + 1)
    x0 = torch.randn(M, dims[0][0], device="cuda").half()
    stages, opts, x = [], [], x0
    for si, (K, Ns) in enumerate(dims):
        Ls = []
        for j, N in enumerate(Ns):
            Wq, s, z = _qlayer(ops, N, K, nbits, seed=seed + 100 * si + j, std=1.0 / K ** 0.5)
            b = (torch.randn(N, device="cuda") * 0.1).half() if bias else None
            Ls.append((Wq, s, z, b, N, torch.full((M, N), float("nan"), device="cuda", dtype=torch.float16)))
        stages.append((x, Ls))
        ok = sub and all(ops.meta_scalable(L[1], L[2], L[4], K, 64, nbits) for L in Ls)
        opts.append(ops.OPT_META_SCALABLE if ok else 0)
        x = Ls[-1][5]
    return x0, stages, opts


def _reference(ops, nbits, x0, stages, opts):
    """the same stages as plain stream-ordered launches (hqq_hip_gemv_grouped) into fresh buffers"""
    outs_all, x = [], x0
    for (_, Ls), o in zip(stages, opts):
        K = x.shape[-1]
        outs = ops.gemv_grouped(x, [L[:5] for L in Ls], K, 64, nbits, opts=o)
        outs_all.append(outs)
        x = outs[-1]
    return outs_all


def _poison(stages):
    for _, Ls in stages:
        for L in Ls:
            L[5].fill_(float("nan"))


CASES = [
    (4, [(1024, [512, 1024]), (1024, [2048]), (2048, [1024])], 1, False),
    (4, [(1280, [64, 34, 1152]), (1152, [640]), (640, [128, 128, 128, 256])], 1, True),      # ragged K, 4 layers, bias, tiny layers (waves without a row)
    (4, [(1024, [512, 1024]), (1024, [2048]), (2048, [1024]), (1024, [64])], 3, True),        # three activation rows
    (2, [(1024, [512, 1024]), (1024, [2048]), (2048, [1024])], 1, False),
    (2, [(1024, [1024]), (1024, [1024])], 4, False),
    (8, [(1024, [512, 1024]), (1024, [256]), (256, [1024])], 2, True),
    (4, [(4096, [4096, 4096, 4096]), (4096, [4096]), (4096, [11008, 11008]), (11008, [4096])] * 2, 1, False),   # two Llama-2-7B blocks, chained
]


@pytest.mark.parametrize("sub", [False, True])
@pytest.mark.parametrize("case", CASES)
def test_chained_launches_equal_stream_ordered_launches(ops, case, sub):
    nbits, dims, M, bias = case
    x0, stages, opts = _build(ops, nbits, dims, M, bias, sub)
    want = _reference(ops, nbits, x0, stages, opts)
    chain = ops.LaunchChain(stages, nbits, opts=opts)
    for rep in range(3):   # re-runnable: counters and status are cleared by every run; stale outputs of the previous run are poisoned
        _poison(stages)
        chain.run()
        torch.cuda.synchronize()
        assert chain.status() == 0, "an in-kernel wait gave up"
        for (_, Ls), outs in zip(stages, want):
            for L, o in zip(Ls, outs):
                assert not torch.isnan(L[5]).any()
                assert torch.equal(L[5], o), "a chained launch must give the bits of the stream-ordered launch"


def test_chained_launches_against_dequantise_and_fp32_matmul(ops):
    nbits, dims = 4, [(4096, [4096, 4096, 4096]), (4096, [4096]), (4096, [11008, 11008]), (11008, [4096])]
    x0, stages, opts = _build(ops, nbits, dims, 1, False, True, seed=7)
    chain = ops.LaunchChain(stages, nbits, opts=opts)
    chain.run()
    torch.cuda.synchronize()
    assert chain.status() == 0
    for x, Ls in stages:
        for L in Ls:
            Wd = ops.dequantize(L[0], L[1].reshape(-1), L[2].reshape(-1), L[4], x.shape[-1], 64, nbits)
            torch.testing.assert_close(L[5].float(), x.float() @ Wd.float().t(), rtol=1e-3, atol=2e-3)


def test_chain_is_graph_capturable_and_reproducible_under_replay(ops):
    """40 replays of a 24-stage chain (six 7B blocks' shapes): every replay must leave exactly the bits of the first eager run —
    a consumer that read x early, or a stale line of the previous replay, changes them"""
    nbits = 4
    dims = [(4096, [4096, 4096, 4096]), (4096, [4096]), (4096, [11008, 11008]), (11008, [4096])] * 6
    x0, stages, opts = _build(ops, nbits, dims, 1, False, True, seed=3)
    want = _reference(ops, nbits, x0, stages, opts)
    chain = ops.LaunchChain(stages, nbits, opts=opts)
    chain.run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        chain.run()
    for rep in range(40):
        if rep % 8 == 0:
            _poison(stages)
        g.replay()
    torch.cuda.synchronize()
    assert chain.status() == 0
    for (_, Ls), outs in zip(stages, want):
        for L, o in zip(Ls, outs):
            assert torch.equal(L[5], o)


def test_chain_under_uneven_load(ops):
    """the hand-off with another stream hammering HBM next to it (uneven load is where a missing drain or acquire shows)"""
    nbits = 4
    dims = [(4096, [4096]), (4096, [4096]), (4096, [11008, 11008]), (11008, [4096])] * 3
    x0, stages, opts = _build(ops, nbits, dims, 1, False, True, seed=5)
    want = _reference(ops, nbits, x0, stages, opts)
    chain = ops.LaunchChain(stages, nbits, opts=opts)
    big = torch.empty(64 << 20, device="cuda", dtype=torch.float32)
    noise = torch.cuda.Stream()
    for rep in range(10):
        _poison(stages)
        torch.cuda.synchronize()
        with torch.cuda.stream(noise):
            for _ in range(4):
                big.add_(1.0)
        chain.run()
        torch.cuda.synchronize()
        assert chain.status() == 0
        for (_, Ls), outs in zip(stages, want):
            for L, o in zip(Ls, outs):
                assert torch.equal(L[5], o)


def test_a_wait_that_nobody_answers_gives_up_and_says_so(ops):
    """bounded spins: a link waiting for arrivals that never come runs on after spin_limit polls and sets the status word"""
    Wq, s, z = _qlayer(ops, 256, 512, 4, seed=0, std=0.05)
    x = torch.randn(1, 512, device="cuda").half()
    y = torch.zeros(1, 256, device="cuda", dtype=torch.float16)
    sync = torch.zeros(ops.CHAIN_COUNTER_BYTES // 4 + 32, dtype=torch.int32, device="cuda")
    ops.gemv_chained(x, [(Wq, s, z, None, 256)], 512, 64, 4, [y], 0, wait=sync.data_ptr(), wait_arrivals=7, signal=None,
                     status=sync.data_ptr() + ops.CHAIN_COUNTER_BYTES, spin_limit=64)
    torch.cuda.synchronize()
    assert int(sync[ops.CHAIN_COUNTER_BYTES // 4].item()) != 0


def test_chain_reports_what_it_does_not_cover(ops):
    Wq, s, z = _qlayer(ops, 64, 256, 4, seed=0, std=0.05)
    x = torch.zeros(1, 256, device="cuda", dtype=torch.float16)
    y = torch.zeros(1, 64, device="cuda", dtype=torch.float16)
    with pytest.raises(NotImplementedError):
        ops.gemv_chained(x, [(Wq, s, z, None, 64)], 256, 64, 3, [y])          # 3-bit containers
    with pytest.raises(NotImplementedError):
        ops.gemv_chained(x, [(Wq, s, z, None, 64)], 256, 128, 4, [y])         # group size
    with pytest.raises(NotImplementedError):
        ops.gemv_chained(x, [(Wq, s, z, None, 64)], 256, 64, 4, [y], opts=ops.OPT_FACTORED)
