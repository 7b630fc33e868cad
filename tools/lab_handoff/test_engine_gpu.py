"""(lab) the persistent decode engine's tests — cut out of tests/test_round2_gpu.py when engine.hip left the library (round 4); they need a lab build that links engine.hip and the DecodePlan wrapper of tools/lab_handoff/ops_handoff.py"""
# ------------------------------------------------------------------------------------------------
# the persistent decode engine
# ------------------------------------------------------------------------------------------------
def _chain(ops, nbits, dims, grid, sub, bias):
    torch.manual_seed(1)
    x0 = torch.randn(1, dims[0][0], device="cuda").half()
    layers, stages = [], []
    x = x0
    for si, (K, Ns) in enumerate(dims):
        Ls = []
        for j, N in enumerate(Ns):
            Wq, s, z = _qlayer(ops, N, K, nbits, seed=100 * si + j, std=1.0 / K ** 0.5)
            b = (torch.randn(N, device="cuda") * 0.1).half() if bias else None
            y = torch.full((1, N), float("nan"), device="cuda", dtype=torch.float16)
            Ls.append((Wq, s, z, b, N, y))
        stages.append((x, Ls))
        layers.append(Ls)
        x = Ls[-1][5]   # the next stage READS what this one writes
    allsc = all(ops.meta_scalable(L[1], L[2], L[4], K, 64, nbits) for (K, _), Ls in zip(dims, layers) for L in Ls)
    plan = ops.DecodePlan(stages, nbits, opts=ops.OPT_META_SCALABLE if (sub and allsc) else 0, grid=grid)
    for rep in range(3):   # re-runnable: the sync words are cleared by every run
        plan.run()
    torch.cuda.synchronize()
    assert plan.status() == 0
    xr = x0
    for (K, Ns), Ls in zip(dims, layers):
        outs = ops.gemv_grouped(xr, [(L[0], L[1], L[2], L[3], L[4]) for L in Ls], K, 64, nbits, opts=0)
        for L, o in zip(Ls, outs):
            torch.testing.assert_close(L[5].float(), o.float(), rtol=1e-3, atol=1e-3)
            Wd = ops.dequantize(L[0], L[1].reshape(-1), L[2].reshape(-1), L[4], K, 64, nbits)
            ref = xr.float() @ Wd.float().t() + (0 if L[3] is None else L[3].float())
            torch.testing.assert_close(L[5].float(), ref, rtol=1e-3, atol=2e-3)
        xr = outs[-1]


@pytest.mark.parametrize("sub", [False, True])
@pytest.mark.parametrize("case", [
    (4, [(1024, [512, 1024]), (1024, [2048]), (2048, [1024])], 8, False),
    (4, [(1024, [512, 1024]), (1024, [2048]), (2048, [1024])], 0, False),          # one workgroup per CU: most own no row at all
    (4, [(1280, [64, 34, 1152]), (1152, [640]), (640, [128, 128, 128, 256])], 5, True),   # ragged K (not a multiple of 1024), 4 layers, bias
    (2, [(1024, [512, 1024]), (1024, [2048]), (2048, [1024])], 16, False),
    (8, [(1024, [512, 1024]), (1024, [256])], 0, True),
    (4, [(4096, [4096, 4096, 4096]), (4096, [4096]), (4096, [11008, 11008]), (11008, [4096])], 0, False),   # one Llama-2-7B block, chained
])
def test_decode_engine_chain_with_real_dependencies(ops, case, sub):
    """every stage reads the buffer the previous stage wrote (a wrong or late hand-off shows up as NaN or stale values);
    compared with the per-launch kernels and with dequantise + fp32 matmul"""
    nbits, dims, grid, bias = case
    _chain(ops, nbits, dims, grid, sub, bias)


def test_decode_engine_is_reproducible_and_graph_capturable(ops):
    nbits = 4
    dims = [(4096, [4096, 4096, 4096]), (4096, [4096]), (4096, [11008, 11008]), (11008, [4096])] * 3
    torch.manual_seed(0)
    x0 = torch.randn(1, 4096, device="cuda").half()
    stages, x = [], x0
    keep = []
    for si, (K, Ns) in enumerate(dims):
        Ls = []
        for j, N in enumerate(Ns):
            Wq, s, z = _qlayer(ops, N, K, nbits, seed=si * 10 + j, std=1.0 / K ** 0.5)
            Ls.append((Wq, s, z, None, N, torch.zeros(1, N, device="cuda", dtype=torch.float16)))
        stages.append((x, Ls))
        keep.append(Ls)
        x = Ls[0][5]
    plan = ops.DecodePlan(stages, nbits, opts=ops.OPT_META_SCALABLE)
    plan.run(); torch.cuda.synchronize()
    first = [L[5].clone() for Ls in keep for L in Ls]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        plan.run()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    assert plan.status() == 0
    for a, b in zip(first, [L[5] for Ls in keep for L in Ls]):
        assert torch.equal(a, b)
    assert not any(torch.isnan(t).any() for t in first)


def test_decode_engine_reports_what_it_does_not_cover(ops):
    Wq, s, z = _qlayer(ops, 64, 256, 4, seed=0)
    x = torch.zeros(1, 256, device="cuda", dtype=torch.float16)
    y = torch.zeros(1, 64, device="cuda", dtype=torch.float16)
    with pytest.raises(NotImplementedError):
        ops.DecodePlan([(x, [(Wq, s, z, None, 64, y)])], 3)          # 3-bit containers
    with pytest.raises(NotImplementedError):
        ops.DecodePlan([(x, [(Wq, s, z, None, 64, y)])], 4, group_size=128)
    with pytest.raises(ValueError):
        ops.DecodePlan([(torch.zeros(2, 256, device="cuda", dtype=torch.float16), [(Wq, s, z, None, 64, y)])], 4)   # one activation row per stage


