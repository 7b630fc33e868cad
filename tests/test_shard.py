"""Column shard + all-gather (hqq_amd/shard.py): index math on CPU, and the N > 1 path with world_size-2 gloo."""
import os
import socket

import pytest

torch = pytest.importorskip("torch")

from hqq_amd import shard  # noqa: E402

PER = {8: 1, 4: 2, 2: 4, 1: 8}


def _pack_np(nbits, U):   # BitPack layout restated with torch on CPU (test-local helper)
    per = PER[nbits]
    step = U.shape[0] // per
    out = torch.zeros((step, U.shape[1]), dtype=torch.uint8)
    for s in range(per):
        out |= (U[s * step:(s + 1) * step] << (nbits * (per - 1 - s))).to(torch.uint8)
    return out


@pytest.mark.parametrize("nbits", [4, 2, 8])
@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_shard_rows_partition_and_unpermute(nbits, world):
    N = 64
    rows = [shard.shard_rows(N, nbits, r, world) for r in range(world)]
    assert sorted(torch.cat(rows).tolist()) == list(range(N))          # every output row owned exactly once
    # gather order = rank-major, each rank in local order; unpermute must restore 0..N-1
    M = 3
    y = torch.stack([torch.stack([rows[r].float() + 1000 * m for m in range(M)]) for r in range(world)])   # [P, M, N/P]
    back = shard.unpermute(y, N, nbits, world)
    assert torch.equal(back, torch.stack([torch.arange(N).float() + 1000 * m for m in range(M)]))


@pytest.mark.parametrize("nbits", [4, 2])
def test_packed_slice_is_a_self_contained_layer(nbits, monkeypatch):
    """the zero-copy packed-row block of rank r unpacks to exactly the rows shard_rows() names"""
    N, K, gs, world = 32, 128, 64, 2
    G = K // gs
    g = torch.Generator().manual_seed(0)
    U = torch.randint(0, 2 ** nbits, (N * G, gs), generator=g, dtype=torch.uint8)      # [R, gs], row = n*G + k//gs
    Wq = _pack_np(nbits, U)
    scale = torch.rand(N * G, 1, generator=g)
    zero = torch.rand(N * G, 1, generator=g)
    for r in range(world):
        Wl, sc, ze, b, n_loc = shard.shard_packed(Wq, scale, zero, None, N, K, gs, nbits, r, world)
        assert Wl.data_ptr() == Wq[r * Wq.shape[0] // world].data_ptr()               # a view, not a copy
        rows = shard.shard_rows(N, nbits, r, world)
        want_U = U.reshape(N, G, gs)[rows].reshape(n_loc * G, gs)
        assert torch.equal(_pack_np(nbits, want_U), Wl)
        assert torch.equal(sc, scale.reshape(N, G)[rows].reshape(-1, 1)) and torch.equal(ze, zero.reshape(N, G)[rows].reshape(-1, 1))
    with pytest.raises(ValueError):
        shard.shard_rows(36, 4, 0, 8)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nbits, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        N, K, gs, M = 32, 128, 64, 3
        G = K // gs
        g = torch.Generator().manual_seed(0)                                     # same layer on every rank
        U = torch.randint(0, 2 ** nbits, (N * G, gs), generator=g, dtype=torch.uint8)
        Wq = _pack_np(nbits, U)
        scale = torch.rand(N * G, 1, generator=g) * 0.01
        zero = torch.rand(N * G, 1, generator=g) * 8
        bias = torch.rand(N, generator=g)
        x = torch.randn(M, K, generator=g)
        Wfull = ((U.float() - zero) * scale).reshape(N, K)
        rows = shard.shard_rows(N, nbits, rank, world)
        # stand-in for the HIP kernel on CPU: dense math on exactly this rank's rows (the GPU tests cover the kernel itself)
        local = lambda xx: xx @ Wfull[rows].t() + bias[rows]                      # noqa: E731
        sh = shard.ShardedHQQForward(Wq, scale, zero, bias, N, K, gs, nbits, local_forward=local)
        y = sh(x)
        ok = torch.allclose(y, x @ Wfull.t() + bias, atol=1e-5) and sh.n_loc == N // world and tuple(y.shape) == (M, N)
        # one activation row: per-slab gathers straight into the reference's column order (shard.gather_columns), no un-permute
        y1 = sh(x[:1])
        ok = ok and tuple(y1.shape) == (1, N) and torch.allclose(y1, x[:1] @ Wfull.t() + bias, atol=1e-5)
        # a long prompt: chunks of OVERLAP_ROWS rows, each chunk's gather issued asynchronously behind its local GEMM (ragged last chunk)
        sh.OVERLAP_ROWS = 5
        xl = torch.randn(23, K, generator=g)
        yl = sh(xl)
        ok = ok and tuple(yl.shape) == (23, N) and torch.allclose(yl, xl @ Wfull.t() + bias, atol=1e-5)
        yl3 = sh(xl.reshape(1, 23, K))
        ok = ok and tuple(yl3.shape) == (1, 23, N) and torch.equal(yl3[0], yl)
        # a layer the plan replicates (exchange group too small to shard): every rank computes it whole, no collective is issued
        rep = shard.ShardedHQQForward(Wq, scale, zero, bias, N, K, gs, nbits, local_forward=lambda xx: xx @ Wfull.t() + bias, replicate=True)
        calls = []
        orig = dist.all_gather_into_tensor
        dist.all_gather_into_tensor = lambda *a_, **k_: (calls.append(1), orig(*a_, **k_))[1]
        try:
            yr = rep(x)
        finally:
            dist.all_gather_into_tensor = orig
        ok = ok and not calls and rep.n_loc == N and torch.allclose(yr, x @ Wfull.t() + bias, atol=1e-5)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("nbits", [4, 2])
def test_sharded_forward_all_gather_gloo_world2(nbits):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, nbits, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(2))
    assert res == [(0, True), (1, True)]


def test_peer_exchange_layout_is_the_same_on_every_rank_and_rows_do_not_overlap():
    """the arena layout is pure arithmetic on (points, world): every rank must derive the same offsets (peers address each other's rows
    as base + offset), rows and flag lines must not overlap"""
    from hqq_amd.shard import PeerExchange
    points = [[8192, 1024, 1024], [8192], [28672, 28672], [8192]]
    total = PeerExchange._layout_bytes(points, 8)
    assert total % PeerExchange.ALIGN == 0 and total >= 128 * 5 + 2 * sum(sum(p) for p in points)
    # replay the layout the constructor uses and check it against the total
    off, spans = 128 * len(points) + 128, []
    for pt in points:
        for n in pt:
            off = (off + 255) // 256 * 256
            spans.append((off, off + 2 * n))
            off += 2 * n
    assert (off + 255) // 256 * 256 == total
    assert all(a1 <= b0 for (a0, a1), (b0, b1) in zip(spans, spans[1:])) and spans[0][0] >= 128 * 5


def test_peer_exchange_host_logic_without_a_gpu():
    """argument checks, the arena views and the re-use rule are host logic: rows are views of the rank's own arena at the layout's
    offsets; the kernel call itself refuses CPU tensors (no fallback)"""
    import pytest
    from hqq_amd.shard import PeerExchange
    with pytest.raises(ValueError, match="at least two exchange points"):
        PeerExchange.local_group([[512]], 4, torch.float16, "cpu", 2)
    with pytest.raises(ValueError, match="cannot be split"):
        PeerExchange.local_group([[510], [512]], 4, torch.float16, "cpu", 4)
    with pytest.raises(ValueError, match="1..4 layers"):
        PeerExchange.local_group([[512] * 5, [512]], 4, torch.float16, "cpu", 2)
    points = [[1024, 256], [512]]
    grp = PeerExchange.local_group(points, 4, torch.bfloat16, "cpu", 2)
    assert [g.rank for g in grp] == [0, 1] and all(g.world == 2 for g in grp)
    a0 = grp[0]._arenas[0]
    assert grp[0].full(0, 1).shape == (1, 256) and grp[0].full(0, 1).dtype == torch.bfloat16
    assert grp[0].full(0, 0).data_ptr() == a0.data_ptr() + grp[0]._row_off[0][0] and grp[0]._row_off[0][0] % 256 == 0
    assert grp[1].full(1, 0).data_ptr() == grp[1]._arenas[1].data_ptr() + grp[1]._row_off[1][0]
    # every rank addresses rank 1's rows the same way: base of arena 1 + the shared offsets
    assert grp[0]._full_ptrs[1][1] == grp[1]._full_ptrs[1][1] == [grp[0]._arenas[1].data_ptr() + o for o in grp[0]._row_off[1]]
    y = [torch.zeros(1, 512, dtype=torch.bfloat16), torch.zeros(1, 128, dtype=torch.bfloat16)]
    with pytest.raises(ValueError, match="holds 2 layers"):
        grp[0].run(0, y[:1])
    with pytest.raises(RuntimeError, match="need tensors on the GPU"):
        grp[0].run(1, [torch.zeros(1, 256, dtype=torch.bfloat16)])
    assert grp[0]._last is None                      # a call that was not enqueued does not count as the last exchange
    grp[0]._last = 1                                 # (as after an enqueued exchange of point 1)
    with pytest.raises(RuntimeError, match="alternate between at least two points"):
        grp[0].run(1, [torch.zeros(1, 256, dtype=torch.bfloat16)])


def test_exchange_plan_replicates_small_groups_only():
    """plan_exchange_groups: pure arithmetic on (packed bytes per exchange group, world) — the same plan on every rank; one rank has nothing to plan;
    at 8 ranks the 7B block's q|k|v / o / down fall under the launch-model threshold (35 MB) and gate|up does not; of the 70B block only o (33.5 MB, the
    borderline case: 3.8 us more streaming against one exchange) does"""
    b7 = [3 * 4096 * 4096 // 2, 4096 * 4096 // 2, 2 * 11008 * 4096 // 2, 4096 * 11008 // 2]
    b70 = [(8192 + 2 * 1024) * 8192 // 2, 8192 * 8192 // 2, 2 * 28672 * 8192 // 2, 8192 * 28672 // 2]
    assert shard.plan_exchange_groups(b7, 1) == ["sharded"] * 4
    assert shard.plan_exchange_groups(b7, 8) == ["replicated-small", "replicated-small", "sharded", "replicated-small"]
    assert shard.plan_exchange_groups(b70, 8) == ["sharded", "replicated-small", "sharded", "sharded"]
    assert shard.plan_exchange_groups(b70, 8, threshold=0) == ["sharded"] * 4
    assert shard.plan_exchange_groups(b70, 8, threshold=1 << 40) == ["replicated-small"] * 4
    thr2, thr8 = shard.replicate_below_bytes(2), shard.replicate_below_bytes(8)
    assert thr2 > thr8 > 0 and shard.replicate_below_bytes(1) == 0   # the fewer ranks, the less a shard takes off a launch
