"""GPU tests of the peer-memory exchange (csrc/exchange.hip, hqq_hip_exchange; hqq_amd.shard.PeerExchange): one kernel per exchange
point that stores a rank's slices straight into every rank's full rows in the reference's column order.  The expected rows are those
of hqq_amd.shard.shard_rows — the function tests/test_shard.py proves against whole layers."""
import os
import subprocess
import sys

import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "gpu tests need a ROCm device"
    from hqq_amd import ops as o
    assert o.is_available(), "libhqq_hip.so must load on the GPU box (no fallback)"
    return o


def _slices(full, N, nbits, world):
    from hqq_amd import shard
    return [full[:, shard.shard_rows(N, nbits, r, world).to(full.device)].contiguous() for r in range(world)]


@pytest.mark.parametrize("nbits,world,points", [
    (4, 2, [[512, 256, 256], [512]]),
    (4, 8, [[8192, 1024, 1024], [8192], [28672, 28672], [8192]]),      # a Llama-2-70B block over 8 ranks
    (2, 4, [[4096], [11008, 11008], [4096]]),
    (3, 8, [[4096, 4096], [4096]]),                                      # 3-bit shards: one run per rank
    (8, 3, [[3 * 40], [3 * 8, 3 * 24]]),                                 # runs that are not multiples of 16 bytes: the 2-byte copy loop
    (4, 16, [[2 * 16 * 8], [2 * 16 * 24]]),
])
@pytest.mark.parametrize("dtype", ["float16", "bfloat16"])
def test_every_rank_ends_up_with_the_reference_ordered_rows(ops, nbits, world, points, dtype):
    from hqq_amd.shard import PeerExchange
    dt = getattr(torch, dtype)
    # Ranks of ONE process cannot be relied on to be resident together (streams share hardware queues: a kernel would wait for one queued
    # behind it), so here the waits give up at once (spin_limit 1: the stores still happen) and the test clears flags and status itself.
    # The protocol proper — real waits, flags lowered and raised again over many rounds — is the two-process test below.
    group = PeerExchange.local_group(points, nbits, dt, "cuda", world, spin_limit=1)
    g = torch.Generator().manual_seed(7)
    for rnd in range(3):
        for e, pt in enumerate(points):
            fulls = [torch.randn(1, N, generator=g).to(dt).cuda() for N in pt]
            parts = [_slices(f, N, nbits, world) for f, N in zip(fulls, pt)]        # [layer][rank]
            for r in range(world):
                group[r].run(e, [parts[j][r] for j in range(len(pt))])
            torch.cuda.synchronize()
            for r in range(world):
                group[r]._arenas[r][:group[r]._status_off + 128].zero_()
                for j, f in enumerate(fulls):
                    assert torch.equal(group[r].full(e, j), f), f"round {rnd}, point {e}, layer {j}, rank {r}"


@pytest.mark.parametrize("nbits,world,points", [
    (4, 8, [[8192, 1024, 1024], [8192]]),
    (2, 4, [[4096], [11008, 11008]]),
    (3, 8, [[4096, 4096], [4096]]),
    (8, 3, [[3 * 40], [3 * 8, 3 * 24]]),          # runs that are not multiples of 16 bytes: the 2-byte copy loop, strided rows
])
@pytest.mark.parametrize("M", [2, 32, 64])
def test_a_decode_batch_lands_in_the_reference_order_without_an_unpermute(ops, nbits, world, points, M):
    """round 5 (ABI 6): M <= HQQ_EXCHANGE_MAX_ROWS activation rows per exchange — row m of a rank's slab run is stored at the same columns of every peer's
    row m (strided slab writes), so every rank ends with the full [M, N] outputs in the reference's column order; an arena built for 64 rows serves any
    smaller batch; a batch larger than the arena is refused"""
    from hqq_amd.shard import PeerExchange
    group = PeerExchange.local_group(points, nbits, torch.float16, "cuda", world, spin_limit=1, rows=64)
    g = torch.Generator().manual_seed(M)
    for e, pt in enumerate(points):
        fulls = [torch.randn(M, N, generator=g).half().cuda() for N in pt]
        parts = [_slices(f, N, nbits, world) for f, N in zip(fulls, pt)]        # [layer][rank]: [M, N / world] each, local order
        for r in range(world):
            group[r].run(e, [parts[j][r] for j in range(len(pt))])
        torch.cuda.synchronize()
        for r in range(world):
            group[r]._arenas[r][:group[r]._status_off + 128].zero_()
            for j, f in enumerate(fulls):
                assert torch.equal(group[r].full(e, j, rows=M), f), f"point {e}, layer {j}, rank {r}"
    small = PeerExchange.local_group(points, nbits, torch.float16, "cuda", world, spin_limit=1, rows=1)
    with pytest.raises(ValueError):
        small[0].run(0, [torch.zeros(2, N // world, dtype=torch.float16, device="cuda") for N in points[0]])


def test_the_reuse_rule_and_the_argument_checks(ops):
    from hqq_amd.shard import PeerExchange
    with pytest.raises(ValueError):
        PeerExchange.local_group([[512]], 4, torch.float16, "cuda", 2)             # one point only
    with pytest.raises(ValueError):
        PeerExchange.local_group([[510], [512]], 4, torch.float16, "cuda", 4)      # does not split
    grp = PeerExchange.local_group([[512], [512]], 4, torch.float16, "cuda", 1)
    y = torch.zeros(1, 512, device="cuda", dtype=torch.float16)
    grp[0].run(0, [y])
    with pytest.raises(RuntimeError):
        grp[0].run(0, [y])                                                           # the same point twice in a row
    with pytest.raises(ValueError):
        grp[0].run(1, [y, y])                                                        # wrong number of layers


def test_a_rank_that_never_arrives_is_reported_not_waited_for(ops):
    from hqq_amd.shard import PeerExchange
    grp = PeerExchange.local_group([[512], [512]], 4, torch.float16, "cuda", 2, spin_limit=256)
    y = torch.ones(1, 256, device="cuda", dtype=torch.float16)
    grp[0].run(0, [y])                       # rank 1 never runs its exchange
    torch.cuda.synchronize()
    assert grp[0].status() == 2              # 1 + the missing rank


_WORKER = r"""
import os, sys, torch
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from hqq_amd import shard
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)                      # every rank on GPU 0: the peers' arenas are mapped through IPC handles all the same
dist.init_process_group(backend="gloo")
points, nbits = [[8192, 1024, 1024], [8192], [4096, 4096], [8192]], 4
px = shard.PeerExchange(points, nbits, torch.float16, "cuda:0")
g = torch.Generator().manual_seed(11)         # the same stream of random rows on every rank
bad = 0
for rnd in range(25):
    for e, pt in enumerate(points):
        fulls = [torch.randn(1, N, generator=g).half().cuda() for N in pt]
        mine = [f[:, shard.shard_rows(N, nbits, rank, world).cuda()].contiguous() for f, N in zip(fulls, pt)]
        px.run(e, mine)
        torch.cuda.synchronize()
        bad += sum(int(not torch.equal(px.full(e, j), f)) for j, f in enumerate(fulls))
# under hipGraph replay (what bench.py --gpus N does): the four points captured once with fixed slice buffers, refilled between replays
bufs = [[torch.empty(1, N // world, device="cuda", dtype=torch.float16) for N in pt] for pt in points]
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for e in range(len(points)):
        px.run(e, bufs[e])
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
dist.barrier()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, capture_error_mode="thread_local"):
    for e in range(len(points)):
        px.run(e, bufs[e])
for rnd in range(20):
    fulls = [[torch.randn(1, N, generator=g).half().cuda() for N in pt] for pt in points]
    for e, pt in enumerate(points):
        for j, N in enumerate(pt):
            bufs[e][j].copy_(fulls[e][j][:, shard.shard_rows(N, nbits, rank, world).cuda()])
    graph.replay()
    torch.cuda.synchronize()
    bad += sum(int(not torch.equal(px.full(e, j), fulls[e][j])) for e, pt in enumerate(points) for j in range(len(pt)))
    dist.barrier()   # (a rank must not refill its slices... they are private; the barrier only keeps the ranks' rounds aligned for the compare)

# the product path: two column-sharded layers whose decode outputs are exchanged by the kernel (hqq_amd.shard.ShardedHQQForward, peer=...)
from hqq_amd import ops
gw = torch.Generator().manual_seed(5)
px2 = shard.PeerExchange([[1024], [512]], nbits, torch.float16, "cuda:0")
layers = []
for e, (N, K) in enumerate(((1024, 512), (512, 1024))):
    W = (torch.randn(N, K, generator=gw) * 0.05).half().cuda()
    Wq, s, z = ops.quantize(W, nbits=nbits, group_size=64, round_zero=True)
    s, z = s.half(), z.half()
    layers.append((shard.ShardedHQQForward(Wq, s, z, None, N, K, 64, nbits, peer=(px2, e)), ops.dequantize(Wq, s.reshape(-1), z.reshape(-1), N, K, 64, nbits)))
for rnd in range(10):
    for fwd, Wd in layers:
        x = torch.randn(1, Wd.shape[1], generator=g).half().cuda()
        y = fwd(x)
        torch.cuda.synchronize()
        if not torch.allclose(y.float(), x.float() @ Wd.float().t(), rtol=1e-3, atol=2e-3):
            bad += 1
st = px.status() + px2.status()
dist.barrier()
print(f"rank {rank}: mismatches {bad}, status {st}", flush=True)
sys.exit(0 if (bad == 0 and st == 0) else 3)
"""


def test_two_processes_exchange_through_ipc_mapped_arenas(ops, tmp_path):
    """two ranks = two processes (gloo for the handle exchange, both on GPU 0): the arenas are reached through hipIpc handles exactly as
    on a multi-GPU node, only the stores do not leave the device"""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29613", WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
            o += "\n[timeout]"
        outs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n----\n".join(outs)
