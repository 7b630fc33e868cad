"""The N > 1 path of bench.py — what tools/node_first_run.sh launches on a multi-GPU node — as a 2-rank dry run on ONE GPU: both ranks launch their
shard's kernels on GPU 0, gloo carries the collectives, the peer arenas are mapped through IPC handles.  Plumbing, not a measurement: the point is
that the first node-hour is not spent on a typo.  Nothing here has run over xGMI."""
import json
import os
import socket
import subprocess
import sys

import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(mode, bs, plan="sharded"):
    env = dict(os.environ, HQQ_BENCH_ONE_GPU="1", HQQ_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    if mode == "auto":
        env.pop("HQQ_BENCH_EXCHANGE", None)
    else:
        env["HQQ_BENCH_EXCHANGE"] = mode
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--bs", str(bs), "--blocks", "2", "--steps", "3", "--warmup", "1", "--random-codes", "--no-single-gpu-reference", "--plan", plan]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().split("\n") if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly ONE JSON line"
    return json.loads(lines[0]), r.stderr


@pytest.mark.parametrize("mode,bs", [("auto", 1), ("rows1", 1), ("peer", 1), ("peer", 32), ("gather", 32)])
def test_two_rank_bench_dry_run_on_one_gpu(mode, bs):
    assert torch.cuda.is_available()
    d, err = _run(mode, bs)
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "strong"
    assert d["ms_per_step"] > 0 and d["value"] > 0 and "roofline" in d
    x = d["exchange"]
    assert x["ms_per_step"] > 0
    if mode == "peer":   # the peer-memory kernel ran (it validates itself against the collective at start-up) and its bounded waits never gave up
        assert x["exchange_kernels_per_step"] > 0 and x["peer_status"] == 0, (x, err[-800:])
        assert x["collective_launches_per_step"] == 0 and x["unpermute_kernels_per_step"] == 0
    else:
        assert x["collective_launches_per_step"] > 0
        if bs == 1 and mode == "rows1":   # one activation row: per-slab gathers straight into the reference's column order, no un-permute (gloo: issued one by one)
            assert x["unpermute_kernels_per_step"] == 0


@pytest.mark.parametrize("mode,bs", [("auto", 1), ("peer", 1), ("gather", 32)])
def test_two_rank_bench_dry_run_with_the_adaptive_plan(mode, bs):
    """--plan adaptive (hqq_amd.shard.plan_exchange_groups): at two ranks the 70B block's q|k|v (42 MB) and o (34 MB) are below the replicate threshold
    (62 MB) -> held whole by both ranks, no exchange behind them; gate|up and down stay sharded: 2 exchange points per block instead of 4."""
    from hqq_amd import shard
    d, err = _run(mode, bs, plan="adaptive")
    pl = d["plan"]
    assert pl["name"] == "adaptive"
    assert pl["groups"] == {"q|k|v": "replicated-small", "o": "replicated-small", "gate|up": "sharded", "down": "sharded"}, pl
    assert pl["exchange_points_per_block"] == 2
    assert 41e6 < shard.replicate_below_bytes(2) < 80e6
    x = d["exchange"]
    assert x["points_per_step"] == 2 * 2 and x["us_per_point"] > 0
    if mode == "peer":
        assert x["exchange_kernels_per_step"] == 4 and x["peer_status"] == 0, (x, err[-800:])
    # `value` counts a replicated layer once: the job's bytes are below world x one rank's bytes
    assert pl["job_bytes_per_step"] < 2 * d["config"]["bytes_per_step_per_gpu"]
    assert abs(d["value"] - pl["job_bytes_per_step"] / (d["ms_per_step"] * 1e-3) / 1e9) < 0.01 * d["value"]
