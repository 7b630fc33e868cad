#!/usr/bin/env python3
"""bench.py — the contract benchmark of the HQQ forward hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload auto|decode|decode70b|prefill] [--nbits 4|2|3] [--bs M]

Headline (N = 1; BASELINE.json configs[1], the configuration the metric is quoted on):
  one *step* = one token (bs=1) through every quantised linear of a Llama-2-7B decoder stack —
  32 blocks x {q,k,v,o 4096x4096; gate,up 11008x4096; down 4096x11008}, nbits=4 group_size=64 axis=1,
  128 dependent fused unpack->dequantize->GEMV launches (q|k|v, o, gate|up, down grouped) replayed from one
  hipGraph, streaming 3.65 GB of packed weights + meta from HBM (far beyond the 256 MiB Infinity Cache, so
  every step is HBM traffic).  Weights are synthetic N(0, 0.02^2) fp16 tensors quantised on the GPU by the
  HIP half-quadratic solver before timing.  Arithmetic: exact weights (round16(round16(q - z) * s), the
  reference's), in the three-op form wherever hqq_hip_meta_check allows it.
  `legs` adds the other shapes the metric names (bs=32; one 4096x4096 layer at bs=1 / bs=32), the int3 / int2 stacks, bs=128 and the
  prefill block; `quantize` times the other hot path; `end_to_end` the fused decode loop; `cpu_baseline` times the reference's per-call arithmetic
  (unpack -> (W_r - zero) * scale -> matmul, hqq/core/bitpack.py:31-38, quantize.py:183-199, :880-882) restated
  in torch eager on the host cores of this box.
  `--workload prefill` runs configs[2]: M tokens (default 8192 = 4 x 2048) through one block's seven linears.

Multi-GPU (launched by torch.distributed.run, one rank per GPU, RCCL).  Default for N > 1 (`auto`) is BASELINE
configs[4]: the Llama-2-70B linear shapes (q,o 8192x8192; k,v 1024x8192; gate,up 28672x8192; down 8192x28672),
nbits=4, every layer's output columns sharded over the N ranks (packed-row blocks of the reference layout,
hqq_amd/shard.py) — STRONG scaling of fixed layers: x is replicated, every exchange point (after q|k|v, o,
gate|up, down) completes the outputs in the reference's column order inside the timed region, and the exchange is
also timed on its own.  At one activation row the exchange is per-slab RCCL all-gathers in one coalesced launch (straight into
the reference's column order) or, where the backend cannot coalesce, the shard-wide all-gather + un-permute; HQQ_BENCH_EXCHANGE=peer
opts into one small kernel storing the rank's slices into every rank's full rows over peer memory (csrc/exchange.hip; fine-grained
arenas mapped through IPC handles, validated against the collective at start-up; a wait that gives up fails the run) — opt-in until
it has run over xGMI on a real node; rows1 / gather force a collective form; the JSON's `exchange` block says which ran.
`--workload decode --gpus N` keeps round 1's weak-scaling variant (every rank streams a 7B-stack-sized shard).

Prints ONE JSON line on rank 0.  `value` = algorithmic GB/s streamed by the whole job (SURVEY.md §8d bytes:
W_q + scale + zero + x + y); tok/s is reported next to it.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_PEAK_TFLOPS = 2500.0  # fp16/bf16 dense

LLAMA2_7B_BLOCK = [("q", 4096, 4096), ("k", 4096, 4096), ("v", 4096, 4096), ("o", 4096, 4096),
                   ("gate", 11008, 4096), ("up", 11008, 4096), ("down", 4096, 11008)]
LLAMA2_70B_BLOCK = [("q", 8192, 8192), ("k", 1024, 8192), ("v", 1024, 8192), ("o", 8192, 8192),
                    ("gate", 28672, 8192), ("up", 28672, 8192), ("down", 8192, 28672)]
# exchange points of a column-sharded block: outputs that are consumed together are gathered together
EXCHANGE_GROUPS = [("q", "k", "v"), ("o",), ("gate", "up"), ("down",)]
N_BLOCKS_7B, N_BLOCKS_70B = 32, 80


def wq_bytes(N, K, nbits, gs=64):
    R = N * K // gs
    if nbits == 3:
        return 4 * gs * ((R + 9) // 10)
    return N * K * nbits // 8


def gemv_bytes(N, K, nbits, M=1, gs=64):
    """algorithmic bytes of one fused forward call, fp16 meta"""
    return wq_bytes(N, K, nbits, gs) + 2 * 2 * (N * K // gs) + 2 * K * M + 2 * N * M


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="auto", choices=["auto", "decode", "decode70b", "prefill"],
                    help="auto: the 7B stack on one GPU, the column-sharded 70B stack (configs[4]) on several")
    ap.add_argument("--nbits", type=int, default=4)
    ap.add_argument("--bs", type=int, default=1, help="decode batch (rows of x), 1..64 (int4/int2/int8 from 5 up, fp16 or bf16; 1..4 for int3)")
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16"], help="compute dtype (bf16: decode only)")
    ap.add_argument("--prefill-tokens", type=int, default=8192)
    ap.add_argument("--blocks", type=int, default=0, help="decoder blocks (default: 32 for the 7B stack, 80 for the 70B one)")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--no-group", action="store_true", help="one launch per layer instead of one per exchange group (q/k/v, o, gate/up, down)")
    ap.add_argument("--plan", default=os.environ.get("HQQ_BENCH_PLAN", "sharded"), choices=["sharded", "adaptive"],
                    help="N > 1, 70B strong scaling: sharded = every exchange group column-sharded (4 exchanges per block); adaptive = "
                         "hqq_amd.shard.plan_exchange_groups: a group too small to shard is held and computed WHOLE by every rank, no exchange behind it")
    ap.add_argument("--no-single-gpu-reference", action="store_true", help="N > 1, 70B strong scaling: skip timing the unsharded stack on rank 0 alone")
    ap.add_argument("--prefill-route", default="auto", choices=["auto", "fused", "dense", "library"],
                    help="prefill: auto = ops.forward's own choice per layer (the product path); fused = the fused MFMA dequant-GEMM (gemm_pipe.hip); "
                         "dense = dequantise kernel + the in-tree MFMA GEMM (gemm_dense.hip); library = dequantise kernel + hipBLASLt (comparison only)")
    ap.add_argument("--library-gemm", action="store_true", help="prefill: same as --prefill-route library")
    ap.add_argument("--shapes", default="auto", choices=["auto", "7b", "70b"], help="prefill: the block's layer shapes (auto = Llama-2-7B; 70b = BASELINE configs[4]'s shapes on one GPU)")
    ap.add_argument("--prefill-chunk", type=int, default=0, help="prefill: tokens per forward call (0 = all of them in one call: the weights are rebuilt once)")
    ap.add_argument("--streams", type=int, default=1, help="study mode: deal the launches over this many parallel graph branches (ignores the decoder's dependency chain)")
    ap.add_argument("--no-legs", action="store_true", help="skip the extra legs (bs=32, single layer, int3 / int2, prefill, quantise, end to end)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--random-codes", action="store_true", help="skip the solver: random packed codes + meta (faster setup)")
    ap.add_argument("--gemv-mode", default="exact", choices=["exact", "exact4", "factored"],
                    help="exact: reference-identical weights, three-op rebuild where the meta allows it (default); exact4: always the "
                         "general four-op rebuild; factored: fp32 affine map factored out of the dot product (not the reference's weights)")
    return ap.parse_args()


class Layer:
    __slots__ = ("name", "N", "K", "Wq", "scale", "zero", "opts")


def make_layer(ops, name, N, K, nbits, dev, seed, random_codes, cd=torch.float16, rows=None):
    """one synthetic quantised layer; rows = (first, count) keeps only that packed-row block of it (a column shard, hqq_amd/shard.py)"""
    L = Layer()
    L.name, L.N, L.K = name, N, K
    g = torch.Generator(device=dev).manual_seed(seed)
    if random_codes:
        R = N * K // 64
        prow = (R + 9) // 10 if nbits == 3 else R * nbits // 8
        if nbits == 3:
            L.Wq = torch.randint(0, 2 ** 30, (prow, 64), dtype=torch.int32, device=dev, generator=g)
        else:
            L.Wq = torch.randint(0, 256, (prow, 64), dtype=torch.uint8, device=dev, generator=g)
        L.scale = (torch.rand(R, 1, device=dev, generator=g) * 0.004 + 0.001).to(cd)
        L.zero = (torch.rand(R, 1, device=dev, generator=g) * (2 ** nbits - 1)).to(cd)
    else:
        W = (torch.randn(N, K, device=dev, generator=g) * 0.02).half()
        Wq, s, z = ops.quantize(W, nbits=nbits, group_size=64, round_zero=(nbits == 4))
        # HQQLinear.cuda(): meta is cast to compute_dtype (quantize.py:515-583)
        L.Wq, L.scale, L.zero = Wq, s.to(cd), z.to(cd)
    if nbits == 3 and ops.w3s_covers(N, K, 64):
        # what HQQLinearHIP holds for a 3-bit layer: the levels re-laid out ONCE, at patch time, into the stream layout (csrc/w3s.h) — the
        # reference's optimised backends re-lay out at patch time too (hqq/backends/torchao.py:202-241, marlin.py:74-123); scale / zero unchanged
        L.Wq = ops.w3s_pack(L.Wq, N, K)
        L.opts = ops.OPT_W3S | (ops.OPT_META_SCALABLE if (cd == torch.float16 and ops.w3s_meta_scalable(L.scale, L.zero, N, K)) else 0)
        return L
    L.opts = ops.OPT_META_SCALABLE if (cd == torch.float16 and nbits in (8, 4, 3, 2) and ops.meta_scalable(L.scale, L.zero, N, K, 64, nbits)) else 0
    return L


def _timed(run, steps, warmup, dist=None, dev=None):
    """K timed steps bracketed by barrier + synchronize; returns (wall s/step, HIP-event s/step), MAX over ranks"""
    for _ in range(warmup):
        run()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        run()
    ev1.record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ev_elapsed = ev0.elapsed_time(ev1) * 1e-3
    if dist is not None:
        t = torch.tensor([elapsed, ev_elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, ev_elapsed = float(t[0]), float(t[1])
    return elapsed / steps, ev_elapsed / steps


def _graphed(step, use_graph, rank=0):
    """run `step` once eagerly on a side stream (lazy loads, workspace growth, communicator set-up), then capture it"""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    if not use_graph:
        return step, False
    try:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            step()
        graph.replay()
        torch.cuda.synchronize()
        return graph.replay, True
    except Exception as e:   # capture unsupported for some op: run eagerly, say so
        if rank == 0:
            print(f"[bench] graph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
        torch.cuda.synchronize()
        return step, False


def reference_forward_cpu(Wq, scale, zero, x, nbits, N, K, gs=64):
    """HQQBackend.PYTORCH's forward restated with the reference's own torch ops on CPU tensors (8 / 4 / 2 bit, axis 1): BitPack.unpack_*
    (hqq/core/bitpack.py:14-22, 31-38, 53-64: every slab shifted out of the packed bytes into rows [s step, (s + 1) step) of a compute-dtype
    buffer), Quantizer.dequantize ((W_r - zero) * scale, quantize.py:183-199), torch.matmul(x, W.t()) (quantize.py:880-882).
    The `cpu_baseline` leg times this; tests/test_host_api.py pins it to the fixtures the imported reference wrote (tests/golden/cfg1_*)."""
    per = 8 // nbits
    mask = (1 << nbits) - 1
    step = Wq.shape[0]
    tmp = torch.empty([per * step, gs], dtype=scale.dtype)
    for s_ in range(per):
        tmp[s_ * step:(s_ + 1) * step] = (Wq >> (nbits * (per - 1 - s_))) & mask
    W = ((tmp - zero) * scale).reshape(N, K)
    return torch.matmul(x, W.t())


def cpu_baseline(nbits):
    """HQQBackend.PYTORCH's per-call arithmetic on this box's host cores, bounded to ~10 s each:
    (1) restated in torch eager with the reference's own ops — BitPack.unpack (bitpack.py:31-38 / :53-64 / :14-22), Quantizer.dequantize
        ((W_r - zero) * scale, quantize.py:183-199), torch.matmul(x, W.t()) (quantize.py:880-882), fp16, all host threads;
    (2) the C oracle (oracle/hqq_oracle.c, OpenMP, double accumulation) as a second figure."""
    import numpy as np
    N = K = 4096
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    nb = gemv_bytes(N, K, nbits)
    stack_calls = sum(gemv_bytes(n, k, nbits) for _, n, k in LLAMA2_7B_BLOCK) * N_BLOCKS_7B / nb
    out = {"unit": "GB/s", "cores": cores, "threads": cores, "host_cores": cores, "kind": "port"}
    try:   # (name the CPU the baseline ran on)
        with open("/proc/cpuinfo") as f:
            out["cpu_model"] = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), None)
    except OSError:
        out["cpu_model"] = None
    # ---- (1) torch eager ----
    if nbits in (8, 4, 2):
        g = torch.Generator().manual_seed(0)
        per = 8 // nbits
        R = N * K // 64
        Wq = torch.randint(0, 256, (R // per, 64), dtype=torch.uint8, generator=g)
        scale = (torch.rand(R, 1, generator=g) * 0.004 + 0.001).half()
        zero = (torch.rand(R, 1, generator=g) * (2 ** nbits - 1)).half()
        x = torch.randn(1, K, generator=g).half()

        def fwd():
            return reference_forward_cpu(Wq, scale, zero, x, nbits, N, K)

        # torch's fp16 CPU kernels do not scale to every core of a big host: try a few thread counts (2 s each) and keep the best
        best = None
        with torch.no_grad():
            for th in sorted({min(cores, 8), min(cores, 32), min(cores, 64), cores}):
                torch.set_num_threads(th)
                fwd()
                t0 = time.perf_counter()
                reps = 0
                while True:
                    fwd()
                    reps += 1
                    el = time.perf_counter() - t0
                    if el > 2.5 or reps >= 200:
                        break
                if best is None or el / reps < best[0]:
                    best = (el / reps, th, reps, el)
        t, th, reps, el = best
        out["cores"] = out["threads"] = th   # the threads actually used (the fastest of the counts tried); host_cores = what the box has
        out.update({"value": round(nb / t / 1e9, 4), "ms_per_layer_call": round(t * 1e3, 3),
                    "tok_s_7b_stack_equiv": round(1.0 / (t * stack_calls), 4),
                    "sample": f"torch {torch.__version__} CPU eager restatement of HQQBackend.PYTORCH's forward (unpack -> (W_r - zero) * scale -> matmul, fp16) on one "
                              f"4096x4096 int{nbits} gs=64 layer, bs=1: {reps} calls in {el:.1f} s with torch.set_num_threads({th}) — the fastest of 8/32/64/{cores} "
                              f"threads on this {cores}-core host"})
    # ---- (2) C oracle ----
    try:
        from oracle import hqq_oracle as orc
        rng = np.random.default_rng(0)
        R = N * K // 64
        U = rng.integers(0, 2 ** nbits, size=(R, 64), dtype=np.uint8)
        P = orc.pack(nbits, U)
        s = orc.to_cd((rng.random((R, 1), dtype=np.float32) * 0.004 + 0.001), orc.F16)
        z = orc.to_cd((rng.random((R, 1), dtype=np.float32) * (2 ** nbits - 1)), orc.F16)
        xo = orc.to_cd(rng.standard_normal((1, K), dtype=np.float32), orc.F16)
        orc.forward(nbits, P, s, z, None, xo, N, K, 64, orc.F16)
        t0 = time.perf_counter()
        reps = 0
        while True:
            orc.forward(nbits, P, s, z, None, xo, N, K, 64, orc.F16)
            reps += 1
            el = time.perf_counter() - t0
            if el > 6.0 or reps >= 200:
                break
        t = el / reps
        c = {"value": round(nb / t / 1e9, 4), "ms_per_layer_call": round(t * 1e3, 3),
             "sample": f"oracle/hqq_oracle.c forward (unpack+dequantize+matmul, OpenMP, double accumulation), same layer: {reps} calls in {el:.1f} s"}
        if "value" in out:
            out["c_port"] = c
        else:
            out.update(c)
            out["tok_s_7b_stack_equiv"] = round(1.0 / (t * stack_calls), 4)
    except Exception as e:   # the oracle is optional here (it is the checker, not the product)
        out["c_port"] = {"error": repr(e)}
    # ---- (3) the other hot path: Quantizer.quantize (solver + final levels) through the C oracle, bounded sample ----
    try:
        from oracle import hqq_oracle as orc
        rng = np.random.default_rng(1)
        Wn = (rng.standard_normal((512, 4096), dtype=np.float32) * 0.02)   # 1/8 of a 4096 x 4096 layer: groups are independent
        import ctypes
        try:
            gomp = ctypes.CDLL("libgomp.so.1")
        except OSError:
            gomp = None
        best = None
        for th in (sorted({min(cores, 8), min(cores, 32), cores}) if gomp is not None else [0]):   # OpenMP does not scale to every core of a big host either
            if gomp is not None:
                gomp.omp_set_num_threads(int(th))
            t0 = time.perf_counter()
            r = orc.quantize(Wn, nbits=nbits if nbits in (8, 4, 3, 2) else 4, group_size=64)
            el = time.perf_counter() - t0
            if best is None or el < best[0]:
                best = (el, th, int(r["iters_run"]))
        el, th, its = best
        out["quantize"] = {"s_per_4096x4096_layer": round(el * 8, 3), "iters_run": its, "threads": th,
                           "sample": f"oracle/hqq_oracle.c quantize (the reference's CPU float32 solver restated, OpenMP, {th or 'default'} threads: the fastest of 8/32/{cores}) "
                                     f"on 512 x 4096 weights: {el:.2f} s, x 8 for the layer"}
    except Exception as e:
        out["quantize"] = {"error": repr(e)}
    return out


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and rank == 0:
        print(f"[bench] note: --gpus {a.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    assert torch.cuda.is_available(), "bench.py needs a ROCm GPU (the HIP path has no CPU fallback)"
    if os.environ.get("HQQ_BENCH_ONE_GPU"):   # debug: every rank on GPU 0 (single-GPU boxes; use with HQQ_BENCH_BACKEND=gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("HQQ_BENCH_BACKEND", "nccl")   # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    from hqq_amd import ops
    assert ops.is_available(), "libhqq_hip.so must be built (python -c 'import __graft_entry__ as g; g.build()')"
    nbits = a.nbits
    workload = a.workload
    if workload == "auto":
        workload = "decode" if world == 1 else "decode70b"
    decode = workload in ("decode", "decode70b")
    big = workload == "decode70b" or (workload == "prefill" and a.shapes == "70b")
    cd = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    assert a.dtype == "f16" or (decode and (a.bs >= 5 or nbits in (4, 2))), "bf16: decode only; bs <= 4 needs int4 / int2 (bs 5..64: int8/4/2)"
    M = a.bs if decode else a.prefill_tokens
    BLOCK = LLAMA2_70B_BLOCK if big else LLAMA2_7B_BLOCK
    nblocks = (a.blocks or (N_BLOCKS_70B if big else N_BLOCKS_7B)) if decode else 1
    strong = big and world > 1
    if strong:
        per = 1 if nbits == 3 else 8 // nbits
        for _, N, _ in BLOCK:
            assert N % (per * world) == 0, f"column shard: N={N} must divide by {per} * {world} (hqq_amd/shard.py)"

    # the shard plan (strong scaling only): which exchange groups are column-sharded — and exchange — and which every rank holds whole
    plan = ["sharded"] * len(EXCHANGE_GROUPS)
    if strong and a.plan == "adaptive":
        from hqq_amd import shard as _shard_plan
        plan = _shard_plan.plan_exchange_groups([sum(wq_bytes(N, K, nbits) for name, N, K in BLOCK if name in grp) for grp in EXCHANGE_GROUPS], world)
    whole = {n: (pl != "sharded") for grp, pl in zip(EXCHANGE_GROUPS, plan) for n in grp}
    XGROUPS = [grp for grp, pl in zip(EXCHANGE_GROUPS, plan) if pl == "sharded"]   # the groups with an exchange behind them

    # ---- the resident stack: every layer distinct in HBM ----
    t_setup = time.perf_counter()
    blocks = []
    for b in range(nblocks):
        blk = {}
        for i, (name, N, K) in enumerate(BLOCK):
            # strong scaling: rank r holds N / P output columns of the layer (as a layer of its own: the shard of a packed-row
            # block of the reference layout is exactly the reference layout of a layer with N / P rows — hqq_amd/shard.py —
            # so a synthetic shard is quantised directly; tests/test_shard.py proves the slicing against whole layers)
            n_loc = N // world if (strong and not whole[name]) else N
            blk[name] = make_layer(ops, name, n_loc, K, nbits, dev, seed=(1000 * rank if not strong else (0 if whole[name] else 7919 * rank)) + 16 * b + i,
                                   random_codes=a.random_codes, cd=cd)   # (a replicated layer is the same layer on every rank: same seed)
        blocks.append(blk)
    gx = torch.Generator(device=dev).manual_seed(1)       # x is replicated: same seed on every rank
    xs = {K: torch.randn(M, K, device=dev, generator=gx).to(cd) for K in sorted({K for _, _, K in BLOCK})}
    dimN = {n: blocks[0][n].N for n, _, _ in BLOCK}
    # per exchange group: local outputs [len(group)][M, N_loc] and, for P > 1, the gathered + un-permuted [M, N] per layer
    out_local = {grp: [torch.empty(M, dimN[n], device=dev, dtype=cd) for n in grp] for grp in EXCHANGE_GROUPS}
    out_flat, out_gath, out_full = {}, {}, {}
    if world > 1:
        for grp in XGROUPS:
            tot = sum(dimN[n] for n in grp)
            out_flat[grp] = torch.empty(M * tot, device=dev, dtype=cd)            # this rank's outputs of the group, back to back
            out_gath[grp] = torch.empty(world * M * tot, device=dev, dtype=cd)    # rank-major concatenation
            out_full[grp] = [torch.empty(M, world * dimN[n], device=dev, dtype=cd) for n in grp]
            off = 0
            views = []
            for n in grp:   # the kernels write straight into the flat send buffer
                views.append(out_flat[grp][off:off + M * dimN[n]].view(M, dimN[n]))
                off += M * dimN[n]
            out_local[grp] = views
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup

    grouped = decode and not a.no_group and nbits in (4, 3, 2, 8, 1)
    base_opts = ops.OPT_FACTORED if a.gemv_mode == "factored" else 0

    extra_opts = int(os.environ.get("HQQ_BENCH_OPTS", "0"), 0)   # study switch: option bits OR-ed into every decode call (e.g. 4096 = OPT_BATCH_OLD)

    def group_opts(Ls):
        lay = (ops.OPT_W3S if (Ls[0].opts & ops.OPT_W3S) else 0) | extra_opts   # (the layout bit describes the tensor: it travels in every mode)
        if a.gemv_mode != "exact":
            return base_opts | lay
        return (ops.OPT_META_SCALABLE if all(L.opts & ops.OPT_META_SCALABLE for L in Ls) else 0) | lay

    per_slab = 1 if nbits == 3 else 8 // nbits

    # One activation row (the decode headline): slab s of a layer's full output row is the rank-major concatenation of every rank's run s
    # (hqq_amd.shard.gather_columns), so per-slab all-gathers land the outputs STRAIGHT in the reference's column order — one coalesced
    # collective per exchange point where the backend can (RCCL: one grouped launch), no un-permute kernels.  Probed once below; a backend
    # that cannot falls back to the shard-wide gather + un-permute.
    xmode = {"coalesced": False}

    def exchange_rows1(grp, coalesce):
        cm = getattr(dist, "_coalescing_manager", None) if coalesce else None

        def issue():
            for j, n in enumerate(grp):
                nl = dimN[n]
                N_, n1 = world * nl, nl // per_slab
                src, dst = out_local[grp][j].reshape(-1), out_full[grp][j].reshape(-1)
                for s_ in range(per_slab):
                    dist.all_gather_into_tensor(dst[s_ * (N_ // per_slab):(s_ + 1) * (N_ // per_slab)], src[s_ * n1:(s_ + 1) * n1])
        if cm is not None:
            with cm(device=dev):
                issue()
        else:
            issue()

    def exchange(grp):
        """all-gather the group's shard outputs, then restore the reference's column order: rank r's packed-row block holds,
        per slab s, output columns s * N/per + [r * n', (r + 1) * n'), n' = N / (per * P)  (hqq_amd.shard.unpermute,
        the function tests/test_shard.py checks against whole layers, writing into a preallocated buffer here)"""
        from hqq_amd import shard
        if xmode.get("peer") is not None:   # one kernel: slices stored straight into every rank's full rows over peer memory
            xmode["peer"].run(XGROUPS.index(grp), out_local[grp])
            return
        if M == 1 and xmode.get("rows1"):
            exchange_rows1(grp, xmode["coalesced"])
            return
        dist.all_gather_into_tensor(out_gath[grp], out_flat[grp])
        tot = sum(dimN[n] for n in grp)
        g = out_gath[grp].view(world, M * tot)
        off = 0
        for j, n in enumerate(grp):
            nl = dimN[n]
            out_full[grp][j].copy_(shard.unpermute(g[:, off:off + M * nl].reshape(world, M, nl), world * nl, nbits, world))
            off += M * nl

    # --streams S > 1 (study mode, not the headline): the step's launches are dealt over S parallel graph branches, i.e. the
    # dependency chain q|k|v -> o -> gate|up -> down of a real decoder is NOT modelled and consecutive launches may overlap.
    S = max(1, a.streams)
    branch_streams = [torch.cuda.Stream() for _ in range(S - 1)] if S > 1 else []
    if S > 1:
        out_local_s = [{g: [torch.empty_like(t) for t in ts] for g, ts in out_local.items()} for _ in range(S)]

    route = "library" if a.library_gemm else a.prefill_route
    PREFILL_CHUNK = a.prefill_chunk if a.prefill_chunk > 0 else max(M, 1)
    route_kw = {"auto": {"fused": None}, "fused": {"fused": True}, "dense": {"fused": False}, "library": {"fused": False, "library_gemm": True}}[route]

    def step(bs_x=None, outs_by_grp=None, only_exchange=False):
        X = xs if bs_x is None else bs_x
        OL = out_local if outs_by_grp is None else outs_by_grp
        if S > 1 and bs_x is None:
            main = torch.cuda.current_stream()
            for st in branch_streams:
                st.wait_stream(main)
            i = 0
            for blk in blocks:
                for grp in EXCHANGE_GROUPS:
                    b = i % S
                    i += 1
                    with torch.cuda.stream(main if b == 0 else branch_streams[b - 1]):
                        Ls = [blk[name] for name in grp]
                        ops.gemv_grouped(X[Ls[0].K], [(L.Wq, L.scale, L.zero, None, L.N) for L in Ls], Ls[0].K, 64, nbits,
                                         outs=out_local_s[b][grp], opts=group_opts(Ls))
            for st in branch_streams:
                main.wait_stream(st)
            return
        for blk in blocks:
            for grp in EXCHANGE_GROUPS:
                Ls = [blk[name] for name in grp]
                if not only_exchange:
                    if grouped:   # q/k/v and gate/up read the same x: one launch per exchange group
                        ops.gemv_grouped(X[Ls[0].K], [(L.Wq, L.scale, L.zero, None, L.N) for L in Ls], Ls[0].K, 64, nbits,
                                         outs=OL[grp], opts=group_opts(Ls))
                    else:
                        for j, L in enumerate(Ls):   # prefill: chunks of PREFILL_CHUNK tokens (configs[2]: 65,536 tokens = 8 x 8192)
                            for c0 in range(0, M, PREFILL_CHUNK):
                                ops.forward(X[L.K][c0:c0 + PREFILL_CHUNK], L.Wq, L.scale, L.zero, None, L.N, L.K, 64, nbits, out=OL[grp][j][c0:c0 + PREFILL_CHUNK],
                                            opts=group_opts([L]), **route_kw)
                if world > 1 and bs_x is None and not whole[grp[0]]:
                    exchange(grp)

    # auto (default): the collective — coalesced per-slab gathers on RCCL, else the shard-wide gather.  peer: the peer-memory kernel
    # (csrc/exchange.hip) when it validates against the collective at start-up — opt-in until it has run over xGMI on a real node (advisor,
    # round 3: nothing here has been measured on more than one GPU).  rows1 / gather force one collective form.
    xenv = os.environ.get("HQQ_BENCH_EXCHANGE", "auto")
    if world > 1 and M <= 64 and strong and xenv == "peer":   # (round 5: a decode batch of up to 64 rows goes through the same kernel: strided slab writes, no un-permute)
        # Build the peer arenas (collective), then VALIDATE three rounds of every exchange point against the collective on fresh random
        # slices; every rank must agree, else the mode is dropped.  Waits are bounded: a peer that never delivers is reported, not hung on.
        from hqq_amd import shard as _shard
        ok, why, px = 1.0, "", None
        try:   # (collective, and consistent: either every rank gets its object or every rank raises — hqq_amd/shard.py)
            px = _shard.PeerExchange([[world * dimN[n] for n in grp] for grp in XGROUPS], nbits, cd, dev, rows=M)
        except Exception as e:   # noqa: BLE001
            ok, why = 0.0, f"{type(e).__name__}: {e}"
        if px is not None:
            gv = torch.Generator(device=dev).manual_seed(977 + rank)
            alive = True   # (a rank whose peer path raised keeps taking part in the collectives of the remaining rounds)
            for rnd in range(3):
                # the collectives first, all of them, outside any try: a rank whose peer path fails below must not leave the others in one
                for grp in XGROUPS:
                    for t in out_local[grp]:
                        t.copy_(torch.randn(t.shape, device=dev, generator=gv).to(cd))
                    exchange(grp)                       # the collective (xmode holds no peer object yet) -> out_full
                want = {grp: [t.clone() for t in out_full[grp]] for grp in XGROUPS}
                torch.cuda.synchronize()
                if not alive:
                    continue
                try:   # the peer path: no collective inside, waits bounded
                    for e, grp in enumerate(XGROUPS):
                        px.run(e, out_local[grp])
                        torch.cuda.synchronize()
                        for j in range(len(grp)):
                            if not torch.equal(px.full(e, j), want[grp][j]):
                                ok, why = 0.0, f"rows differ from the collective's (round {rnd}, point {e}, layer {j})"
                    if px.status() != 0:
                        ok, why = 0.0, f"a wait gave up (status {px.status()})"
                except Exception as e:   # noqa: BLE001
                    ok, why, alive = 0.0, f"{type(e).__name__}: {e}", False
        okt = torch.tensor([ok], device=dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        if float(okt) > 0:
            xmode["peer"] = px
            for e, grp in enumerate(XGROUPS):
                out_full[grp] = [px.full(e, j) for j in range(len(grp))]
        elif why:
            print(f"[bench] rank {rank}: peer-memory exchange not used: {why}", file=sys.stderr)
    if world > 1 and M == 1 and xenv not in ("gather", "peer") and xmode.get("peer") is None:
        # probe the per-slab exchange once, eagerly, on every rank.  Only the COALESCED form (one launch per exchange point) is worth having:
        # issued one by one it is `per` times as many collectives as the shard-wide gather
        for coalesce in ((True,) if dist.get_backend() == "nccl" else ((False,) if xenv == "rows1" else ())):
            try:
                for grp in XGROUPS:
                    exchange_rows1(grp, coalesce)
                torch.cuda.synchronize()
                ok = torch.ones(1, device=dev)
            except Exception as e:   # noqa: BLE001
                if rank == 0:
                    print(f"[bench] per-slab exchange (coalesce={coalesce}) not available: {type(e).__name__}: {e}", file=sys.stderr)
                ok = torch.zeros(1, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)   # every rank must agree on the mode
            if float(ok) > 0:
                xmode["rows1"], xmode["coalesced"] = True, coalesce
                break
    use_graph = not a.no_graph and os.environ.get("HQQ_BENCH_GRAPH", "1") != "0" and (world == 1 or dist.get_backend() == "nccl")
    run, graphed = _graphed(step, use_graph, rank)
    mode_name = {"exact": "exact", "exact4": "exact (four-op rebuild forced)", "factored": "factored"}[a.gemv_mode]

    sec_per_step, dev_sec_per_step = _timed(run, a.steps, a.warmup, dist, dev)
    if xmode.get("peer") is not None:   # a wait that gave up inside the timed region: the rows of that exchange were undefined — not a measurement
        st_ = xmode["peer"].status()
        assert st_ == 0, f"rank {rank}: the peer-memory exchange reported a wait that gave up (status {st_}) inside the timed region"

    # ---- accounting ----
    launches_per_step = nblocks * (len(EXCHANGE_GROUPS) if grouped else len(BLOCK))
    stages_per_step = nblocks * len(EXCHANGE_GROUPS)
    bytes_per_step_rank = nblocks * sum(gemv_bytes(dimN[n], K, nbits, M) for n, _, K in BLOCK)
    flops_per_step_rank = nblocks * sum(2.0 * M * dimN[n] * K for n, _, K in BLOCK)
    out = {
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(sec_per_step * 1e3, 5),
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
    }
    model = "llama2-70b" if big else "llama2-7b"
    if decode:
        # whole-job algorithmic bytes: a sharded layer's shards add up to the layer; a replicated layer counts ONCE (every rank streaming it is
        # redundant work the plan chose, not throughput)
        job_bytes = nblocks * sum((1 if (strong and whole[n]) else world) * gemv_bytes(dimN[n], K, nbits, M) for n, _, K in BLOCK)
        gbs = job_bytes / sec_per_step / 1e9
        n_scal = sum(1 for blk in blocks for L in blk.values() if L.opts & ops.OPT_META_SCALABLE)
        out.update({
            "metric": f"int{nbits} gs=64 dequant-GEMV decode throughput, {model} linear stack bs={M} (algorithmic GB/s; tok/s alongside)",
            "value": round(gbs, 2), "unit": "GB/s",
            "tok_s": round(M / sec_per_step, 2),
            "config": {"workload": f"{model} linear stack ({nblocks} blocks x q,k,v,o,gate,up,down), nbits={nbits} gs=64 axis=1, bs={M} decode, "
                                   f"{'bf16' if a.dtype == 'bf16' else 'fp16'}, " +
                                   f"{launches_per_step} fused dequant-GEMV launches/step ({'q|k|v, o, gate|up, down grouped' if grouped else 'one per layer'})" +
                                   (", hipGraph replay" if graphed else ", eager launches") + (f", {S} parallel branches (dependency chain NOT modelled)" if S > 1 else ""),
                       "global_batch": M,
                       "parallelism": "single-gpu" if world == 1 else
                                      (f"output-column shard x{world} of fixed layers (strong scaling) + " + ("peer-memory exchange kernel (hqq_hip_exchange, xGMI stores)" if xmode.get("peer") is not None else "RCCL per-slab all-gathers (coalesced)" if xmode.get("rows1") else "RCCL all-gather + un-permute") + " per exchange point, in the timed region"
                                       if strong else f"column-shard x{world} + RCCL all-gather (weak: {world}x wider layers)"),
                       "gemv_mode": mode_name, "layers_with_three_op_rebuild": f"{n_scal}/{nblocks * len(BLOCK)}",
                       "bytes_per_step_per_gpu": bytes_per_step_rank, "setup_s": round(t_setup, 2)},
        })
        if strong:
            out["plan"] = {"name": a.plan, "groups": dict(zip(["|".join(g_) for g_ in EXCHANGE_GROUPS], plan)), "exchange_points_per_block": len(XGROUPS),
                           "job_bytes_per_step": job_bytes,
                           "note": "hqq_amd.shard.plan_exchange_groups; a replicated-small group is computed whole by every rank (counted once in `value`), no exchange behind it"}
        unit_launches = stages_per_step   # the unit the roofline is quoted per: one dependent stage (= one launch on the launch path)
        avg_launch_s = dev_sec_per_step / unit_launches
        ach = (bytes_per_step_rank / unit_launches) / avg_launch_s / 1e9
        out["roofline"] = {"bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                           # PMC traffic is committed for the configuration it was measured on only (7B, bs=1 fp16, exact mode)
                           "traffic": None,   # (not counted in this run; the committed counter run of this configuration follows)
                           "traffic_committed_pmc": _pmc_traffic(nbits) if (M == 1 and a.dtype == "f16" and a.gemv_mode != "factored" and not big) else None,
                           "kernel": _decode_kernel_name(nbits, M, a.dtype, a.gemv_mode),
                           "avg_launch_us": round(avg_launch_s * 1e6, 3),
                           "bytes_per_launch": bytes_per_step_rank // unit_launches,
                           "note": "avg launch = HIP-event time of the timed region / dependent stages (includes inter-kernel gaps"
                                   + ("; and the all-gathers" if world > 1 else "") + ")"}
        if world > 1:   # the exchange alone (same graph structure without the GEMV launches)
            xrun, _ = _graphed(lambda: step(only_exchange=True), use_graph, rank)
            xs_, _ = _timed(xrun, max(5, a.steps // 2), 3, dist, dev)
            rows1 = bool(M == 1 and xmode.get("rows1"))
            peer = xmode.get("peer") is not None
            x_points = nblocks * len(XGROUPS)                              # exchange points per step under the plan
            x_layers = nblocks * sum(len(g_) for g_ in XGROUPS)
            n_slab_gathers = x_layers * per_slab
            out["exchange"] = {"ms_per_step": round(xs_ * 1e3, 5),
                               "mode": ("peer-memory stores (hqq_hip_exchange): one kernel per exchange point writes the rank's slices into every rank's full rows in the reference's column order and waits for the others' flags; validated against the collective at start-up"
                                        if peer else
                                        ("per-slab all-gathers straight into the reference's column order" + (", one coalesced RCCL launch per exchange point" if xmode["coalesced"] else ", issued one by one")) if rows1 else "one all-gather of the shard outputs per exchange point + un-permute copies"),
                               "points_per_step": x_points, "us_per_point": round(xs_ * 1e6 / max(1, x_points), 3),
                               "exchange_kernels_per_step": x_points if peer else 0,
                               "peer_status": xmode["peer"].status() if peer else None,
                               "peer_memory": xmode["peer"].memory_kind if peer else None,
                               "collective_launches_per_step": 0 if peer else ((x_points if xmode["coalesced"] else n_slab_gathers) if rows1 else x_points),
                               "all_gathers_per_step": 0 if peer else (n_slab_gathers if rows1 else x_points),
                               "unpermute_kernels_per_step": 0 if (rows1 or peer) else x_layers,
                               "bytes_sent_per_rank_per_step": 2 * M * nblocks * sum(dimN[n] for n, _, _ in BLOCK if not whole[n]),
                               "note": "the exchange of every exchange point, timed without the GEMV launches"}
            if strong and not a.no_single_gpu_reference:
                # the SAME fixed stack on ONE GPU (rank 0 alone, the others wait): the single-GPU time a strong-scaling figure refers to
                single = None
                if rank == 0:
                    try:
                        fb = []
                        for b_ in range(nblocks):
                            fb.append({name: make_layer(ops, name, N, K, nbits, dev, seed=424242 + 16 * b_ + i, random_codes=True, cd=cd)
                                       for i, (name, N, K) in enumerate(BLOCK)})
                        fo = {grp: [torch.empty(M, fb[0][n].N, device=dev, dtype=cd) for n in grp] for grp in EXCHANGE_GROUPS}

                        def full_step():
                            for blk in fb:
                                for grp in EXCHANGE_GROUPS:
                                    Ls = [blk[name] for name in grp]
                                    ops.gemv_grouped(xs[Ls[0].K], [(L.Wq, L.scale, L.zero, None, L.N) for L in Ls], Ls[0].K, 64, nbits,
                                                     outs=fo[grp], opts=group_opts(Ls))
                        frun, fg = _graphed(full_step, not a.no_graph, rank)
                        fw, fd = _timed(frun, max(5, a.steps // 2), 3)
                        fbytes = nblocks * sum(gemv_bytes(N, K, nbits, M) for _, N, K in BLOCK)
                        single = {"ms_per_step": round(fw * 1e3, 5), "value": round(fbytes / fw / 1e9, 2), "unit": "GB/s", "graph": fg,
                                  "note": "the whole (unsharded) stack of this workload on rank 0's GPU alone, random codes, same kernels, no exchange"}
                        del fb, fo
                        torch.cuda.empty_cache()
                    except Exception as e:   # (memory, mostly: 35 GB of int4 70B weights beside the shard)
                        single = {"error": repr(e)}
                dist.barrier()
                if rank == 0:
                    out["single_gpu"] = single
    else:
        Mc = min(M, PREFILL_CHUNK)
        took_fused = [bool(ops._C.lib().hqq_hip_forward_prefers_fused(nbits, Mc, N_, K_, 64, 1)) for _, N_, K_ in BLOCK] if route == "auto" else []
        route_txt, route_kernel = {
            "fused": ("fused MFMA dequant-GEMM (hqq_hip_gemm: pipelined kernel)", "hqq::gemm_pipe_f16_kernel (LDS-DMA rings, weights rebuilt into MFMA fragments)"),
            "dense": ("dequantise kernel + in-tree MFMA GEMM (hqq_hip_dequantize + hqq_hip_gemm_dense)", "hqq::gd::dense_gemm_kernel + hqq::dequantize"),
            "library": ("dequantise kernel + library GEMM (comparison, not a product path)", "hqq::dequantize + hipBLASLt"),
            "auto": (f"ops.forward's own route: {sum(took_fused)}/{len(BLOCK)} layers on the fused MFMA dequant-GEMM, the rest dequantise kernel + in-tree MFMA GEMM "
                     "(no library call)", "hqq::gd::dense_gemm_kernel + hqq::dequantize" if not all(took_fused) else "hqq::gemm_pipe_f16_kernel"),
        }[route]
        tfl = world * flops_per_step_rank / sec_per_step / 1e12
        out.update({
            "metric": f"int{nbits} gs=64 dequant-GEMM prefill throughput, Llama-2-{'70B' if big else '7B'} block M={M} (tok/s; TFLOP/s alongside)",
            "value": round(M / sec_per_step, 2), "unit": "tok/s (one block's 7 linears)", "tflops": round(tfl, 2),
            "config": {"workload": f"llama2-{'70b' if big else '7b'} one block (q,k,v,o,gate,up,down), nbits={nbits} gs=64 axis=1, M={M} prefill tokens"
                                   + (f" as {-(-M // PREFILL_CHUNK)} chunks of {PREFILL_CHUNK}" if M > PREFILL_CHUNK else "") + f", fp16, {route_txt}",
                       "global_batch": M, "parallelism": "single-gpu" if world == 1 else f"column-shard x{world} + RCCL all-gather"},
        })
        ach = flops_per_step_rank / dev_sec_per_step / 1e12
        out["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS, 4),
                           "traffic": None, "kernel": route_kernel, "mfma_busy": _pmc_dense_busy()}

    # ---- the other shapes the metric names, same resident weights (N = 1, default decode only) ----
    if decode and world == 1 and not a.no_legs and not big and M == 1 and a.dtype == "f16" and S == 1 and nbits in (4, 2, 8):
        legs = []

        def leg(name, fn, nbytes, rows, launches, kernel):
            r, g_ = _graphed(fn, use_graph, rank)
            w, d = _timed(r, max(5, a.steps // 2), 3)
            legs.append({"name": name, "ms_per_step": round(w * 1e3, 5), "value": round(nbytes / w / 1e9, 2), "unit": "GB/s", "tok_s": round(rows / w, 2),
                         "launches_per_step": launches, "avg_launch_us": round(d / launches * 1e6, 3),
                         "roofline_frac": round(nbytes / d / 1e9 / HBM_PEAK_GBS, 4), "kernel": kernel, "graph": g_})

        xs32 = {K: torch.randn(32, K, device=dev, generator=gx).to(cd) for K in xs}
        ol32 = {grp: [torch.empty(32, dimN[n], device=dev, dtype=cd) for n in grp] for grp in EXCHANGE_GROUPS}
        leg("7b-stack bs=32", lambda: step(bs_x=xs32, outs_by_grp=ol32), nblocks * sum(gemv_bytes(N, K, nbits, 32) for _, N, K in BLOCK), 32,
            nblocks * len(EXCHANGE_GROUPS), _decode_kernel_name(nbits, 32, a.dtype, a.gemv_mode))
        # ---- three legs that carry the argument of DESIGN.md section 3.7 in the driver's own run ----
        bytes_7b = nblocks * sum(gemv_bytes(N, K, nbits, 1) for _, N, K in BLOCK)
        n_launch = nblocks * len(EXCHANGE_GROUPS)
        kname1 = _decode_kernel_name(nbits, 1, a.dtype, a.gemv_mode)
        # (i) KERNEL STUDY, not a decoder: the same 128 launches dealt over three parallel graph branches — no launch waits for its predecessor.
        #     What the stream-ordered chain costs is the difference to the headline.
        if a.gemv_mode == "exact":
            try:
                br = [torch.cuda.Stream() for _ in range(2)]
                ol3 = [{g_: [torch.empty_like(t) for t in ts] for g_, ts in out_local.items()} for _ in range(3)]

                def step_nodep():
                    main_s = torch.cuda.current_stream()
                    for st in br:
                        st.wait_stream(main_s)
                    i = 0
                    for blk in blocks:
                        for grp in EXCHANGE_GROUPS:
                            b = i % 3
                            i += 1
                            with torch.cuda.stream(main_s if b == 0 else br[b - 1]):
                                Ls = [blk[name] for name in grp]
                                ops.gemv_grouped(xs[Ls[0].K], [(L.Wq, L.scale, L.zero, None, L.N) for L in Ls], Ls[0].K, 64, nbits, outs=ol3[b][grp], opts=group_opts(Ls))
                    for st in br:
                        main_s.wait_stream(st)
                leg("7b-stack bs=1, NO dependency chain: 3 parallel graph branches (kernel study — a decoder's launches depend on each other)", step_nodep, bytes_7b, 1, n_launch, kname1)
                legs[-1]["study"] = True
                del ol3
            except Exception as e:
                legs.append({"name": "7b-stack bs=1, NO dependency chain", "error": repr(e)})
            # (ii) HQQ_OPT_FACTORED: the group's affine map taken out of the dot product — NOT the reference's twice-rounded weights (opt-in; tests state its error)
            if nbits in (4, 2, 8):
                def step_factored():
                    for blk in blocks:
                        for grp in EXCHANGE_GROUPS:
                            Ls = [blk[name] for name in grp]
                            ops.gemv_grouped(xs[Ls[0].K], [(L.Wq, L.scale, L.zero, None, L.N) for L in Ls], Ls[0].K, 64, nbits, outs=out_local[grp], opts=ops.OPT_FACTORED)
                leg("7b-stack bs=1 FACTORED (HQQ_OPT_FACTORED: not reference-exact weights, opt-in)", step_factored, bytes_7b, 1, n_launch, _decode_kernel_name(nbits, 1, a.dtype, "factored"))
                legs[-1]["reference_exact"] = False
            # (iii) the Llama-2-70B linear stack (BASELINE.json configs[4]'s shapes, 80 blocks, 35 GB at 4 bits) on THIS ONE GPU: the same kernel with
            #       launches of 19-264 MB instead of 9-51 MB — the per-launch fixed cost amortised
            try:
                free_b, _ = torch.cuda.mem_get_info()
                need = N_BLOCKS_70B * sum(gemv_bytes(N, K, nbits, 1) for _, N, K in LLAMA2_70B_BLOCK)
                if free_b > need * 1.15 + (2 << 30):
                    b70 = [{name: make_layer(ops, name, N, K, nbits, dev, seed=70000 + 16 * b_ + i, random_codes=a.random_codes, cd=cd) for i, (name, N, K) in enumerate(LLAMA2_70B_BLOCK)}
                           for b_ in range(N_BLOCKS_70B)]
                    x70 = {K: torch.randn(1, K, device=dev, generator=gx).to(cd) for K in sorted({K for _, _, K in LLAMA2_70B_BLOCK})}
                    o70 = {grp: [torch.empty(1, b70[0][n].N, device=dev, dtype=cd) for n in grp] for grp in EXCHANGE_GROUPS}

                    def step70():
                        for blk in b70:
                            for grp in EXCHANGE_GROUPS:
                                Ls = [blk[name] for name in grp]
                                ops.gemv_grouped(x70[Ls[0].K], [(L.Wq, L.scale, L.zero, None, L.N) for L in Ls], Ls[0].K, 64, nbits, outs=o70[grp], opts=group_opts(Ls))
                    leg("70b-stack bs=1 on this one GPU (80 blocks, unsharded)", step70, need, 1, N_BLOCKS_70B * len(EXCHANGE_GROUPS), kname1)
                    legs[-1]["layers_with_three_op_rebuild"] = f"{sum(1 for blk in b70 for L in blk.values() if L.opts & ops.OPT_META_SCALABLE)}/{N_BLOCKS_70B * len(LLAMA2_70B_BLOCK)}"
                    del b70, x70, o70
                    torch.cuda.empty_cache()
                else:
                    legs.append({"name": "70b-stack bs=1 on this one GPU", "skipped": f"needs {need / 1e9:.1f} GB, {free_b / 1e9:.1f} GB free"})
            except Exception as e:
                legs.append({"name": "70b-stack bs=1 on this one GPU", "error": repr(e)})
        qs = [blk["q"] for blk in blocks]   # 32 distinct 4096x4096 layers (268 MB packed: beyond the Infinity Cache)
        y1 = torch.empty(1, 4096, device=dev, dtype=cd)
        y32 = torch.empty(32, 4096, device=dev, dtype=cd)

        def single(xin, y):
            for L in qs:
                ops.gemv(xin, L.Wq, L.scale, L.zero, None, L.N, L.K, 64, nbits, out=y, opts=group_opts([L]))
        leg("4096x4096 bs=1 (one layer per launch)", lambda: single(xs[4096], y1), len(qs) * gemv_bytes(4096, 4096, nbits, 1), 1, len(qs),
            _decode_kernel_name(nbits, 1, a.dtype, a.gemv_mode))
        leg("4096x4096 bs=32 (one layer per launch)", lambda: single(xs32[4096], y32), len(qs) * gemv_bytes(4096, 4096, nbits, 32), 32, len(qs),
            _decode_kernel_name(nbits, 32, a.dtype, a.gemv_mode))
        # the other hot path: Quantizer.quantize = solver + packing, one HIP launch chain per layer
        try:
            qres = []
            sv = _solver_valu()
            for nb_q in [nbits] + [w for w in (4, 3, 2) if w != nbits]:   # the run's width first, then the other widths of BASELINE.json configs[3]
                for nm, N_, K_ in (("4096x4096", 4096, 4096), ("11008x4096", 11008, 4096), ("4096x11008", 4096, 11008)):
                    Wsrc = (torch.randn(N_, K_, device=dev, generator=gx) * 0.02).half()
                    ops.quantize(Wsrc, nbits=nb_q, group_size=64, round_zero=(nb_q == 4))
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(3):
                        _, _, _, info = ops.quantize(Wsrc, nbits=nb_q, group_size=64, round_zero=(nb_q == 4), return_info=True)
                    e1.record()
                    torch.cuda.synchronize()
                    ms = e0.elapsed_time(e1) / 3
                    its = int(info[0].item())
                    qres.append({"nbits": nb_q, "layer": nm, "ms": round(ms, 3), "iters_run": its, "G_element_iters_per_s": round(N_ * K_ * 20 / (ms * 1e-3) / 1e9, 1),
                                 "hbm_floor_ms": round((2 + (4 if nb_q == 3 else nb_q) / 8) * N_ * K_ / (HBM_PEAK_GBS * 1e9) * 1e3, 4)})
                    if sv:   # the bound, stated: VALU issue (tools/solver_valu_count.py reads the instruction count off the ISA of this build)
                        qres[-1]["valu_frac"] = round(qres[-1]["G_element_iters_per_s"] / sv["peak_G_element_iters_per_s"], 4)
                    del Wsrc
            out["quantize"] = {"layers": qres,
                               "roofline": None if not sv else {"bound": "valu", "unit": "G element-iterations/s", "peak": sv["peak_G_element_iters_per_s"],
                                                                "valu_instr_per_element_iteration": sv["valu_instr_per_element_iteration"],
                                                                "issue_cycles_per_element_iteration": sv["issue_cycles_per_element_iteration"],
                                                                "note": "peak = 1024 SIMDs x 2.4 GHz / issue cycles per element-iteration of solve_kernel's fast path (4 cycles per VALU "
                                                                        "wave-instruction, 16 for the v_rcp_f32 of the IEEE division; profiles/solver_valu.json); achieved = N K 20 / time of the "
                                                                        "whole Quantizer.quantize call (solver + error reduction + final levels + packing)"},
                               "note": "Quantizer.quantize (20 proximal iterations computed, stop index applied as the reference does) + bit-packing, fp16 weights in HBM; "
                               "bound: VALU (float32 division, rounding, clamp, sign, the ATen-ordered row sum; the double-precision pow of the shrinkage is evaluated only in waves "
                               "where it can matter), hbm_floor = one read of W + one write of W_q"}
        except Exception as e:
            out["quantize"] = {"error": repr(e)}
        # the stand-alone bit-packing kernels (rows a5, a9, a10, a13), HBM-bound: 11008 x 4096 levels at every width
        try:
            bp = []
            Nb, Kb = 11008, 4096
            Rb = Nb * Kb // 64
            for nb_q in (4, 3, 2):
                U = torch.randint(0, 2 ** nb_q, (Rb, 64), device=dev, dtype=torch.uint8, generator=gx)
                P = ops.pack(nb_q, U)
                sc = (torch.rand(Rb, 1, device=dev, generator=gx) * 0.004 + 0.001).to(cd)
                zc = (torch.rand(Rb, 1, device=dev, generator=gx) * (2 ** nb_q - 1)).to(cd)
                pbytes = P.numel() * P.element_size()

                def tm(fn, n=10):
                    fn()
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(n):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                    return e0.elapsed_time(e1) / n * 1e-3
                for nm, fn, by in (("pack (uint8 levels -> reference container)", lambda: ops.pack(nb_q, U), Rb * 64 + pbytes),
                                   ("unpack (-> uint8 levels)", lambda: ops.unpack(nb_q, P), pbytes + (P.shape[0] * ops.PER[nb_q]) * 64),
                                   ("dequantize (-> fp16 [N, K])", lambda: ops.dequantize(P, sc.reshape(-1), zc.reshape(-1), Nb, Kb, 64, nb_q), pbytes + 4 * Rb + 2 * Nb * Kb)):
                    sec = tm(fn)
                    bp.append({"nbits": nb_q, "op": nm, "us": round(sec * 1e6, 1), "GB_s": round(by / sec / 1e9, 1), "roofline_frac": round(by / sec / 1e9 / HBM_PEAK_GBS, 4)})
                del U, P, sc, zc
            out["bitpack"] = {"layer": f"{Nb}x{Kb}, gs=64 ({Rb} groups)", "ops": bp,
                              "note": "bytes = what the op must read + write once (output allocation by torch's caching allocator inside the timed call)"}
            torch.cuda.empty_cache()
        except Exception as e:
            out["bitpack"] = {"error": repr(e)}
        # 128 rows through every layer of the stack (batched decode / speculative verification / short prompts): the pipelined split-K
        # fused GEMM (csrc/gemm_pipe.hip) against the alternative a caller has — dequantise kernel + library GEMM on the result
        if nbits in (8, 4, 2):
            try:
                xs128 = {K: torch.randn(128, K, device=dev, generator=gx).to(cd) for K in xs}
                y128 = {N_: torch.empty(128, N_, device=dev, dtype=cd) for N_ in sorted({L.N for L in blocks[0].values()})}
                flops128 = 2.0 * 128 * nblocks * sum(N * K for _, N, K in BLOCK)

                def stack128(fused):
                    for blk in blocks:
                        for L in blk.values():
                            ops.forward(xs128[L.K], L.Wq, L.scale, L.zero, None, L.N, L.K, 64, nbits, out=y128[L.N], fused=fused, opts=group_opts([L]),
                                        library_gemm=not fused)
                for nm, fused, kern in (("7b-stack bs=128, fused dequant-GEMM (one launch + split-K reduce per layer)", True, "hqq::gemm_pipe_f16_kernel"),
                                        ("7b-stack bs=128, dequantise kernel + library GEMM (comparison only: no product path calls it)", False, "hqq::dequantize + hipBLASLt")):
                    leg(nm, lambda f=fused: stack128(f), nblocks * sum(gemv_bytes(N, K, nbits, 128) for _, N, K in BLOCK), 128, nblocks * len(BLOCK), kern)
                    legs[-1]["tflops"] = round(flops128 / (legs[-1]["ms_per_step"] * 1e-3) / 1e12, 1)
                    legs[-1]["mfma_frac"] = round(legs[-1]["tflops"] / MFMA_PEAK_TFLOPS, 4)
                # the same rows, the layers as they are, one GROUPED launch per q|k|v / o / gate|up / down (hqq_hip_gemm_grouped, round 6: the route a patched model
                # takes through backends.hip.group_llama_projections): 4 launches + 4 split-K reduces per block instead of 7 + 7, no second copy of any layer
                y128g = {grp: [torch.empty(128, dimN[n], device=dev, dtype=cd) for n in grp] for grp in EXCHANGE_GROUPS}

                def stack128_grouped():
                    for blk in blocks:
                        for grp in EXCHANGE_GROUPS:
                            Ls = [blk[n] for n in grp]
                            ops.gemm_grouped(xs128[Ls[0].K], [(L.Wq, L.scale, L.zero, None, L.N) for L in Ls], Ls[0].K, 64, nbits, outs=y128g[grp], opts=group_opts(Ls))
                leg("7b-stack bs=128, fused dequant-GEMM, q|k|v and gate|up as GROUPED launches (hqq_hip_gemm_grouped: 4 launches + 4 reduces per block, the layers' own tensors)",
                    stack128_grouped, nblocks * sum(gemv_bytes(N, K, nbits, 128) for _, N, K in BLOCK), 128, nblocks * len(EXCHANGE_GROUPS), "hqq::gemm_pipe_f16_kernel")
                legs[-1]["tflops"] = round(flops128 / (legs[-1]["ms_per_step"] * 1e-3) / 1e12, 1)
                legs[-1]["mfma_frac"] = round(legs[-1]["tflops"] / MFMA_PEAK_TFLOPS, 4)
                del y128g
                # the same rows with q|k|v and gate|up held as ONE layer each (HQQLinear.merge / ops.merge_layers: the layers' own levels and constants,
                # stacked and packed again): 4 launches per block instead of 7
                if nbits == 4 and not (blocks[0]["q"].opts & ops.OPT_W3S):
                    mblocks = []
                    for blk in blocks:
                        mb = {}
                        for key, names in (("qkv", ("q", "k", "v")), ("gu", ("gate", "up"))):
                            Wm, sm, zm, Nm = ops.merge_layers([(blk[n].Wq, blk[n].scale, blk[n].zero, blk[n].N) for n in names], blk[names[0]].K, 64, nbits)
                            ok = cd == torch.float16 and a.gemv_mode == "exact" and ops.meta_scalable(sm, zm, Nm, blk[names[0]].K, 64, nbits)
                            mb[key] = (Wm, sm, zm, Nm, blk[names[0]].K, (ops.OPT_META_SCALABLE if ok else 0) | (base_opts if a.gemv_mode != "exact" else 0) | extra_opts)
                        for n in ("o", "down"):
                            mb[n] = (blk[n].Wq, blk[n].scale, blk[n].zero, blk[n].N, blk[n].K, group_opts([blk[n]]))
                        mblocks.append(mb)
                    ym = {Nm: torch.empty(128, Nm, device=dev, dtype=cd) for Nm in sorted({v[3] for v in mblocks[0].values()})}

                    def stack128_merged():
                        for mb in mblocks:
                            for key in ("qkv", "o", "gu", "down"):
                                Wm, sm, zm, Nm, Km, om = mb[key]
                                ops.forward(xs128[Km], Wm, sm, zm, None, Nm, Km, 64, nbits, out=ym[Nm], fused=True, opts=om)
                    leg("7b-stack bs=128, fused dequant-GEMM, q|k|v and gate|up merged into one layer each (HQQLinear.merge: 4 launches per block)", stack128_merged,
                        nblocks * sum(gemv_bytes(N, K, nbits, 128) for _, N, K in BLOCK), 128, nblocks * 4, "hqq::gemm_pipe_f16_kernel")
                    legs[-1]["tflops"] = round(flops128 / (legs[-1]["ms_per_step"] * 1e-3) / 1e12, 1)
                    legs[-1]["mfma_frac"] = round(legs[-1]["tflops"] / MFMA_PEAK_TFLOPS, 4)
                    del mblocks, ym
                    torch.cuda.empty_cache()
            except Exception as e:
                legs.append({"name": "7b-stack bs=128", "error": repr(e)})
        # int3 / int2 (BASELINE.json configs[3]): the same stack quantised at the other bit widths by the HIP solver, one token, stream-ordered launches
        if nbits == 4:
            for nb in (3, 2):
                try:
                    blks = [{name: make_layer(ops, name, N, K, nb, dev, seed=50000 * nb + 16 * b + i, random_codes=a.random_codes, cd=cd)
                             for i, (name, N, K) in enumerate(BLOCK)} for b in range(nblocks)]

                    def step_nb(blks=blks, nb=nb):
                        for blk in blks:
                            for grp in EXCHANGE_GROUPS:
                                Ls = [blk[name] for name in grp]
                                o_ = group_opts(Ls)
                                ops.gemv_grouped(xs[Ls[0].K], [(L.Wq, L.scale, L.zero, None, L.N) for L in Ls], Ls[0].K, 64, nb, outs=out_local[grp], opts=o_)
                    leg(f"7b-stack bs=1 int{nb} (128 stream-ordered launches)", step_nb, nblocks * sum(gemv_bytes(N, K, nb, 1) for _, N, K in BLOCK), 1,
                        nblocks * len(EXCHANGE_GROUPS), _decode_kernel_name(nb, 1, a.dtype, a.gemv_mode))
                    if nb == 3:   # the bytes the launches actually read (stream layout: exactly 3 bits per level) beside SURVEY.md section 8d's (the reference container)
                        stored = nblocks * sum(N * K * 3 // 8 + 4 * (N * K // 64) + 2 * K + 2 * N for _, N, K in BLOCK)
                        legs[-1]["stored_bytes_per_step"] = stored
                        legs[-1]["roofline_frac_on_stored_bytes"] = round(legs[-1]["roofline_frac"] * stored / (nblocks * sum(gemv_bytes(N, K, 3, 1) for _, N, K in BLOCK)), 4)
                        legs[-1]["layout"] = "3-bit stream layout (csrc/w3s.h), re-laid out once at patch time as HQQLinearHIP does; value / roofline_frac on SURVEY.md section 8d bytes"
                        legs[-1]["three_op_rebuild"] = f"{sum(1 for blk in blks for L in blk.values() if L.opts & ops.OPT_META_SCALABLE)}/{nblocks * len(BLOCK)}"
                    del blks
                    torch.cuda.empty_cache()
                except Exception as e:
                    legs.append({"name": f"7b-stack bs=1 int{nb}", "error": repr(e)})
        # prefill (BASELINE.json configs[2], one chunk of 8192 of its 65,536 tokens): one block's seven linears, fused MFMA dequant-GEMM vs the composition
        if nbits in (8, 4, 2):
            try:
                Mp = 8192
                xsp = {K: torch.randn(Mp, K, device=dev, generator=gx).to(cd) for K in xs}
                yp = {N_: torch.empty(Mp, N_, device=dev, dtype=cd) for N_ in sorted({L.N for L in blocks[0].values()})}
                flops_p = 2.0 * Mp * sum(N * K for _, N, K in BLOCK)

                def block_prefill(kw):
                    for L in blocks[0].values():
                        ops.forward(xsp[L.K], L.Wq, L.scale, L.zero, None, L.N, L.K, 64, nbits, out=yp[L.N], opts=group_opts([L]), **kw)
                for nm, kw, kern in ((f"prefill: one 7B block, M={Mp}, fused MFMA dequant-GEMM (gemm_pipe.hip)", {"fused": True}, "hqq::gemm_pipe_f16_kernel"),
                                     (f"prefill: one 7B block, M={Mp}, dequantise kernel + in-tree MFMA GEMM (gemm_dense.hip; ops.forward's route at this M)",
                                      {"fused": None}, "hqq::dequantize + hqq::gd::dense_gemm_kernel"),
                                     (f"prefill: one 7B block, M={Mp}, dequantise kernel + library GEMM (comparison only: no product path calls it)",
                                      {"fused": False, "library_gemm": True}, "hqq::dequantize + hipBLASLt")):
                    leg(nm, lambda k=kw: block_prefill(k), sum(gemv_bytes(N, K, nbits, Mp) for _, N, K in BLOCK), Mp, len(BLOCK), kern)
                    legs[-1]["bound"] = "mfma"
                    legs[-1]["tflops"] = round(flops_p / (legs[-1]["ms_per_step"] * 1e-3) / 1e12, 1)
                    legs[-1]["mfma_frac"] = round(legs[-1]["tflops"] / MFMA_PEAK_TFLOPS, 4)
                    legs[-1].pop("roofline_frac", None)
                del xsp, yp
                torch.cuda.empty_cache()
            except Exception as e:
                legs.append({"name": "prefill 8192", "error": repr(e)})
        # end to end (the reference's headline metric is tok/s of the generate loop, hqq/utils/generation_hf.py:117-540,
        # Readme.md:153): a random-initialised Llama-2-7B-shaped HF model, every decoder linear quantised by the HIP solver, patched to the
        # fused kernels (q|k|v and gate|up grouped), greedy decode with a static KV cache, one captured hipGraph per token
        if nbits == 4 and os.environ.get("HQQ_BENCH_E2E", "1") != "0":
            try:
                from transformers import LlamaConfig, LlamaForCausalLM
                from hqq_amd.backends.hip import group_llama_projections
                from hqq_amd.core.quantize import BaseQuantizeConfig
                from hqq_amd.utils.generation import GraphedGreedyDecoder
                from hqq_amd.utils.model import quantize_model
                from hqq_amd.utils.patching import prepare_for_inference
                del blocks[:]
                torch.cuda.empty_cache()
                t0 = time.perf_counter()
                lcfg = LlamaConfig(hidden_size=4096, intermediate_size=11008, num_hidden_layers=N_BLOCKS_7B, num_attention_heads=32, num_key_value_heads=32,
                                   vocab_size=32000, max_position_embeddings=2048)
                dflt = torch.get_default_dtype()
                torch.set_default_dtype(torch.float16)
                torch.manual_seed(20250)   # the same random-init model in every run: the identity check below is then one fixed comparison, not a new draw per run
                try:
                    with torch.device(dev):
                        model = LlamaForCausalLM(lcfg).eval()
                finally:
                    torch.set_default_dtype(dflt)
                quantize_model(model, BaseQuantizeConfig(nbits=4, group_size=64, axis=1), compute_dtype=torch.float16, device=str(dev))
                import copy
                ref_model = copy.deepcopy(model)       # the same quantised model, left on HQQLinear: decodes under HQQBackend.PYTORCH_FORWARD for the identity check below
                prepare_for_inference(model, backend="hip")
                ngrp = group_llama_projections(model)
                torch.cuda.synchronize()
                t_build = time.perf_counter() - t0
                dec = GraphedGreedyDecoder(model, max_cache_len=256)
                ids = torch.randint(0, 32000, (1, 16), device=dev, generator=gx)
                r = dec.benchmark(ids, new_tokens=64, warmup=8)
                lin_ms = sec_per_step * 1e3
                out["end_to_end"] = {"tok_s": round(r["tok_s"], 2), "ms_per_token": round(r["ms_per_token"], 4), "new_tokens": r["new_tokens"], "prompt_tokens": r["prompt_tokens"],
                                     "linear_stack_ms": round(lin_ms, 4), "linear_stack_share": round(lin_ms / r["ms_per_token"], 3), "build_s": round(t_build, 1),
                                     "model": "random-init Llama-2-7B-shaped LlamaForCausalLM (32 blocks, hidden 4096, intermediate 11008, vocab 32000, fp16), every decoder linear int4 gs=64 "
                                              f"via hqq_amd.utils.model.quantize_model (HIP solver), prepare_for_inference(backend='hip'), {ngrp} grouped q|k|v / gate|up launches",
                                     "loop": "hqq_amd.utils.generation.GraphedGreedyDecoder, fused step (hqq_amd.utils.llama_fused): per decoder block add_rmsnorm -> q|k|v (grouped GEMV) -> "
                                             "rope_cache -> HF's attention function on the static cache -> o -> add_rmsnorm (residual add inside) -> gate|up (grouped GEMV) -> silu_mul -> down; "
                                             "csrc/block.hip restates the HF modules rounding for rounding; one captured hipGraph per token, argmax fed back on the device",
                                     "fused_step": bool(dec.fused), "glue": ("folded into the GEMV launches (csrc/gemv_block.hip): 4 launches + attention per block" if (dec.step is not None and dec.step.folded)
                                                                             else "separate kernels (csrc/block.hip): 8 launches + attention per block")}
                if dec.step is not None and dec.step.folded:
                    out["end_to_end"]["loop"] = ("hqq_amd.utils.generation.GraphedGreedyDecoder, fused step (hqq_amd.utils.llama_fused, glue folded): per decoder block q|k|v (grouped GEMV, RMSNorm in its "
                                                 "prologue, rotary embedding + KV-cache write in its epilogue) -> HF's attention function on the static cache -> o (residual add in its epilogue) -> gate|up as ONE paired layer (RMSNorm "
                                                 "prologue, SiLU * up epilogue) -> down (residual add in its epilogue); one captured hipGraph per token, argmax fed back on the device")
                want_new = None
                # not against itself: the first 8 greedy tokens of this loop against the SAME quantised model decoding with HF's generate under HQQBackend.PYTORCH_FORWARD
                # (dequantise + dense matmul: the reference's arithmetic, hqq/core/quantize.py:894-898)
                try:
                    from hqq_amd.core.quantize import HQQBackend, HQQLinear
                    HQQLinear.set_backend(HQQBackend.PYTORCH_FORWARD)
                    try:
                        with torch.no_grad():
                            want = ref_model.generate(ids, max_new_tokens=8, min_new_tokens=8, do_sample=False)
                    finally:
                        HQQLinear.set_backend(HQQBackend.HIP)
                    got = dec.generate(ids, 8, use_graph=True)
                    want_new = want[0, ids.shape[1]:].clone()
                    n_same = int((got[0, ids.shape[1]:] == want[0, ids.shape[1]:]).to(torch.int32).cumprod(0).sum())
                    out["end_to_end"]["identity_check"] = {"tokens": 8, "identical_prefix": n_same, "identical": bool(n_same == 8),
                                                           "want_tokens": want[0, ids.shape[1]:].tolist(), "got_tokens": got[0, ids.shape[1]:].tolist(),
                                                           "against": "HF generate() of the same quantised 7B-shaped model under HQQBackend.PYTORCH_FORWARD"}
                    if n_same < 8:
                        # where the two arithmetics part: the reference's own logits at that step (teacher-forced on the common prefix).  The two paths compute the same
                        # function in different summation orders (fp16 dequantise + library GEMM there, fused GEMV here: ~1e-3 relative on a logit), so a step whose two
                        # best logits are closer than that can go either way — on a random-init model (near-uniform logits) that happens now and then
                        HQQLinear.set_backend(HQQBackend.PYTORCH_FORWARD)
                        try:
                            with torch.no_grad():
                                lg = ref_model(torch.cat([ids, want[:, ids.shape[1]:ids.shape[1] + n_same]], dim=1)).logits[0, -1].float()
                        finally:
                            HQQLinear.set_backend(HQQBackend.HIP)
                        top = torch.topk(lg, 2)
                        out["end_to_end"]["identity_check"]["first_difference"] = {
                            "at_token": n_same, "reference_top2_tokens": top.indices.tolist(), "reference_top2_logits": [round(float(v), 5) for v in top.values],
                            "reference_top2_gap": round(float(top.values[0] - top.values[1]), 6), "ours_picked": int(got[0, ids.shape[1] + n_same]),
                            "fp16_ulp_at_that_logit": float(torch.finfo(torch.float16).eps * abs(float(top.values[0])))}
                except Exception as e:
                    out["end_to_end"]["identity_check"] = {"error": repr(e)}
                del ref_model
                torch.cuda.empty_cache()
                # the same loop with round 4's separate glue kernels (same box, same model): what the folding buys
                try:
                    dec1 = GraphedGreedyDecoder(model, max_cache_len=256, glue="kernels")
                    r1 = dec1.benchmark(ids, new_tokens=64, warmup=8)
                    out["end_to_end"]["with_separate_glue_kernels"] = {"tok_s": round(r1["tok_s"], 2), "ms_per_token": round(r1["ms_per_token"], 4)}
                    del dec1
                except Exception as e:
                    out["end_to_end"]["with_separate_glue_kernels"] = {"error": repr(e)}
                # the same loop with the decode-attention kernel in place of HF's SDPA call (opt-in: within rounding of SDPA, not bit-identical to it)
                try:
                    dec2 = GraphedGreedyDecoder(model, max_cache_len=256, attention="hip")
                    r2 = dec2.benchmark(ids, new_tokens=64, warmup=8)
                    same2 = None
                    if want_new is not None:   # reported, not relied on: this attention is within rounding of SDPA, its tokens are not identical by construction
                        got2 = dec2.generate(ids, 8, use_graph=True)
                        same2 = int((got2[0, ids.shape[1]:] == want_new).to(torch.int32).cumprod(0).sum())
                    out["end_to_end"]["with_decode_attention_kernel"] = {
                        "tok_s": round(r2["tok_s"], 2), "ms_per_token": round(r2["ms_per_token"], 4), "linear_stack_share": round(lin_ms / r2["ms_per_token"], 3),
                        "identical_prefix_of_8_vs_PYTORCH_FORWARD": same2,
                        "note": "GraphedGreedyDecoder(attention='hip'): hqq_hip_rope_attn_decode (csrc/block.hip: rotary + cache write + one-query attention in one launch) instead of "
                                "rope_cache + F.scaled_dot_product_attention in every block; "
                                "teacher-forced logits within 5e-3 of the default step's (tests/test_model_gpu.py); the headline tok_s above is the token-identical default"}
                    del dec2
                    # the static cache's length: HF's attention function attends over the whole cache behind a mask, the kernel over pos + 1 keys
                    longc = {}
                    for mode in ("sdpa", "hip"):
                        d3 = GraphedGreedyDecoder(model, max_cache_len=2048, attention=mode)
                        r3 = d3.benchmark(ids, new_tokens=32, warmup=4)
                        longc[mode] = {"tok_s": round(r3["tok_s"], 2), "ms_per_token": round(r3["ms_per_token"], 4)}
                        del d3
                    # long context: 3000 prompt tokens in a 4096-position cache, decode at positions 3008..
                    try:
                        ids_long = torch.randint(0, 32000, (1, 3000), device=dev, generator=gx)
                        lp = {}
                        for mode in ("sdpa", "hip"):
                            d4 = GraphedGreedyDecoder(model, max_cache_len=4096, attention=mode)
                            r4 = d4.benchmark(ids_long, new_tokens=24, warmup=4)
                            lp[mode] = {"tok_s": round(r4["tok_s"], 2), "ms_per_token": round(r4["ms_per_token"], 4)}
                            del d4
                        out["end_to_end"]["at_position_3000"] = {"default_sdpa": lp["sdpa"], "with_decode_attention_kernel": lp["hip"],
                                                                 "note": "HF's attention function runs one workgroup per head over the cache bucket above the position (3072 keys); the kernel shares a head's "
                                                                         "3000 keys out over 8 workgroups (hqq_hip_rope_attn_decode, splits = 8) and is at the cache's HBM read time"}
                    except Exception as e:
                        out["end_to_end"]["at_position_3000"] = {"error": repr(e)}
                    out["end_to_end"]["max_cache_len"] = 256
                    out["end_to_end"]["at_max_cache_len_2048"] = {"default_sdpa": longc["sdpa"], "with_decode_attention_kernel": longc["hip"],
                                                                  "note": "same prompt and positions, only the static cache is longer"}
                except Exception as e:
                    out["end_to_end"]["with_decode_attention_kernel"] = {"error": repr(e)}
                del model, dec
                torch.cuda.empty_cache()
            except Exception as e:
                out["end_to_end"] = {"error": repr(e)}
        # one rank's share of the 8-way column-sharded 70B stack (BASELINE.json configs[4]) on THIS GPU alone: the compute side of the scaling
        # curve DESIGN.md section 6 expects, as a driver-run number.  No exchange, no second GPU: NOT a multi-GPU measurement.  Two plans
        # (hqq_amd.shard.plan_exchange_groups): every exchange group sharded, and the adaptive plan that REPLICATES groups too small to shard.
        if nbits == 4 and os.environ.get("HQQ_BENCH_SHARD8", "1") != "0":
            try:
                from hqq_amd import shard as shard_mod
                torch.cuda.empty_cache()
                P8 = 8

                def rank_compute(BLK, nb_, plan, m_, seed0):
                    """one rank's launches of a block list under `plan` (per exchange group: sharded -> N / 8 columns, replicated-small -> the whole layer)"""
                    whole = {n: pl == "replicated-small" for grp, pl in zip(EXCHANGE_GROUPS, plan) for n in grp}
                    sb = [{name: make_layer(ops, name, N_ if whole[name] else N_ // P8, K_, 4, dev, seed=seed0 + 16 * b_ + i, random_codes=True, cd=cd)
                           for i, (name, N_, K_) in enumerate(BLK)} for b_ in range(nb_)]
                    xs8 = {K_: torch.randn(m_, K_, device=dev, generator=gx).to(cd) for K_ in sorted({K_ for _, _, K_ in BLK})}
                    o8 = {grp: [torch.empty(m_, sb[0][n].N, device=dev, dtype=cd) for n in grp] for grp in EXCHANGE_GROUPS}

                    def step8():
                        for blk in sb:
                            for grp in EXCHANGE_GROUPS:
                                Ls = [blk[n] for n in grp]
                                ops.gemv_grouped(xs8[Ls[0].K], [(L.Wq, L.scale, L.zero, None, L.N) for L in Ls], Ls[0].K, 64, 4, outs=o8[grp], opts=group_opts(Ls))
                    r8, _ = _graphed(step8, use_graph, rank)
                    w8, d8 = _timed(r8, max(5, a.steps // 2), 3)
                    rank_bytes = nb_ * sum(gemv_bytes(N_ if whole[name] else N_ // P8, K_, 4, m_) for name, N_, K_ in BLK)
                    del r8, sb
                    torch.cuda.empty_cache()
                    return w8, d8, rank_bytes

                res8 = {}
                for tag, BLK, nb_ in (("llama2-70b", LLAMA2_70B_BLOCK, N_BLOCKS_70B), ("llama2-7b", LLAMA2_7B_BLOCK, N_BLOCKS_7B)):
                    gb = [sum(wq_bytes(N_, K_, 4) for name, N_, K_ in BLK if name in grp) for grp in EXCHANGE_GROUPS]
                    plans = {"sharded": ["sharded"] * len(EXCHANGE_GROUPS), "adaptive": shard_mod.plan_exchange_groups(gb, P8)}
                    rows = []
                    for m_ in (1, 32):
                        full_bytes = nb_ * sum(gemv_bytes(N_, K_, 4, m_) for _, N_, K_ in BLK)
                        for pname, plan in plans.items():
                            if pname == "adaptive" and plan == plans["sharded"]:
                                rows.append({"bs": m_, "plan": "adaptive", "same_as": "sharded"})
                                continue
                            w8, d8, rb = rank_compute(BLK, nb_, plan, m_, 50000)
                            n_l = nb_ * len(EXCHANGE_GROUPS)
                            rows.append({"bs": m_, "plan": pname, "groups": dict(zip(["|".join(g_) for g_ in EXCHANGE_GROUPS], plan)),
                                         "exchange_points_per_block": sum(1 for pl in plan if pl == "sharded"),
                                         "ms_per_step": round(w8 * 1e3, 4), "launches_per_step": n_l, "avg_launch_us": round(d8 / n_l * 1e6, 3),
                                         "rank_GB_s": round(rb / w8 / 1e9, 1), "roofline_frac_of_this_gpu": round(rb / d8 / 1e9 / HBM_PEAK_GBS, 4),
                                         "if_eight_ranks_computed_like_this_and_exchanged_for_free": {
                                             "tok_s": round(m_ / w8, 1), "frac_of_8_gpu_roofline": round(full_bytes / w8 / 1e9 / (P8 * HBM_PEAK_GBS), 4)}})
                    res8[tag] = rows
                out["shard_of_8"] = {"what": "ONE rank's compute of a linear stack column-sharded 8 ways (hqq_amd/shard.py: N / 8 output columns of every sharded layer; int4 gs=64, random codes), "
                                             "timed on this single GPU: one rank's compute, NO exchange, nothing measured on more than one GPU",
                                     "replicate_below_bytes": shard_mod.replicate_below_bytes(P8), "stacks": res8,
                                     "note": "plan 'sharded': every exchange group (q|k|v, o, gate|up, down) column-sharded, 4 exchange points per block.  plan 'adaptive' "
                                             "(hqq_amd.shard.plan_exchange_groups): a group whose packed bytes are below replicate_below_bytes is held and computed WHOLE by every rank — "
                                             "more bytes per launch, one exchange point fewer.  The threshold comes from the single-GPU launch model t = 3.8 us + bytes / 7.7 TB/s and an "
                                             "ASSUMED 4 us per exchange point; the exchange has never run over xGMI (DESIGN.md section 6)"}
            except Exception as e:
                out["shard_of_8"] = {"error": repr(e)}
        out["legs"] = legs

    if rank == 0:
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(nbits)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def _decode_kernel_name(nbits, M, dtype, mode):
    """the kernel hqq_hip_gemv dispatches the bench's launches to (gemv.hip / skinny.hip / gemv3*.hip)"""
    if nbits == 3:
        return ("hqq::gemv_w3s_kernel<3, %d, gs64, exact, %s> (3-bit stream layout)" % (M, "bf16" if dtype == "bf16" else "f16")) if M < 5 else \
               f"hqq::skinny_f16_kernel<3 (stream layout), {(M + 15) // 16}, {'bf16' if dtype == 'bf16' else 'f16'}>"
    if M >= 5:
        return f"hqq::skinny_f16_kernel<{nbits}, {(M + 15) // 16}, {'bf16' if dtype == 'bf16' else 'f16'}>"
    arith = "factored" if mode == "factored" else ("exact, three-op rebuild" if mode == "exact" and dtype == "f16" else "exact")
    return f"hqq::gemv_f16_kernel<{nbits}, {M}, gs64, {arith}, {'bf16' if dtype == 'bf16' else 'f16'}>"


def _solver_valu():
    """instruction census of the solver's fast path (profiles/solver_valu.json, written by tools/solver_valu_count.py from the ISA); None if absent"""
    try:
        with open(os.path.join(ROOT, "profiles", "solver_valu.json")) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def _pmc_dense_busy():
    """MFMA-busy share and effective clock of the dense GEMM from the committed PMC pass (profiles/pmc_traffic.json names the commit); None if absent."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(p) as f:
            d = json.load(f)
        return {"dense_gemm_kernel_8192x4096x4096": d["dense_gemm_8192x4096x4096"]["in_tree"], "commit": d.get("commit")}
    except (OSError, ValueError, KeyError):
        return None


def _pmc_traffic(nbits):
    """HBM bytes per launch from the committed PMC pass (profiles/), already corrected as MI355X_MICROARCH.md prescribes; None if absent."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(p) as f:
            return json.load(f).get(f"int{nbits}_decode_bytes_per_launch")
    except (OSError, ValueError):
        return None


if __name__ == "__main__":
    main()
