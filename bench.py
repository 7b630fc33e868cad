#!/usr/bin/env python3
"""bench.py — the contract benchmark of the HQQ forward hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload decode|prefill] [--nbits 4|2|3]

Workload (BASELINE.json configs[1], the configuration the metric is quoted on):
  one *step* = one token (bs=1) through every quantised linear of a Llama-2-7B decoder stack —
  32 blocks x {q,k,v,o 4096x4096; gate,up 11008x4096; down 4096x11008}, nbits=4 group_size=64 axis=1,
  224 fused unpack->dequantize->GEMV launches streaming 3.65 GB of packed weights + meta from HBM
  (far beyond the 256 MiB Infinity Cache, so every step is HBM traffic).  Weights are synthetic
  N(0, 0.02^2) fp16 tensors quantised on the GPU by the HIP half-quadratic solver before timing.
  `--workload prefill` runs configs[2] instead: M tokens (default 8192 = 4 x 2048) through one block's
  seven linears on the MFMA dequant-GEMM.

Multi-GPU (launched by torch.distributed.run, one rank per GPU, RCCL): output-column shard.  Rank r
owns output rows [r*N, (r+1)*N) of an N*P-row layer (weak scaling: the per-GPU shard is exactly the
N=1 workload), x is replicated, and every exchange point (after q/k/v, o, gate/up, down) is one RCCL
all-gather of the fp16 shard outputs over xGMI.  `value` is the whole-job rate over all ranks.

Prints ONE JSON line on rank 0.  `value` = algorithmic GB/s streamed by the whole job (SURVEY.md §8d
bytes: W_q + scale + zero + x + y); tok/s is reported next to it.
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
# exchange points of a column-sharded block: outputs that are consumed together are gathered together
EXCHANGE_GROUPS = [("q", "k", "v"), ("o",), ("gate", "up"), ("down",)]
N_BLOCKS = 32


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
    ap.add_argument("--workload", default="decode", choices=["decode", "prefill"])
    ap.add_argument("--nbits", type=int, default=4)
    ap.add_argument("--bs", type=int, default=1, help="decode batch (rows of x), 1..64 (int4/int2/int8 from 5 up, fp16 or bf16; 1..4 for int3)")
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16"], help="compute dtype (bf16: decode only)")
    ap.add_argument("--prefill-tokens", type=int, default=8192)
    ap.add_argument("--blocks", type=int, default=N_BLOCKS)
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--no-group", action="store_true", help="one launch per layer instead of one per exchange group (q/k/v, o, gate/up, down)")
    ap.add_argument("--library-gemm", action="store_true", help="prefill: dequantise kernel + hipBLASLt GEMM instead of the fused MFMA kernel")
    ap.add_argument("--streams", type=int, default=1, help="study mode: deal the launches over this many parallel graph branches (ignores the decoder's dependency chain)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--random-codes", action="store_true", help="skip the solver: random packed codes + meta (faster setup)")
    ap.add_argument("--gemv-mode", default="exact", choices=["exact", "factored", "sub"],
                    help="exact: reference-identical weights (default); factored: fp32 affine map factored out of the dot product")
    return ap.parse_args()


class Layer:
    __slots__ = ("name", "N", "K", "Wq", "scale", "zero")


def make_layer(ops, name, N, K, nbits, dev, seed, random_codes, cd=torch.float16):
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
        return L
    W = (torch.randn(N, K, device=dev, generator=g) * 0.02).half()
    Wq, s, z = ops.quantize(W, nbits=nbits, group_size=64, round_zero=(nbits == 4))
    # HQQLinear.cuda(): meta is cast to compute_dtype (quantize.py:515-583)
    L.Wq, L.scale, L.zero = Wq, s.to(cd), z.to(cd)
    return L


def cpu_baseline(nbits):
    """The oracle's dequantize+matmul (the per-call work of HQQBackend.PYTORCH) on this box's host cores, bounded to ~10 s."""
    import numpy as np
    from oracle import hqq_oracle as orc
    N = K = 4096
    rng = np.random.default_rng(0)
    R = N * K // 64
    U = rng.integers(0, 2 ** nbits, size=(R, 64), dtype=np.uint8)
    P = orc.pack(nbits, U)
    s = orc.to_cd((rng.random((R, 1), dtype=np.float32) * 0.004 + 0.001), orc.F16)
    z = orc.to_cd((rng.random((R, 1), dtype=np.float32) * (2 ** nbits - 1)), orc.F16)
    x = orc.to_cd(rng.standard_normal((1, K), dtype=np.float32), orc.F16)
    orc.forward(nbits, P, s, z, None, x, N, K, 64, orc.F16)   # warm-up
    t0 = time.perf_counter()
    reps = 0
    while True:
        orc.forward(nbits, P, s, z, None, x, N, K, 64, orc.F16)
        reps += 1
        el = time.perf_counter() - t0
        if el > 10.0 or reps >= 200:
            break
    t = el / reps
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    nb = gemv_bytes(N, K, nbits)
    return {"value": round(nb / t / 1e9, 4), "unit": "GB/s", "cores": cores, "kind": "port",
            "tok_s_7b_stack_equiv": round(1.0 / (t * (sum(gemv_bytes(n, k, nbits) for _, n, k in LLAMA2_7B_BLOCK) * N_BLOCKS / nb)), 4),
            "ms_per_layer_call": round(t * 1e3, 3),
            "sample": f"oracle/hqq_oracle.c forward (unpack+dequantize+matmul, OpenMP) of one 4096x4096 int{nbits} gs=64 layer, bs=1, "
                      f"{reps} calls in {el:.1f} s"}


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
    ops.set_gemv_mode({"factored": ops.GEMV_FACTORED, "exact": ops.GEMV_EXACT, "sub": 2}[a.gemv_mode])
    nbits = a.nbits
    decode = a.workload == "decode"
    cd = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    assert a.dtype == "f16" or (decode and (a.bs >= 5 or nbits in (4, 2))), "bf16: decode only; bs <= 4 needs int4 / int2 (bs 5..64: int8/4/2)"
    M = a.bs if decode else a.prefill_tokens
    nblocks = a.blocks if decode else 1

    # ---- the resident stack: every layer distinct in HBM ----
    t_setup = time.perf_counter()
    blocks = []
    for b in range(nblocks):
        blk = {}
        for i, (name, N, K) in enumerate(LLAMA2_7B_BLOCK):
            blk[name] = make_layer(ops, name, N, K, nbits, dev, seed=1000 * rank + 16 * b + i, random_codes=a.random_codes, cd=cd)
        blocks.append(blk)
    gx = torch.Generator(device=dev).manual_seed(1)       # x is replicated: same seed on every rank
    xs = {K: torch.randn(M, K, device=dev, generator=gx).to(cd) for K in (4096, 11008)}
    # per exchange group: local outputs [len(group), M, N] and, for P > 1, the gathered [P, len(group), M, N]
    out_local, out_full = {}, {}
    for grp in EXCHANGE_GROUPS:
        N = dict((n, nn) for n, nn, _ in LLAMA2_7B_BLOCK)[grp[0]]
        out_local[grp] = torch.empty(len(grp), M, N, device=dev, dtype=cd)
        if world > 1:
            out_full[grp] = torch.empty(world * len(grp) * M, N, device=dev, dtype=cd)   # rank-major concatenation
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup

    grouped = decode and not a.no_group and nbits in (4, 3, 2, 8, 1)

    # --streams S > 1 (study mode, not the headline): the step's launches are dealt over S parallel graph branches, i.e. the
    # dependency chain q|k|v -> o -> gate|up -> down of a real decoder is NOT modelled and consecutive launches may overlap;
    # it shows what the kernels sustain once launch boundaries are hidden.  Outputs then need one buffer per launch in flight.
    S = max(1, a.streams)
    branch_streams = [torch.cuda.Stream() for _ in range(S - 1)] if S > 1 else []
    if S > 1:
        out_local_s = [{g: torch.empty_like(t) for g, t in out_local.items()} for _ in range(S)]

    def step():
        if S > 1:
            main = torch.cuda.current_stream()
            for st in branch_streams:
                st.wait_stream(main)
            i = 0
            for blk in blocks:
                for grp in EXCHANGE_GROUPS:
                    b = i % S
                    i += 1
                    with torch.cuda.stream(main if b == 0 else branch_streams[b - 1]):
                        ol = out_local_s[b][grp]
                        Ls = [blk[name] for name in grp]
                        ops.gemv_grouped(xs[Ls[0].K], [(L.Wq, L.scale, L.zero, None, L.N) for L in Ls], Ls[0].K, 64, nbits,
                                         outs=[ol[j] for j in range(len(grp))])
            for st in branch_streams:
                main.wait_stream(st)
            return
        for blk in blocks:
            for grp in EXCHANGE_GROUPS:
                ol = out_local[grp]
                if grouped:   # q/k/v and gate/up read the same x: one launch per exchange group
                    Ls = [blk[name] for name in grp]
                    ops.gemv_grouped(xs[Ls[0].K], [(L.Wq, L.scale, L.zero, None, L.N) for L in Ls], Ls[0].K, 64, nbits,
                                     outs=[ol[j] for j in range(len(grp))])
                else:
                    for j, name in enumerate(grp):
                        L = blk[name]
                        ops.forward(xs[L.K], L.Wq, L.scale, L.zero, None, L.N, L.K, 64, nbits, out=ol[j], fused=(not a.library_gemm))
                if world > 1:
                    dist.all_gather_into_tensor(out_full[grp], ol.view(len(grp) * M, -1))

    # ---- graph capture (launch-bound inner loop -> one hipGraph replay per step) ----
    # hipGraph capture of the whole step (incl. the RCCL all-gathers for N > 1; torch captures NCCL collectives).  Backends
    # that stage through the host (gloo debug mode) cannot be captured: launch eagerly there.
    use_graph = not a.no_graph and os.environ.get("HQQ_BENCH_GRAPH", "1") != "0" and (world == 1 or dist.get_backend() == "nccl")
    graph = None
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()          # eager once: communicator set-up, lazy module loads
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    if use_graph:
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                step()
            graph.replay()
            torch.cuda.synchronize()
        except Exception as e:   # capture unsupported for some op: run eagerly, say so
            if rank == 0:
                print(f"[bench] graph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            graph = None
            torch.cuda.synchronize()
    run = graph.replay if graph is not None else step
    mode_name = {ops.GEMV_EXACT: "exact", ops.GEMV_FACTORED: "factored", 2: "exact-sub"}[ops.get_gemv_mode()]

    for _ in range(a.warmup):
        run()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(a.steps):
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

    # ---- accounting ----
    launches_per_step = nblocks * (len(EXCHANGE_GROUPS) if grouped else len(LLAMA2_7B_BLOCK))
    bytes_per_step_rank = nblocks * sum(gemv_bytes(N, K, nbits, M) for _, N, K in LLAMA2_7B_BLOCK)
    flops_per_step_rank = nblocks * sum(2.0 * M * N * K for _, N, K in LLAMA2_7B_BLOCK)
    sec_per_step = elapsed / a.steps
    dev_sec_per_step = ev_elapsed / a.steps
    out = {
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(sec_per_step * 1e3, 5),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
    }
    if decode:
        gbs = world * bytes_per_step_rank / sec_per_step / 1e9
        out.update({
            "metric": f"int{nbits} gs=64 dequant-GEMV decode throughput, Llama-2-7B linear stack bs={M} (algorithmic GB/s; tok/s alongside)",
            "value": round(gbs, 2), "unit": "GB/s",
            "tok_s": round(M / sec_per_step, 2),
            "config": {"workload": f"llama2-7b linear stack ({nblocks} blocks x q,k,v,o,gate,up,down), nbits={nbits} gs=64 axis=1, bs={M} decode, "
                                   f"{'bf16' if a.dtype == 'bf16' else 'fp16'}, {launches_per_step} fused dequant-GEMV launches/step ({'q|k|v, o, gate|up, down grouped' if grouped else 'one per layer'})" + (", hipGraph replay" if graph is not None else ", eager launches") + (f", {S} parallel branches (dependency chain NOT modelled)" if S > 1 else ""),
                       "global_batch": M, "parallelism": "single-gpu" if world == 1 else f"column-shard x{world} + RCCL all-gather (weak: {world}x wider layers)",
                       "gemv_mode": mode_name, "bytes_per_step_per_gpu": bytes_per_step_rank, "setup_s": round(t_setup, 2)},
        })
        avg_launch_s = dev_sec_per_step / launches_per_step
        ach = (bytes_per_step_rank / launches_per_step) / avg_launch_s / 1e9
        out["roofline"] = {"bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                           # PMC traffic is committed for the configuration it was measured on only (bs=1 fp16, exact mode)
                           "traffic": _pmc_traffic(nbits) if (M == 1 and a.dtype == "f16" and mode_name == "exact") else None,
                           "kernel": _decode_kernel_name(nbits, M, a.dtype, mode_name), "avg_launch_us": round(avg_launch_s * 1e6, 3),
                           "bytes_per_launch": bytes_per_step_rank // launches_per_step,
                           "note": "avg launch = HIP-event time of the timed region / launches (includes inter-kernel gaps)"}
    else:
        tfl = world * flops_per_step_rank / sec_per_step / 1e12
        out.update({
            "metric": f"int{nbits} gs=64 dequant-GEMM prefill throughput, Llama-2-7B block M={M} (tok/s; TFLOP/s alongside)",
            "value": round(M / sec_per_step, 2), "unit": "tok/s (one block's 7 linears)", "tflops": round(tfl, 2),
            "config": {"workload": f"llama2-7b one block (q,k,v,o,gate,up,down), nbits={nbits} gs=64 axis=1, M={M} prefill tokens, fp16 " + ("dequantise kernel + library GEMM" if a.library_gemm else "fused MFMA dequant-GEMM"),
                       "global_batch": M, "parallelism": "single-gpu" if world == 1 else f"column-shard x{world} + RCCL all-gather"},
        })
        ach = flops_per_step_rank / dev_sec_per_step / 1e12
        out["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS, 4),
                           "traffic": None, "kernel": "hqq::dequant + hipBLASLt" if a.library_gemm else "hqq::gemm_f16_kernel"}
    if rank == 0:
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(nbits)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def _decode_kernel_name(nbits, M, dtype, mode_name):
    """the kernel hqq_hip_gemv dispatches the bench's launches to (gemv.hip / skinny.hip / gemv3*.hip)"""
    if nbits == 3:
        return "hqq::gemv3s_kernel + gemv3s_finish_kernel (launches >= 19 MB) / hqq::gemv3_f16_kernel"
    if M >= 5:
        return f"hqq::skinny_f16_kernel<{nbits}, {(M + 15) // 16}, {'bf16' if dtype == 'bf16' else 'f16'}>"
    return f"hqq::gemv_f16_kernel<{nbits}, {M}, gs64, {mode_name}, {'bf16' if dtype == 'bf16' else 'f16'}>"


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
