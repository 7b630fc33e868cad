#!/usr/bin/env python3
"""lab: the persistent decode engine against the per-launch kernels (run on the GPU box).
1. a chain of stages with REAL dependencies (x of stage s+1 is an output buffer of stage s), small and ragged shapes
2. the 7B stack: outputs vs per-launch gemv_grouped, determinism, time per token (graph replay) both ways"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from hqq_amd import ops

dev = torch.device("cuda")


def qlayer(N, K, nbits, seed, gain=1.0):
    g = torch.Generator(device=dev).manual_seed(seed)
    W = (torch.randn(N, K, device=dev, generator=g) * (gain / K ** 0.5)).half()
    Wq, s, z = ops.quantize(W, nbits=nbits, group_size=64, round_zero=(nbits == 4))
    return Wq, s.half(), z.half()


def chain_case(nbits, dims, grid, sub, bias=False):
    """dims: list of (K, [N...]) ; stage s+1 reads the LAST output of stage s (its N must equal the next K)"""
    torch.manual_seed(1)
    x0 = torch.randn(1, dims[0][0], device=dev).half()
    layers, stages, ys = [], [], []
    x = x0
    for si, (K, Ns) in enumerate(dims):
        Ls = []
        for j, N in enumerate(Ns):
            Wq, s, z = qlayer(N, K, nbits, 100 * si + j)
            b = (torch.randn(N, device=dev) * 0.1).half() if bias else None
            y = torch.full((1, N), float("nan"), device=dev, dtype=torch.float16)
            Ls.append((Wq, s, z, b, N, y))
        stages.append((x, Ls))
        layers.append(Ls)
        x = Ls[-1][5]
    plan = ops.DecodePlan(stages, nbits, opts=ops.OPT_META_SCALABLE if sub else 0, grid=grid)
    plan.run()
    torch.cuda.synchronize()
    st = plan.status()
    # reference: the per-launch kernels, stage after stage
    xr = x0
    worst = 0.0
    for si, (K, Ns) in enumerate(dims):
        Ls = layers[si]
        outs = ops.gemv_grouped(xr, [(L[0], L[1], L[2], L[3], L[4]) for L in Ls], K, 64, nbits, opts=0)
        for L, o in zip(Ls, outs):
            got = L[5].float()
            ref = o.float()
            err = float(torch.nan_to_num((got - ref).abs() / (1e-3 + 1e-3 * ref.abs()), nan=1e9).max())
            worst = max(worst, err)
        xr = outs[-1]
    print(f"chain nbits={nbits} grid={grid} sub={int(sub)} bias={int(bias)} dims={dims}: status={st} worst_err/tol={worst:.3f}", flush=True)
    return st == 0 and worst <= 1.0


def stack_case(nbits, blocks, sub):
    BLOCK = [("q", 4096, 4096), ("k", 4096, 4096), ("v", 4096, 4096), ("o", 4096, 4096), ("gate", 11008, 4096), ("up", 11008, 4096), ("down", 4096, 11008)]
    GROUPS = [("q", "k", "v"), ("o",), ("gate", "up"), ("down",)]
    t0 = time.time()
    Ls = []
    for b in range(blocks):
        blk = {}
        for i, (name, N, K) in enumerate(BLOCK):
            blk[name] = (N, K) + qlayer(N, K, nbits, 16 * b + i, gain=0.02 * K ** 0.5)
        Ls.append(blk)
    torch.manual_seed(1)
    xs = {K: torch.randn(1, K, device=dev).half() for K in (4096, 11008)}
    dimN = {n: N for n, N, _ in BLOCK}
    out_e = {g: [torch.zeros(1, dimN[n], device=dev, dtype=torch.float16) for n in g] for g in GROUPS}   # engine outputs (shared by all blocks, as in bench.py)
    out_l = {g: [torch.zeros(1, dimN[n], device=dev, dtype=torch.float16) for n in g] for g in GROUPS}
    stages = []
    for blk in Ls:
        for g in GROUPS:
            K = blk[g[0]][1]
            stages.append((xs[K], [(blk[n][2], blk[n][3], blk[n][4], None, blk[n][0], out_e[g][j]) for j, n in enumerate(g)]))
    plan = ops.DecodePlan(stages, nbits, opts=ops.OPT_META_SCALABLE if sub else 0)
    print(f"stack nbits={nbits} blocks={blocks}: setup {time.time()-t0:.1f}s, plan {plan.nbytes} B", flush=True)

    lopts = 0

    def launches():
        for blk in Ls:
            for g in GROUPS:
                K = blk[g[0]][1]
                ops.gemv_grouped(xs[K], [(blk[n][2], blk[n][3], blk[n][4], None, blk[n][0]) for n in g], K, 64, nbits, outs=out_l[g], opts=lopts)

    lopts = ops.OPT_META_SCALABLE if sub else 0
    plan.run(); launches(); torch.cuda.synchronize()
    st = plan.status()
    worst = 0.0
    for g in GROUPS:   # the buffers hold the LAST block's outputs
        for a, b in zip(out_e[g], out_l[g]):
            worst = max(worst, float(torch.nan_to_num((a.float() - b.float()).abs() / (1e-3 + 1e-3 * b.float().abs()), nan=1e9).max()))
    first = {g: [t.clone() for t in out_e[g]] for g in GROUPS}
    same = True
    for _ in range(5):
        plan.run()
    torch.cuda.synchronize()
    for g in GROUPS:
        for a, b in zip(out_e[g], first[g]):
            same = same and torch.equal(a, b)
    print(f"   status={st} worst_err/tol vs launches={worst:.3f} reproducible={same}", flush=True)

    def timed(fn, name):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fn()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            fn()
        for _ in range(5): gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): gr.replay()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 30
        nb = blocks * sum((N * K * nbits // 8) + 4 * (N * K // 64) + 2 * K + 2 * N for _, N, K in BLOCK)
        print(f"   {name}: {ms*1e3:.1f} us/token  {nb/ms/1e6:.0f} GB/s  frac {nb/ms/1e6/8000:.3f}", flush=True)
        return ms

    timed(launches, "launches")
    timed(plan.run, "engine  ")
    print(f"   status after timing={plan.status()}", flush=True)


if __name__ == "__main__":
    ok = True
    for sub in (False, True):
        ok &= chain_case(4, [(1024, [512, 1024]), (1024, [2048]), (2048, [1024])], grid=8, sub=sub)
        ok &= chain_case(4, [(1024, [512, 1024]), (1024, [2048]), (2048, [1024])], grid=0, sub=sub)
        ok &= chain_case(4, [(1280, [64, 34, 1152]), (1152, [640]), (640, [128, 128, 128, 256])], grid=5, sub=sub, bias=True)
        ok &= chain_case(2, [(1024, [512, 1024]), (1024, [2048]), (2048, [1024])], grid=16, sub=sub)
        ok &= chain_case(8, [(1024, [512, 1024]), (1024, [256])], grid=0, sub=sub, bias=True)
    print("CHAINS", "OK" if ok else "FAILED", flush=True)
    if len(sys.argv) > 1 and sys.argv[1] == "chains":
        sys.exit(0 if ok else 1)
    stack_case(4, 32, True)
    stack_case(4, 32, False)
    stack_case(2, 32, True)
