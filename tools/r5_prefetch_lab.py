#!/usr/bin/env python3
"""lab: the 7B int4 decode stack (128 dependent launches, graph replay) with every launch's DEAD units prefetching the first 2 KiB of the rows the
next launch starts on (tools/libhqq_hip_pf.so, built with -DGV_LAB_PREFETCH), against the same library with the prefetch switched off.
    HQQ_AMD_LIB=$PWD/tools/libhqq_hip_pf.so python tools/r5_prefetch_lab.py"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hqq_amd import ops, _C
L = _C.lib()
has_pf = hasattr(L, "hqq_lab_set_prefetch")
if has_pf:
    L.hqq_lab_set_prefetch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    L.hqq_lab_set_prefetch.restype = None
g = torch.Generator(device="cuda").manual_seed(0)
BLOCK = [("q", 4096, 4096), ("k", 4096, 4096), ("v", 4096, 4096), ("o", 4096, 4096), ("gate", 11008, 4096), ("up", 11008, 4096), ("down", 4096, 11008)]
GROUPS = [("q", "k", "v"), ("o",), ("gate", "up"), ("down",)]
def layer(N, K):
    R = N * K // 64
    Wq = torch.randint(0, 256, (R // 2, 64), dtype=torch.uint8, device="cuda", generator=g)
    s = (torch.rand(R, 1, device="cuda", generator=g) * 0.004 + 0.001).half()
    z = (torch.rand(R, 1, device="cuda", generator=g) * 15).round().half()
    return (Wq, s, z, None, N)
blocks = [{n: layer(N, K) for n, N, K in BLOCK} for _ in range(32)]
KOF = {n: K for n, _, K in BLOCK}
sub = all(ops.meta_scalable(L_[1], L_[2], L_[4], KOF[n], 64, 4) for b in blocks for n, L_ in b.items())
opts = ops.OPT_META_SCALABLE if sub else 0
xs = {K: torch.randn(1, K, device="cuda", generator=g).half() for K in (4096, 11008)}
outs = {grp: [torch.empty(1, blocks[0][n][4], dtype=torch.float16, device="cuda") for n in grp] for grp in GROUPS}
seq = [(b, grp) for b in range(32) for grp in GROUPS]
def step(pf):
    for i, (b, grp) in enumerate(seq):
        if has_pf:
            if pf:
                nb, ngrp = seq[(i + 1) % len(seq)]
                nxt = blocks[nb][ngrp[0]]
                L.hqq_lab_set_prefetch(nxt[0].data_ptr(), nxt[4] // 2, KOF[ngrp[0]])
            else:
                L.hqq_lab_set_prefetch(None, 0, 0)
        Ls = [blocks[b][n] for n in grp]
        ops.gemv_grouped(xs[KOF[grp[0]]], Ls, KOF[grp[0]], 64, 4, outs=outs[grp], opts=opts)
def timeit(pf):
    step(pf); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        step(pf)
    for _ in range(5): gr.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): gr.replay()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    return best
want = None
print("lib", os.environ.get("HQQ_AMD_LIB", "default"), "sub", sub, "lab symbol", has_pf)
for rep in range(3):
    a = timeit(False)
    ref = [t.clone() for t in outs[GROUPS[-1]]]
    b = timeit(True) if has_pf else float("nan")
    same = all(torch.equal(t, r) for t, r in zip(outs[GROUPS[-1]], ref))
    print(f"  no prefetch {a:.4f} ms ({3647750144 / a / 1e6 / 8000:.3f})   dead-unit prefetch {b:.4f} ms ({3647750144 / b / 1e6 / 8000:.3f})   outputs equal {same}", flush=True)
