#!/usr/bin/env python3
"""lab: a few eager steps of the 7B stack for rocprofv3 (per-dispatch rows): `launch` = 128 gemv launches per step, `engine` = 1."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from hqq_amd import ops
mode = sys.argv[1] if len(sys.argv) > 1 else "launch"
nbits, blocks = 4, 32
dev = torch.device("cuda")
BLOCK = [("q", 4096, 4096), ("k", 4096, 4096), ("v", 4096, 4096), ("o", 4096, 4096), ("gate", 11008, 4096), ("up", 11008, 4096), ("down", 4096, 11008)]
GROUPS = [("q", "k", "v"), ("o",), ("gate", "up"), ("down",)]
def qlayer(N, K, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    W = (torch.randn(N, K, device=dev, generator=g) * 0.02).half()
    Wq, s, z = ops.quantize(W, nbits=nbits, group_size=64, round_zero=True)
    return Wq, s.half(), z.half()
Ls = [{name: (N, K) + qlayer(N, K, 16 * b + i) for i, (name, N, K) in enumerate(BLOCK)} for b in range(blocks)]
xs = {K: torch.randn(1, K, device=dev).half() for K in (4096, 11008)}
dimN = {n: N for n, N, _ in BLOCK}
out = {g: [torch.zeros(1, dimN[n], device=dev, dtype=torch.float16) for n in g] for g in GROUPS}
if mode == "engine":
    stages = []
    for blk in Ls:
        for g in GROUPS:
            K = blk[g[0]][1]
            stages.append((xs[K], [(blk[n][2], blk[n][3], blk[n][4], None, blk[n][0], out[g][j]) for j, n in enumerate(g)]))
    plan = ops.DecodePlan(stages, nbits, opts=ops.OPT_META_SCALABLE)
    step = plan.run
else:
    def step():
        for blk in Ls:
            for g in GROUPS:
                K = blk[g[0]][1]
                ops.gemv_grouped(xs[K], [(blk[n][2], blk[n][3], blk[n][4], None, blk[n][0]) for n in g], K, 64, nbits, outs=out[g], opts=ops.OPT_META_SCALABLE)
for _ in range(4):
    step()
torch.cuda.synchronize()
