#!/usr/bin/env python3
"""hqq_hip_attn_decode against the position and the number of workgroups per head, cache of 4096 (development aid; needs an MI355X)."""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from hqq_amd import ops  # noqa: E402

H, L, D = 32, 4096, 128
g = torch.Generator(device="cuda").manual_seed(0)
q = torch.randn(1, H, 1, D, device="cuda", generator=g).half()
kc = torch.randn(H, L, D, device="cuda", generator=g).half()
vc = torch.randn(H, L, D, device="cuda", generator=g).half()
out = torch.empty(H * D, dtype=torch.float16, device="cuda")


def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(n): fn()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for pos in (15, 100, 255, 700, 1500, 3000, 4095):
    p = torch.tensor([pos], device="cuda")
    row = []
    for S in (1, 2, 4, 8, 16):
        ws = ops.attn_workspace("cuda", H, D, S)
        row.append(f"S={S}: {t(lambda: ops.attn_decode(q, kc, vc, p, out, D ** -0.5, splits=S, workspace=ws)):6.2f}")
    print(f"position {pos:5d}  " + "  ".join(row) + "  us")
