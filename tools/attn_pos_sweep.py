#!/usr/bin/env python3
"""One-query attention against the position, cache of 4096 (development aid; needs an MI355X): torch's SDPA over a bucket of the cache (what the
default fused step calls) vs hqq_hip_attn_decode.   python tools/attn_pos_sweep.py"""
import sys

import torch
import torch.nn.functional as F

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
    b = 256
    while b < pos + 1: b *= 2
    mask = torch.zeros(1, 1, 1, b, dtype=torch.float16, device="cuda"); mask[..., pos + 1:] = float("-inf")
    ks, vs = kc[:, :b].unsqueeze(0), vc[:, :b].unsqueeze(0)
    ts = t(lambda: F.scaled_dot_product_attention(q, ks, vs, attn_mask=mask, scale=D ** -0.5))
    th = t(lambda: ops.attn_decode(q, kc, vc, p, out, D ** -0.5))
    print(f"position {pos:5d}: SDPA over a bucket of {b:5d}: {ts:7.2f} us    attn_decode: {th:7.2f} us")
