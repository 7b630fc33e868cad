#!/usr/bin/env python3
"""Which kernels one decode-shaped SDPA call launches, by how it is called (development aid; run under rocprofv3 --kernel-trace).
    python tools/sdpa_probe.py VARIANT    (a: [1,1,1,L] additive mask, b: [1,H,1,L] contiguous mask, c: b + pre-made contiguous q)"""
import sys

import torch
import torch.nn.functional as F

v = sys.argv[1]
H, L, D = 32, 256, 128
torch.manual_seed(0)
q = torch.randn(1, H, 1, D, device="cuda", dtype=torch.float16)
k = torch.randn(1, H, L, D, device="cuda", dtype=torch.float16)
vv = torch.randn(1, H, L, D, device="cuda", dtype=torch.float16)
m1 = torch.zeros(1, 1, 1, L, device="cuda", dtype=torch.float16)
m1[..., 100:] = float("-inf")
mH = m1.expand(1, H, 1, L).contiguous()
outs = []
for _ in range(20):
    if v == "a":
        o = F.scaled_dot_product_attention(q, k, vv, attn_mask=m1, dropout_p=0.0, scale=D ** -0.5, is_causal=False)
    else:
        o = F.scaled_dot_product_attention(q, k, vv, attn_mask=mH, dropout_p=0.0, scale=D ** -0.5, is_causal=False)
    o = o.transpose(1, 2).contiguous()
    outs.append(o)
torch.cuda.synchronize()
ref = F.scaled_dot_product_attention(q, k, vv, attn_mask=m1, dropout_p=0.0, scale=D ** -0.5).transpose(1, 2).contiguous()
print("variant", v, "equal to variant a:", torch.equal(outs[-1], ref))
