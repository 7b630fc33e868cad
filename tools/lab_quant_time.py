#!/usr/bin/env python3
"""lab: solver time per layer (development aid; needs an MI355X)"""
import sys, torch
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from hqq_amd import ops
for N, K in ((4096, 4096), (11008, 4096)):
    W = (torch.randn(N, K, generator=torch.Generator().manual_seed(0)) * 0.02).half().cuda()
    for _ in range(2): ops.quantize(W, nbits=4, group_size=64, round_zero=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.quantize(W, nbits=4, group_size=64, round_zero=True)
    e1.record(); torch.cuda.synchronize()
    print(f"{N}x{K}: {e0.elapsed_time(e1) / 5:.3f} ms", end="  ")
print()
