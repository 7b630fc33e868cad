import torch, sys
sys.path.insert(0, '.')
from hqq_amd import ops
N, K = 11008, 4096
for nb in (8, 4, 2, 1):
    U = torch.randint(0, 2 ** nb, (N * K // 64, 64), dtype=torch.uint8, device="cuda")
    for _ in range(3): P = ops.pack(nb, U)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): P = ops.pack(nb, U)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    by = U.numel() + P.numel()
    print(f"pack {nb}-bit: {us:.1f} us  {by / us / 1e3:.0f} GB/s  {by / us / 1e3 / 8000:.3f}", flush=True)
