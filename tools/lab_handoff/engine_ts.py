#!/usr/bin/env python3
"""lab: where a stage of the persistent engine spends its time (needs tools/libhqq_hip_lab.so: HQQ_AMD_LIB=tools/libhqq_hip_lab.so)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from hqq_amd import ops, _C

dev = torch.device("cuda")
nbits = int(sys.argv[1]) if len(sys.argv) > 1 else 4
blocks = 32
BLOCK = [("q", 4096, 4096), ("k", 4096, 4096), ("v", 4096, 4096), ("o", 4096, 4096), ("gate", 11008, 4096), ("up", 11008, 4096), ("down", 4096, 11008)]
GROUPS = [("q", "k", "v"), ("o",), ("gate", "up"), ("down",)]


def qlayer(N, K, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    W = (torch.randn(N, K, device=dev, generator=g) * 0.02).half()
    Wq, s, z = ops.quantize(W, nbits=nbits, group_size=64, round_zero=(nbits == 4))
    return Wq, s.half(), z.half()


Ls = [{name: (N, K) + qlayer(N, K, 16 * b + i) for i, (name, N, K) in enumerate(BLOCK)} for b in range(blocks)]
xs = {K: torch.randn(1, K, device=dev).half() for K in (4096, 11008)}
dimN = {n: N for n, N, _ in BLOCK}
out = {g: [torch.zeros(1, dimN[n], device=dev, dtype=torch.float16) for n in g] for g in GROUPS}
stages = []
for blk in Ls:
    for g in GROUPS:
        K = blk[g[0]][1]
        stages.append((xs[K], [(blk[n][2], blk[n][3], blk[n][4], None, blk[n][0], out[g][j]) for j, n in enumerate(g)]))
plan = ops.DecodePlan(stages, nbits, opts=ops.OPT_META_SCALABLE)
nst = len(stages)
nwg = torch.cuda.get_device_properties(0).multi_processor_count
ts = torch.zeros(nwg * nst * 8, dtype=torch.int64, device=dev)
L = _C.lib()
L.hqq_hip_lab_set_engine_ts.argtypes = [ctypes.c_void_p]
L.hqq_hip_lab_set_engine_ts(ts.data_ptr())
for _ in range(3):
    plan.run()
torch.cuda.synchronize()
ts.zero_()
plan.run()
torch.cuda.synchronize()
T = ts.cpu().numpy().reshape(nwg, nst, 8).astype(np.float64) / 100.0   # us (100 MHz)
t_start = T[:, 0, 1].min()
print(f"status={plan.status()} total = {(T[:, -1, 2].max() - T[:, 0, 0].min()):.1f} us for {nst} stages")
names = ["q|k|v", "o", "gate|up", "down"]
for k in range(4):
    sel = np.arange(k, nst - 4, 4)[2:]    # skip the first two blocks
    nxt = sel + 1
    B1 = T[:, sel, 1]            # all streamers of the WG done
    pub = T[:, sel, 2]           # y published (stores drained)
    seen = T[:, sel, 3]          # all WGs arrived
    xst = T[:, sel, 4]           # next x staged
    w0 = T[:, sel, 5]; w14 = T[:, sel, 6]
    xprev = T[:, sel - 1, 4]     # x of this stage staged (end of previous hand-off)
    # chip-level: stage span = max over WGs of seen(s) - max over WGs of seen(s-1)
    seen_prev = T[:, sel - 1, 3]
    span = seen.max(0) - seen_prev.max(0)
    print(f"{names[k]:8s} stage span (all arrived -> all arrived): {np.median(span):6.2f} us")
    print(f"         stream phase per WG (x staged -> B1): med {np.median(B1 - xprev):6.2f}  p10 {np.percentile(B1 - xprev, 10):6.2f}  p90 {np.percentile(B1 - xprev, 90):6.2f}")
    print(f"         wave0 done - x staged: med {np.median(w0 - xprev):6.2f};  wave14: med {np.median(w14 - xprev):6.2f}")
    print(f"         reduce+publish (B1 -> drained): med {np.median(pub - B1):6.2f}  p90 {np.percentile(pub - B1, 90):6.2f}")
    print(f"         wait for the last arrival (drained -> seen): med {np.median(seen - pub):6.2f}  p90 {np.percentile(seen - pub, 90):6.2f}   [skew: last B1 - first B1 = {np.median(B1.max(0) - B1.min(0)):5.2f}]")
    print(f"         last publish -> seen (hand-off latency): med {np.median(seen.min(0) - pub.max(0)):6.2f}   max-seen - last publish {np.median(seen.max(0) - pub.max(0)):6.2f}")
    print(f"         x staging (seen -> staged): med {np.median(xst - seen):6.2f}  p90 {np.percentile(xst - seen, 90):6.2f}")
