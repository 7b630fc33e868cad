"""lab: in-kernel timeline of a chain of overlapped launches (build: tools/build_variant.sh chaints gemv_chain.hip -DGC_LAB_TS).
   HQQ_AMD_LIB=tools/libhqq_hip_chaints.so python tools/chain_ts.py [blocks]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from hqq_amd import _C, ops

blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 4
graph = (sys.argv[2] != "eager") if len(sys.argv) > 2 else True
dev = torch.device("cuda")
BLOCK = [("q", 4096, 4096), ("k", 4096, 4096), ("v", 4096, 4096), ("o", 4096, 4096), ("gate", 11008, 4096), ("up", 11008, 4096), ("down", 4096, 11008)]
GROUPS = [("q", "k", "v"), ("o",), ("gate", "up"), ("down",)]
g = torch.Generator(device=dev).manual_seed(0)
def layer(N, K):
    R = N * K // 64
    Wq = torch.randint(0, 256, (R // 2, 64), dtype=torch.uint8, device=dev, generator=g)
    s = (torch.rand(R, 1, device=dev, generator=g) * 0.004 + 0.001).half()
    z = (torch.rand(R, 1, device=dev, generator=g) * 15).half()
    return Wq, s, z
xs = {K: torch.randn(1, K, device=dev).half() for K in (4096, 11008)}
stages = []
for b in range(blocks):
    L = {n: layer(N, K) + (N, K) for n, N, K in BLOCK}
    for grp in GROUPS:
        K = L[grp[0]][4]
        stages.append((xs[K], [(L[n][0], L[n][1], L[n][2], None, L[n][3], torch.empty(1, L[n][3], device=dev, dtype=torch.float16)) for n in grp]))
n = len(stages)
chain = ops.LaunchChain(stages, 4, opts=0)
lib = _C.lib()
ts = torch.zeros(n * 4096 * 8, dtype=torch.int64, device=dev)
chain.run(); torch.cuda.synchronize()
lib.hqq_hip_lab_set_chain_ts.argtypes = [ctypes.c_void_p]
lib.hqq_hip_lab_set_chain_ts(ts.data_ptr())
if graph:
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        chain.run()
    for _ in range(5):
        gr.replay()
    torch.cuda.synchronize()
    ts.zero_(); torch.cuda.synchronize()
    gr.replay()
else:
    chain.run()
torch.cuda.synchronize()
print("status", chain.status(), "graph", graph)
T = ts.cpu().numpy().reshape(n, 4096, 8).astype(np.float64)
t00 = None
names = ["start", "issued2", "prebuilt", "flag", "x staged", "pre-contract", "loop end", "signalled"]
print("us since the first wave of stage 0 starts; per stage: first wave start | medians over waves of each stamp | last wave end")
for s in range(n):
    t = T[s]
    live = t[:, 0] > 0
    if not live.any():
        print(s, "no stamps"); continue
    t = t[live]
    if t00 is None:
        t00 = t[:, 0].min()
    u = (t - t00) / 100.0   # 100 MHz -> us
    med = np.median(u, axis=0)
    print(f"stage {s:3d} waves {live.sum():4d} first {u[:,0].min():8.2f} | " + " ".join(f"{nm} {m:7.2f}" for nm, m in zip(names, med)) + f" | flag p10 {np.percentile(u[:,3],10):7.2f} last end {u[:,7].max():8.2f}")
