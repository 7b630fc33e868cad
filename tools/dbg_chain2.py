import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from hqq_amd import ops, _C
nbits=4
dims=[(1024, [512, 1024])]
torch.manual_seed(1)
x0 = torch.randn(1, 1024, device="cuda").half()
Ls=[]
for j,N in enumerate(dims[0][1]):
    W = (torch.randn(N, 1024, generator=torch.Generator().manual_seed(j)) / 32).half().cuda()
    Wq, s, z = ops.quantize(W, nbits=nbits, group_size=64, round_zero=True)
    Ls.append((Wq, s.half(), z.half(), None, N, torch.full((1,N), float("nan"), device="cuda", dtype=torch.float16)))
plan = ops.DecodePlan([(x0, Ls)], nbits, opts=0, grid=8)
ts = torch.zeros(8192, dtype=torch.int64, device="cuda")
L = _C.lib(); L.hqq_hip_lab_set_engine_ts.argtypes=[ctypes.c_void_p]; L.hqq_hip_lab_set_engine_ts(ts.data_ptr())
plan.run(); torch.cuda.synchronize()
d = ts.cpu().numpy().view(np.uint32)
part = d[:4096].view(np.float32)
tab = d[4096:4096+64].view(np.int32)
print("tab fr:", tab[:16].tolist()); print("tab lr:", tab[16:32].tolist())
h = plan._host.raw
import struct
hdr = struct.unpack_from("<IIiiiiiiiiii", h, 0)
print("hdr", hdr)
maxf = hdr[10]
print("maxf", maxf)
P = part[:15*maxf*2*16].reshape(15, maxf, 2, 16)
for w in range(3):
    for f in range(3):
        print(w, f, "slab0", P[w,f,0].tolist()[:6], "nan" if np.isnan(P[w,f]).any() else "")
yb = d[4096+64:4096+64+8].view(np.float16)
print("ybuf head", yb.tolist())
print("y nan", torch.nonzero(torch.isnan(Ls[0][5][0])).flatten().tolist()[:8])
