import sys, numpy as np, torch
sys.path.insert(0, ".")
from hqq_amd import ops
from oracle import hqq_oracle as oracle
def _layer(N, K, nbits, seed, round_zero):
    g = torch.Generator().manual_seed(seed)
    R = N * K // 64
    U = torch.randint(0, 2 ** nbits, (R, 64), generator=g, dtype=torch.uint8)
    s = (torch.rand(R, 1, generator=g) * 0.004 + 0.001).half()
    z = torch.rand(R, 1, generator=g) * (2 ** nbits - 1)
    z = (z.round() if round_zero else z).half()
    return U, s, z
nbits, M, Ns, K = 8, 300, (1024, 128, 128, 72), 1024
layers, ref = [], []
for i, N in enumerate(Ns):
    U, s_, z_ = _layer(N, K, nbits, 17 * i + N + K, round_zero=(i % 2 == 0))
    P = oracle.pack(nbits, U.numpy())
    bias = None if i == 1 else torch.randn(N, generator=torch.Generator().manual_seed(i)).half()
    Wd = oracle.dequantize(nbits, P, s_.numpy(), z_.numpy(), N, K, 64, 1)
    layers.append((torch.from_numpy(P).cuda(), s_.cuda(), z_.cuda(), None if bias is None else bias.cuda(), N))
    ref.append((Wd, bias))
x = torch.randn(M, K, generator=torch.Generator().manual_seed(5)).half()
import ctypes
out8 = (ctypes.c_int * 8)()
ys = ops.gemm_grouped(x.cuda(), layers, K, 64, nbits)
for i, ((Pd, sd, zd, b, N), (Wd, bias), y) in enumerate(zip(layers, ref, ys)):
    yo, _ = oracle.matmul(x.numpy(), Wd, None if bias is None else bias.numpy(), 1)
    want = torch.from_numpy(yo.astype(np.float32))
    d = (y.float().cpu() - want).abs()
    alone = ops.gemm(x.cuda(), Pd, sd, zd, b, N, K, 64, nbits)
    d2 = (alone.float().cpu() - want).abs()
    idx = int(d.argmax()); r, c = idx // N, idx % N
    print(i, N, "grouped max", float(d.max()), "at", (r, c), "want", float(want[r, c]), "got", float(y[r, c]), "| alone max", float(d2.max()), "| bad cols", sorted(set((d > 0.05).nonzero()[:, 1].tolist()))[:20], "bad rows", sorted(set((d > 0.05).nonzero()[:, 0].tolist()))[:10])
