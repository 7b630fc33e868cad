import sys, torch
sys.path.insert(0, ".")
from hqq_amd import ops
g = torch.Generator().manual_seed(0)
K, nbits = 4096, 2
x = torch.randn(1, K, generator=g).half().cuda()
nw = (1 + 0.1 * torch.randn(K, generator=g)).half().cuda()
# the staged x, exactly: one-hot layer
N1 = 4096
U = torch.zeros(N1, K, dtype=torch.uint8); U[torch.arange(N1), torch.arange(N1)] = 1
W1 = ops.pack(nbits, U.reshape(N1 * K // 64, 64).cuda())
s1 = torch.ones(N1 * K // 64, 1).half().cuda(); z1 = torch.zeros(N1 * K // 64, 1).half().cuda()
xs = torch.zeros(1, N1, dtype=torch.float16, device="cuda")
ops.gemv_block(x, nw, 1e-5, [(W1, s1, z1, N1)], K, 64, nbits, [xs], ops.BLOCK_NORM)
for N in (256, 1024, 4096):
    R = N * K // 64
    Uq = torch.randint(0, 4, (R, 64), generator=g, dtype=torch.uint8)
    W = ops.pack(nbits, Uq.cuda())
    s = (torch.rand(R, 1, generator=g) * 0.004 + 0.001).half().cuda()
    z = (torch.rand(R, 1, generator=g) * 3).half().cuda()
    for o in (0,):
        a = torch.zeros(1, N, dtype=torch.float16, device="cuda")
        ops.gemv_block(x, nw, 1e-5, [(W, s, z, N)], K, 64, nbits, [a], ops.BLOCK_NORM, opts=o)
        h = torch.zeros(1, N, dtype=torch.float16, device="cuda")
        ops.gemv_block(xs, None, 0.0, [(W, s, z, N)], K, 64, nbits, [h], ops.BLOCK_RESID, opts=o)
        y = ops.gemv(xs, W, s, z, None, N, K, 64, nbits, opts=o)
        Wd = ops.dequantize(W, s.reshape(-1), z.reshape(-1), N, K, 64, nbits).double()
        ref = (xs.double() @ Wd.t())
        print("N", N, "norm-kernel vs resid-kernel ndiff", int((a != h).sum()), "| resid vs gemv", int((h != y).sum()),
              "| max err vs fp64: norm", float((a.double() - ref).abs().max()), "resid", float((h.double() - ref).abs().max()),
              "| worse-than-resid count", int(((a.double() - ref).abs() > (h.double() - ref).abs()).sum()), "better", int(((a.double() - ref).abs() < (h.double() - ref).abs()).sum()))
