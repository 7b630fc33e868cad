import sys, torch
sys.path.insert(0, ".")
from hqq_amd import ops
out = sys.argv[1]
res = {}
g = torch.Generator().manual_seed(0)
for nbits in (4, 3, 2):
    for (N, K) in ((128, 64), (256, 128), (512, 256), (256, 1024), (4096, 4096), (1024, 11008)):
        R = N * K // 64
        if nbits == 3:
            U = torch.randint(0, 8, (R, 64), generator=g, dtype=torch.uint8)
            P = ops.pack(3, U.cuda())
            W = ops.w3s_pack(P, N, K)
            o = ops.OPT_W3S
        else:
            U = torch.randint(0, 2 ** nbits, (R, 64), generator=g, dtype=torch.uint8)
            W = ops.pack(nbits, U.cuda())
            o = 0
        s = (torch.rand(R, 1, generator=g) * 0.004 + 0.001).half().cuda()
        z = (torch.rand(R, 1, generator=g) * (2 ** nbits - 1)).round().half().cuda()
        for M in (1, 3):
            x = torch.randn(M, K, generator=g).half().cuda()
            for oo in (o, o | ops.OPT_META_SCALABLE):
                y = ops.gemv(x, W, s, z, None, N, K, 64, nbits, opts=oo)
                res[(nbits, N, K, M, oo)] = y.cpu()
torch.save(res, out)
print(len(res))
