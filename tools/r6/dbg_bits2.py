import sys, torch
sys.path.insert(0, ".")
from hqq_amd import ops
out = sys.argv[1]
res = {}
g = torch.Generator().manual_seed(0)
RZ = 1
def mk(N, K, nbits):
    R = N * K // 64
    if nbits == 3:
        U = torch.randint(0, 8, (R, 64), generator=g, dtype=torch.uint8)
        W = ops.w3s_pack(ops.pack(3, U.cuda()), N, K)
    else:
        U = torch.randint(0, 2 ** nbits, (R, 64), generator=g, dtype=torch.uint8)
        W = ops.pack(nbits, U.cuda())
    s = (torch.rand(R, 1, generator=g) * 0.004 + 0.001).half().cuda()
    z = (torch.rand(R, 1, generator=g) * (2 ** nbits - 1))
    z = (z.round() if RZ else z).half().cuda()
    return W, s, z
import itertools
for nbits, sub, rz in itertools.product((4, 3, 2), (0, 1), (0, 1)):
    if sub and not rz: continue
    o = (ops.OPT_W3S if nbits == 3 else 0) | (ops.OPT_META_SCALABLE if sub else 0)
    globals()['RZ'] = rz
    for K in (256, 512, 4096):
        x = torch.randn(1, K, generator=g).half().cuda()
        nw = (1 + 0.1 * torch.randn(K, generator=g)).half().cuda()
        # NORM: two layers
        Ls = [mk(256, K, nbits) + (256,), mk(128, K, nbits) + (128,)]
        outs = [torch.zeros(1, 256, dtype=torch.float16, device="cuda"), torch.zeros(1, 128, dtype=torch.float16, device="cuda")]
        ops.gemv_block(x, nw, 1e-5, Ls, K, 64, nbits, outs, ops.BLOCK_NORM, opts=o)
        res[(nbits, sub, rz, K, "norm")] = [t.cpu() for t in outs]
        # RESID
        L1 = mk(256, K, nbits) + (256,)
        h = torch.randn(1, 256, generator=g).half().cuda()
        ops.gemv_block(x, None, 0.0, [L1], K, 64, nbits, [h], ops.BLOCK_RESID, opts=o)
        res[(nbits, sub, rz, K, "resid")] = h.cpu()
        # plain gemv of the same layer for reference
        res[(nbits, sub, rz, K, "gemv")] = ops.gemv(x, L1[0], L1[1], L1[2], None, 256, K, 64, nbits, opts=o).cpu()
        # NORM | SILU on a "paired" layer (any layer of 2N rows works for the arithmetic)
        L2 = mk(512, K, nbits) + (512,)
        a = torch.zeros(1, 256, dtype=torch.float16, device="cuda")
        ops.gemv_block(x, nw, 1e-5, [L2], K, 64, nbits, [a], ops.BLOCK_NORM | ops.BLOCK_SILU, opts=o)
        res[(nbits, sub, rz, K, "silu")] = a.cpu()
        # NORM | ROPE
        hd, nh, L_ = 64, 4, 16
        Lq, Lk, Lv = mk(256, K, nbits) + (256,), mk(256, K, nbits) + (256,), mk(256, K, nbits) + (256,)
        qo = torch.zeros(nh, hd, dtype=torch.float16, device="cuda"); kc = torch.zeros(nh, L_, hd, dtype=torch.float16, device="cuda"); vc = torch.zeros_like(kc)
        cos = torch.randn(hd, generator=g).half().cuda(); sin = torch.randn(hd, generator=g).half().cuda(); pos = torch.tensor([5], device="cuda")
        ops.gemv_block(x, nw, 1e-5, [Lq, Lk, Lv], K, 64, nbits, [qo, kc, vc], ops.BLOCK_NORM | ops.BLOCK_ROPE, opts=o, rope=(cos, sin, pos, hd, L_))
        res[(nbits, sub, rz, K, "rope")] = [qo.cpu(), kc.cpu(), vc.cpu()]
torch.save(res, out)
print(len(res))
