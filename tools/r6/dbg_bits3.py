import sys, torch
sys.path.insert(0, ".")
from hqq_amd import ops
out = sys.argv[1]
g = torch.Generator().manual_seed(0)
K, nbits = 4096, 2
res = {}
x = torch.randn(1, K, generator=g).half().cuda()
nw = (1 + 0.1 * torch.randn(K, generator=g)).half().cuda()
# one-hot layer: row n picks k = (n * 37) % K
N = 4096
U = torch.zeros(N, K, dtype=torch.uint8)
ks = torch.arange(N) % K
U[torch.arange(N), ks] = 1
W = ops.pack(nbits, U.reshape(N * K // 64, 64).cuda())
s = torch.ones(N * K // 64, 1).half().cuda(); z = torch.zeros(N * K // 64, 1).half().cuda()
for sub in (0, 1):
    outs = [torch.zeros(1, N, dtype=torch.float16, device="cuda")]
    ops.gemv_block(x, nw, 1e-5, [(W, s, z, N)], K, 64, nbits, outs, ops.BLOCK_NORM, opts=(ops.OPT_META_SCALABLE if sub else 0))
    res[("xstaged", sub)] = outs[0].cpu()
    for rep in range(3):
        o2 = [torch.zeros(1, N, dtype=torch.float16, device="cuda")]
        ops.gemv_block(x, nw, 1e-5, [(W, s, z, N)], K, 64, nbits, o2, ops.BLOCK_NORM, opts=(ops.OPT_META_SCALABLE if sub else 0))
        assert torch.equal(o2[0].cpu(), res[("xstaged", sub)]), "nondeterministic"
# reference normalised x by torch (HF order, fp32 sum by torch)
hf = x.float(); var = hf.pow(2).mean(-1, keepdim=True); xn = (hf * torch.rsqrt(var + 1e-5)).half(); xr = (nw * xn)
res["ref"] = xr[0, ks.cuda()].cpu()
torch.save(res, out); print("ok")
