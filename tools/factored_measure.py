"""How far HQQ_OPT_FACTORED (and the exact mode) are from the reference's outputs and from the double-accumulated oracle:
   max abs, max rel, outputs beyond rtol = atol = 1e-3 (north_star's forward tolerance).   python tools/factored_measure.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from hqq_amd import ops
from oracle import hqq_oracle as orc

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def stats(y, ref):
    d = (y.double() - ref.double()).abs()
    lim = 1e-3 + 1e-3 * ref.double().abs()
    return int((d > lim).sum()), float(d.max()), float((d / ref.double().abs().clamp_min(1e-6)).max()), float((d / lim).max())


def row(tag, y_e, y_f, ref):
    e, f = stats(y_e.float(), ref), stats(y_f.float(), ref)
    print(f"{tag:46s} n={ref.numel():6d} | exact: beyond {e[0]:4d} max|d| {e[1]:.2e} worst d/tol {e[3]:.2f} | factored: beyond {f[0]:4d} max|d| {f[1]:.2e} max rel {f[2]:.2e} worst d/tol {f[3]:.2f}")


print("against the REFERENCE's y_f16 (tests/golden, HQQBackend.PYTORCH on the CPU):")
for nbits in (4, 2):
    for name, N in ((f"cfg1_1024_{nbits}b", 1024), (f"cfg2_4096_{nbits}b", 4096)):
        g = dict(np.load(os.path.join(GOLD, name + ".npz")))
        x = torch.from_numpy(g["x_f32"]).cuda().half()
        ref = torch.from_numpy(g["y_f16"].astype(np.float32)).cuda()
        if "Wq_packed" in g:
            Wq, s, z = (torch.from_numpy(g[k]).cuda() for k in ("Wq_packed", "scale_f16", "zero_f16"))
        else:
            torch.manual_seed(0)
            W = (torch.randn(N, N) * 0.02).half().float()
            Wq, s, z = ops.quantize(W.cuda(), nbits=nbits, group_size=64, round_zero=(nbits == 4))
            s, z = s.half(), z.half()
        row(name, ops.gemv(x, Wq, s, z, None, N, N, 64, nbits, opts=0), ops.gemv(x, Wq, s, z, None, N, N, 64, nbits, opts=ops.OPT_FACTORED), ref)
print("against the oracle (reference-exact weights, double accumulation, one fp16 rounding), solver-quantised N(0, 0.02^2) layers, x ~ N(0,1), 4 rows:")
for nbits in (4, 2):
    for (N, K) in ((4096, 4096), (11008, 4096), (4096, 11008)):
        W = (torch.randn(N, K, generator=torch.Generator().manual_seed(N + K)) * 0.02).half().cuda()
        Wq, s, z = ops.quantize(W, nbits=nbits, group_size=64, round_zero=(nbits == 4))
        s, z = s.half(), z.half()
        Wd = ops.dequantize(Wq, s.reshape(-1), z.reshape(-1), N, K, 64, nbits).cpu().numpy()
        x = torch.randn(4, K, generator=torch.Generator().manual_seed(1)).half()
        yo, y32 = orc.matmul(x.numpy(), Wd, None, orc.F16)
        ref = torch.from_numpy(yo.astype(np.float32)).cuda()
        xd = x.cuda()
        row(f"int{nbits} {N}x{K} vs oracle fp16 output", ops.gemv(xd, Wq, s, z, None, N, K, 64, nbits, opts=0), ops.gemv(xd, Wq, s, z, None, N, K, 64, nbits, opts=ops.OPT_FACTORED), ref)
        ref32 = torch.from_numpy(y32).cuda()
        row(f"int{nbits} {N}x{K} vs oracle UNROUNDED fp32 sum", ops.gemv(xd, Wq, s, z, None, N, K, 64, nbits, opts=0), ops.gemv(xd, Wq, s, z, None, N, K, 64, nbits, opts=ops.OPT_FACTORED), ref32)
