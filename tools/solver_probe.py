#!/usr/bin/env python3
"""lab: where does the HIP solver leave the oracle's float32 sequence?  (run on the GPU box)
1. mismatch counts of ops.quantize vs every quant golden + cfg1 (levels, zero bits, scale bits)
2. one proximal iteration stage by stage (tools/solver_probe.hip) vs a numpy float32 restatement"""
import ctypes, glob, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from hqq_amd import ops
from oracle import hqq_oracle as orc


def counts(name):
    g = np.load(os.path.join(ROOT, "tests", "golden", name))
    if "W" in g:
        W = g["W"]
    else:
        torch.manual_seed(0)
        W = torch.nn.Linear(1024, 1024, bias=False).weight.data.numpy()
    nbits = int(g["nbits"]) if "nbits" in g else int(name.split("_")[2][0])
    gs = int(g["gs"]) if "gs" in g else 64
    Wq, s, z, info = ops.quantize(torch.from_numpy(W).cuda(), nbits=nbits, group_size=gs, round_zero=(nbits == 4), return_info=True)
    got_u = ops.unpack(ops.PACK_BITS[nbits], Wq).cpu().numpy()[: W.size // gs]
    want_u = ops.unpack(ops.PACK_BITS[nbits], torch.from_numpy(g["Wq_packed"]).cuda()).cpu().numpy()[: W.size // gs]
    nbad = int((got_u != want_u).sum())
    zb = z.cpu().numpy().reshape(-1).view(np.uint32).astype(np.int64); zw = g["zero_f32"].reshape(-1).view(np.uint32).astype(np.int64)
    sb = s.cpu().numpy().reshape(-1).view(np.uint32); sw = g["scale_f32"].reshape(-1).view(np.uint32)
    dz = np.abs(zb - zw)
    print(f"{name:34s} nbits={nbits} gs={gs} numel={W.size:8d} levels_off={nbad:4d} zero_neq={int((dz != 0).sum()):6d}/{dz.size} max_ulp={int(dz.max())} scale_neq={int((sb != sw).sum())} info={info.cpu().numpy().tolist()}", flush=True)
    return W, nbits, gs


def stagewise(W, nbits, gs):
    lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libsolver_probe.so"))
    R = W.size // gs
    Wd = torch.from_numpy(np.ascontiguousarray(W, dtype=np.float32).reshape(-1)).cuda()
    outs = {k: torch.empty(R if k in ("sc", "ze") else R * gs, dtype=torch.float32, device="cuda") for k in ("sc", "ze", "q", "wr", "e", "pw", "pwf", "t", "we", "t3")}
    maxv = float(2 ** nbits - 1)
    inv_beta = np.float32(1.0 / 10.0)
    pexp = float(np.float32(0.7 - 1.0))
    vp = ctypes.c_void_p
    lib.probe_run.argtypes = [vp, ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_float, ctypes.c_double] + [vp] * 10 + [vp]
    rc = lib.probe_run(Wd.data_ptr(), R, gs, maxv, int(nbits == 4), float(inv_beta), pexp, *[outs[k].data_ptr() for k in ("sc", "ze", "q", "wr", "e", "pw", "pwf", "t", "we", "t3")], None)
    torch.cuda.synchronize()
    assert rc == 0
    o = {k: v.cpu().numpy() for k, v in outs.items()}
    # numpy float32 restatement (every op rounds once to float32, as the oracle's C does)
    f = np.float32
    w = W.reshape(R, gs).astype(f)
    mn, mx = w.min(1), w.max(1)
    denom = (mx - mn).astype(f)
    sc = ((f(1.0) / denom).astype(f) * f(maxv)).astype(f)
    sc = np.where(np.abs(denom) <= f(1e-4), f(1.0), sc); sc = np.minimum(sc, f(2e4)).astype(f)
    ze = ((-mn) * sc).astype(f)
    if nbits == 4: ze = np.rint(ze).astype(f)
    q = np.clip(np.rint(((w * sc[:, None]).astype(f) + ze[:, None]).astype(f)), 0, maxv).astype(f)
    wr = ((q - ze[:, None]).astype(f) / sc[:, None]).astype(f)
    e = (w - wr).astype(f)
    a = np.abs(e)
    with np.errstate(divide="ignore"):
        pw = np.power(a.astype(np.float64), pexp).astype(f)
    t = (a - (inv_beta * pw).astype(f)).astype(f); t = np.where(t < 0, f(0), t).astype(f)
    we = (t * np.sign(e).astype(f)).astype(f)
    t3 = (q - ((w - we).astype(f) * sc[:, None]).astype(f)).astype(f)
    ref = {"sc": sc, "ze": ze, "q": q.reshape(-1), "wr": wr.reshape(-1), "e": e.reshape(-1), "pw": pw.reshape(-1), "pwf": pw.reshape(-1), "t": t.reshape(-1), "we": we.reshape(-1), "t3": t3.reshape(-1)}
    for k in ("sc", "ze", "q", "wr", "e", "pw", "pwf", "t", "we", "t3"):
        gb, rb = o[k].view(np.uint32), ref[k].view(np.uint32)
        neq = gb != rb
        print(f"   stage {k:4s}: {int(neq.sum()):7d} of {neq.size} differ" + (f"  e.g. dev={o[k][neq][0]!r} ref={ref[k][neq][0]!r}" if neq.any() else ""), flush=True)


if __name__ == "__main__":
    files = sorted(os.path.basename(p) for p in glob.glob(os.path.join(ROOT, "tests", "golden", "quant_*.npz")))
    for n in files + ["cfg1_1024_4b.npz", "cfg1_1024_3b.npz", "cfg1_1024_2b.npz"]:
        try:
            W, nbits, gs = counts(n)
        except Exception as ex:   # keep going: this is a survey
            print(n, "FAILED", repr(ex), flush=True)
            continue
        if n in ("quant_4b_64x2048_normal.npz", "cfg1_1024_4b.npz", "quant_2b_192x256.npz"):
            stagewise(W, nbits, gs)
