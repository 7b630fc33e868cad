"""Per-wave timeline of the decode launches (lab library tools/r6/libhqq_hip_ts.so):  HQQ_AMD_LIB=tools/r6/libhqq_hip_ts.so python tools/r6/ts_run.py"""
import ctypes, sys, numpy as np, torch
sys.path.insert(0, ".")
from hqq_amd import ops, _C
import bench
dev = torch.device("cuda")
L = _C.lib()
lab = ctypes.CDLL(_C.LIB_PATH)
BLOCK = bench.LLAMA2_7B_BLOCK
nb = 12
blocks = [{n: bench.make_layer(ops, n, N, K, 4, dev, seed=16 * b + i, random_codes=False) for i, (n, N, K) in enumerate(BLOCK)} for b in range(nb)]
xs = {K: torch.randn(1, K, device=dev).half() for K in (4096, 11008)}
outs = {grp: [torch.empty(1, blocks[0][n].N, device=dev, dtype=torch.float16) for n in grp] for grp in bench.EXCHANGE_GROUPS}
ts = torch.zeros(8192 * 11, dtype=torch.int64, device=dev)
lab.hqq_lab_set_ts.argtypes = [ctypes.c_void_p]
lab.hqq_lab_set_ts(ctypes.c_void_p(ts.data_ptr()))
def launch(blk, grp):
    Ls = [blk[n] for n in grp]
    ops.gemv_grouped(xs[Ls[0].K], [(L_.Wq, L_.scale, L_.zero, None, L_.N) for L_ in Ls], Ls[0].K, 64, 4, outs=outs[grp],
                     opts=ops.OPT_META_SCALABLE if all(L_.opts & ops.OPT_META_SCALABLE for L_ in Ls) else 0)
seq = [(b, g) for b in range(nb) for g in bench.EXCHANGE_GROUPS]
for target in range(4):
    # the sequence up to a launch of the target kind in the last block: its stamps are what is left in the buffer
    upto = (nb - 1) * 4 + target
    def run_seq():
        for (b, g) in seq[:upto + 1]:
            launch(blocks[b], g)
    run_seq(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        run_seq()
    for rep in range(3):   # replayed: the launches run back to back, as in bench.py
        ts.zero_()
        torch.cuda.synchronize()
        gr.replay()
        torch.cuda.synchronize()
    t = ts.cpu().numpy().reshape(-1, 11)
    nw = 2048 if target in (1, 3) else 4096   # (o / down: 256 workgroups of 8 waves; the entries behind are an earlier launch's)
    t = t[:nw]
    t = t[t[:, 0] > 0]
    base = t[:, 0].min()
    rel = (t[:, :8] - base) / 100.0   # us
    ncons = t[:, 8]
    def pct(v): return " ".join(f"{np.percentile(v, q):5.2f}" for q in (10, 50, 90, 100))
    g = bench.EXCHANGE_GROUPS[target]
    print(f"== {'|'.join(g)}: {len(t)} waves, units per wave {ncons.min()}..{ncons.max()} (mean {ncons.mean():.2f}); us since the first wave's start, p10 p50 p90 max")
    print("  wave start            ", pct(rel[:, 0]))
    print("  first unit requested  ", pct(rel[:, 1]), "   (start -> request:", pct(rel[:, 1] - rel[:, 0]), ")")
    print("  x staged (at barrier) ", pct(rel[:, 2]))
    print("  barrier passed        ", pct(rel[:, 3]))
    live = ncons > 0
    print("  first consume begins  ", pct(rel[live, 4]))
    print("  first consume done    ", pct(rel[live, 5]), "   (its duration incl. the wait:", pct(rel[live, 5] - rel[live, 4]), ")")
    print("  last consume done     ", pct(rel[live, 6]))
    print("  wave exit             ", pct(rel[:, 7]))
    if target in (0, 2):
        # who waits between its start and its first request?  (kernel arguments beyond the preloaded 16 dwords are scalar loads; VMEM issue queues per CU)
        d = rel[:, 1] - rel[:, 0]
        xcc = t[:, 9]; hw = t[:, 10]; cu = (hw >> 8) & 15; se = (hw >> 13) & 7; sh = (hw >> 12) & 1
        cuid = xcc * 1000 + se * 100 + sh * 16 + cu
        wave_in_wg = np.arange(len(t)) % 4
        print("  start -> request by wave of its workgroup:", " | ".join(f"w{w}: " + pct(d[wave_in_wg == w]) for w in range(4)))
        order = np.zeros(len(t), dtype=int)
        for c in np.unique(cuid):
            idx = np.where(cuid == c)[0]
            order[idx[np.argsort(rel[idx, 0], kind="stable")]] = np.arange(len(idx))
        print("  CUs seen:", len(np.unique(cuid)), " waves per CU:", np.bincount(np.unique(cuid, return_inverse=True)[1]).min(), "..", np.bincount(np.unique(cuid, return_inverse=True)[1]).max())
        for lo, hi in ((0, 1), (1, 4), (4, 8), (8, 12), (12, 16), (16, 64)):
            m = (order >= lo) & (order < hi)
            if m.any(): print(f"  waves {lo}..{hi - 1} to start on their CU: start", pct(rel[m, 0]), " start -> request", pct(d[m]), " request at", pct(rel[m, 1]))
        for x in range(8):
            m = xcc == x
            if m.any(): print(f"  XCC {x}: first wave starts {rel[m, 0].min():5.2f}  start -> request", pct(d[m]), "  exit", pct(rel[m, 7]))
        slow = d > 0.8
        print(f"  waves with start -> request > 0.8 us: {slow.sum()}; their start", pct(rel[slow, 0]) if slow.any() else "", "; per CU count max", np.bincount(np.unique(cuid[slow], return_inverse=True)[1]).max() if slow.any() else 0, "over", len(np.unique(cuid[slow])), "CUs")
    for nc in np.unique(ncons):
        m = ncons == nc
        print(f"  waves with {nc} units: {m.sum():5d}; start", pct(rel[m, 0]), " first request", pct(rel[m, 1]), " exit", pct(rel[m, 7]), " request -> exit", pct(rel[m, 7] - rel[m, 1]))
    if target in (1, 3):
        xcc = t[:, 9]
        for x in range(8):
            m = xcc == x
            if m.any(): print(f"  XCC {x}: first wave starts {rel[m, 0].min():5.2f}  exit", pct(rel[m, 7]))
    # per-unit time once streaming: (last consume done - first consume done) / (units - 1)
    m = ncons > 1
    per = (rel[m, 6] - rel[m, 5]) / (ncons[m] - 1)
    print("  us per unit after the first (per wave):", pct(per))
