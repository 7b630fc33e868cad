"""bs = 1: us per launch of each of the 7B block's four launches (32 distinct blocks, graph replay, launches of ONE kind back to back):  python tools/r6/shapes1.py [nbits]"""
import sys, torch
sys.path.insert(0, ".")
from hqq_amd import ops
import bench
nbits = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda")
BLOCK = bench.LLAMA2_7B_BLOCK
blocks = [{n: bench.make_layer(ops, n, N, K, nbits, dev, seed=16 * b + i, random_codes=True) for i, (n, N, K) in enumerate(BLOCK)} for b in range(32)]
xs = {K: torch.randn(1, K, device=dev).half() for K in (4096, 11008)}
row = []
for grp in bench.EXCHANGE_GROUPS:
    outs = [torch.empty(1, blocks[0][n].N, device=dev, dtype=torch.float16) for n in grp]
    def run():
        for blk in blocks:
            Ls = [blk[n] for n in grp]
            ops.gemv_grouped(xs[Ls[0].K], [(L.Wq, L.scale, L.zero, None, L.N) for L in Ls], Ls[0].K, 64, nbits, outs=outs,
                             opts=ops.OPT_META_SCALABLE if all(L.opts & ops.OPT_META_SCALABLE for L in Ls) else 0)
    run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): g.replay()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 / len(blocks) * 1e3)
    row.append(best)
print("  ".join(f"{'|'.join(g_)} {t:6.2f} us" for g_, t in zip(bench.EXCHANGE_GROUPS, row)) + f"   block {sum(row):6.2f} us", flush=True)
