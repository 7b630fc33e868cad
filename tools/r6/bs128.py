"""bs = 65..1024 through the fused GEMM on the 7B block's launches (grouped), us per launch:  python tools/r6/bs128.py"""
import sys, torch
sys.path.insert(0, ".")
from hqq_amd import ops
import bench
dev = torch.device("cuda")
BLOCK = bench.LLAMA2_7B_BLOCK
blocks = [{n: bench.make_layer(ops, n, N, K, 4, dev, seed=16 * b + i, random_codes=False) for i, (n, N, K) in enumerate(BLOCK)} for b in range(8)]
for M in (128, 256, 512, 1024):
    xs = {K: torch.randn(M, K, device=dev).half() for K in (4096, 11008)}
    row = []
    for grp in bench.EXCHANGE_GROUPS:
        outs = [torch.empty(M, blocks[0][n].N, device=dev, dtype=torch.float16) for n in grp]
        def run():
            for blk in blocks:
                Ls = [blk[n] for n in grp]
                ops.gemm_grouped(xs[Ls[0].K], [(L.Wq, L.scale, L.zero, None, L.N) for L in Ls], Ls[0].K, 64, 4, outs=outs, opts=ops.OPT_META_SCALABLE if all(L.opts & ops.OPT_META_SCALABLE for L in Ls) else 0)
        run(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            run()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): g.replay()
        e1.record(); torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) / 10 / len(blocks) * 1e3)
    print(f"M={M:5d}  " + "  ".join(f"{'|'.join(g_)} {t:7.1f} us" for g_, t in zip(bench.EXCHANGE_GROUPS, row)) + f"   block {sum(row):7.1f} us", flush=True)
