#!/bin/bash
# lab: the shape rule of the skinny GEMM's two tiles — launches of <= 2048 packed rows on the narrow tile (default) vs forced wide
mkdir -p gpurun_out/r3
cat > /tmp/tsk3.py <<'PY'
import sys, torch
sys.path.insert(0, sys.argv[1])
from hqq_amd import ops
gs = 64
g = torch.Generator().manual_seed(0)
def layer(N, K, nbits):
    R = N * K // gs
    P = ops.pack(nbits, torch.randint(0, 2 ** nbits, (R, gs), generator=g, dtype=torch.uint8).cuda())
    s = (torch.rand(R, 1, generator=g) * 0.004 + 0.001).half().cuda()
    z = (torch.rand(R, 1, generator=g) * (2 ** nbits - 1)).round().half().cuda()
    return P, s, z
for nbits in (4, 2):
    for (N, K, nl) in ((4096, 4096, 40 if nbits == 4 else 72), (4096, 11008, 14 if nbits == 4 else 26), (2048, 4096, 72), (8192, 8192, 10)):
        Ls = [layer(N, K, nbits) for _ in range(nl)]
        sub = ops.OPT_META_SCALABLE if all(ops.meta_scalable(L[1], L[2], N, K, gs, nbits) for L in Ls) else 0
        row = []
        for M in (8, 32, 64):
            x = (torch.randn(M, K, generator=g) * 0.5).half().cuda()
            ys = [torch.empty(M, N, dtype=torch.float16, device="cuda") for _ in range(nl)]
            res = {}
            for name, o in (("narrow-rule", sub), ("forced-wide", sub | ops.OPT_SKINNY_WIDE)):
                f = lambda: [ops.forward(x, L[0], L[1], L[2], None, N, K, gs, nbits, out=y, opts=o) for L, y in zip(Ls, ys)]
                f(); torch.cuda.synchronize()
                Wd = ops.dequantize(Ls[1][0], Ls[1][1].reshape(-1), Ls[1][2].reshape(-1), N, K, gs, nbits)
                err = (ys[1].float() - x.float() @ Wd.float().t()).abs().max().item()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    f()
                for _ in range(3): gr.replay()
                torch.cuda.synchronize()
                best = 1e9
                for _ in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); gr.replay(); gr.replay(); e1.record(); torch.cuda.synchronize()
                    best = min(best, e0.elapsed_time(e1) * 1e3 / (2 * nl))
                res[name] = (best, err)
            row.append(f"M={M}: rule {res['narrow-rule'][0]:.2f} us (err {res['narrow-rule'][1]:.1e}) | wide {res['forced-wide'][0]:.2f} us (err {res['forced-wide'][1]:.1e})")
        print(f"int{nbits} {N}x{K}: " + "   ".join(row), flush=True)
        del Ls
PY
python /tmp/tsk3.py $GRAFT_REPO_ROOT 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3/lab_skinny3.txt
