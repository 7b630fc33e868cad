#!/bin/bash
# lab: skinny GEMM at M = 8 / 32 over pools of distinct 7B-shaped layers (graph replay), shipped library vs variants ($@), with a
# correctness check of every variant against dequantise + fp32 matmul
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
cat > /tmp/tsk.py <<'PY'
import sys, torch
sys.path.insert(0, sys.argv[1])
from hqq_amd import ops
gs, nbits = 64, 4
g = torch.Generator().manual_seed(0)
def layer(N, K):
    R = N * K // gs
    P = ops.pack(nbits, torch.randint(0, 16, (R, gs), generator=g, dtype=torch.uint8).cuda())
    s = (torch.rand(R, 1, generator=g) * 0.004 + 0.001).half().cuda()
    z = (torch.rand(R, 1, generator=g) * 15).round().half().cuda()
    return P, s, z
out = []
for (N, K, nl) in ((4096, 4096, 40), (11008, 4096, 14), (4096, 11008, 14)):
    Ls = [layer(N, K) for _ in range(nl)]
    for M in (8, 32):
        x = (torch.randn(M, K, generator=g) * 0.5).half().cuda()
        ys = [torch.empty(M, N, dtype=torch.float16, device="cuda") for _ in range(nl)]
        f = lambda: [ops.forward(x, L[0], L[1], L[2], None, N, K, gs, nbits, out=y) for L, y in zip(Ls, ys)]
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
        out.append(f"{N}x{K} M={M}: {best:.2f} us (err {err:.1e})")
print("  " + " | ".join(out))
PY
for rep in 1 2; do
for v in "" "$@"; do
  if [ -z "$v" ]; then echo -n "shipped:"; python /tmp/tsk.py $R 2>&1 | grep -v amdgpu.ids; else echo -n "$v:"; HQQ_AMD_LIB=$R/tools/libhqq_hip_$v.so python /tmp/tsk.py $R 2>&1 | grep -v amdgpu.ids; fi
done; done 2>&1 | tee gpurun_out/r3/lab_skinny.txt
