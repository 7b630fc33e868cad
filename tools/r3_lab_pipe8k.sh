#!/bin/bash
# lab: the pipelined GEMM at 8192 rows over the 7B shapes, shipped library vs variants ($@ = names built by tools/build_variant.sh)
R=$GRAFT_REPO_ROOT
cat > /tmp/t8k.py <<'PY'
import sys, torch
sys.path.insert(0, sys.argv[1])
from hqq_amd import ops
gs, nbits, M = 64, 4, 8192
g = torch.Generator().manual_seed(0)
out = []
for (N, K) in ((4096, 4096), (12288, 4096), (22016, 4096), (4096, 11008)):
    R = N * K // gs
    P = ops.pack(nbits, torch.randint(0, 16, (R, gs), generator=g, dtype=torch.uint8).cuda())
    s = (torch.rand(R, 1, generator=g) * 0.004 + 0.001).half().cuda()
    z = (torch.rand(R, 1, generator=g) * 15).round().half().cuda()
    x = torch.randn(M, K, generator=g).half().cuda()
    y = torch.empty(M, N, dtype=torch.float16, device="cuda")
    f = lambda: ops.gemm(x, P, s, z, None, N, K, gs, nbits, out=y, opts=ops.OPT_META_SCALABLE)
    for _ in range(3): f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / 10)
    out.append(f"{N}x{K}: {best:.1f} us {2.0 * M * N * K / best / 1e6:.0f} TF")
print("  " + " | ".join(out))
PY
for rep in 1 2; do
for v in "" "$@"; do
  if [ -z "$v" ]; then echo -n "shipped:"; python /tmp/t8k.py $R; else echo -n "$v:"; HQQ_AMD_LIB=$R/tools/libhqq_hip_$v.so python /tmp/t8k.py $R; fi
done; done 2>&1 | tee gpurun_out/r3/lab_pipe8k.txt
