#!/bin/bash
# lab: which part of a pipelined-GEMM step costs what.  Build the variants first, e.g.
#   for v in NODMA NOVALU NOMFMA NOBAR NODSR; do bash tools/build_variant.sh gp_$v gemm_pipe.hip "-DGP_LAB_$v"; done
#   bash tools/build_variant.sh gp_ONLYDSR gemm_pipe.hip "-DGP_LAB_NODMA -DGP_LAB_NOVALU -DGP_LAB_NOMFMA"   (and so on)
R=$GRAFT_REPO_ROOT
cat > /tmp/t.py <<'PY'
import sys, torch
sys.path.insert(0, sys.argv[1])
from hqq_amd import ops
N, K, gs, nbits = 4096, 4096, 64, 4
g = torch.Generator().manual_seed(0)
R = N * K // gs
P = ops.pack(nbits, torch.randint(0, 16, (R, gs), generator=g, dtype=torch.uint8).cuda())
s = (torch.rand(R, 1, generator=g) * 0.004 + 0.001).half().cuda()
z = (torch.rand(R, 1, generator=g) * 15).round().half().cuda()
for M, KS in ((128, 1), (128, 8), (1024, 1)):
    x = torch.randn(M, K, generator=g).half().cuda()
    y = torch.empty(M, N, dtype=torch.float16, device="cuda")
    f = lambda: ops.gemm(x, P, s, z, None, N, K, gs, nbits, out=y, opts=ops.OPT_META_SCALABLE | (KS << 24))
    for _ in range(3): f()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(20): f()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): gr.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"  M={M} KS={KS}: {e0.elapsed_time(e1) * 1e3 / 60:.1f} us", end="")
print()
PY
for v in "" gp_NODMA gp_NOVALU gp_NOMFMA gp_NOBAR gp_NODSR $EXTRA_VARIANTS; do
  if [ -z "$v" ]; then echo -n "shipped:"; python /tmp/t.py $R; else echo -n "$v:"; HQQ_AMD_LIB=$R/tools/libhqq_hip_$v.so python /tmp/t.py $R; fi
done
