#!/bin/bash
# lab: skinny GEMM tile shapes x forced K splits over the four launches of a 7B block (pools of distinct layers > 256 MiB, graph replay)
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
cat > /tmp/tsk2.py <<'PY'
import sys, torch
sys.path.insert(0, sys.argv[1])
from hqq_amd import ops
import os
gs, nbits = 64, int(os.environ.get('NB', '4'))
g = torch.Generator().manual_seed(0)
def layer(N, K):
    R = N * K // gs
    P = ops.pack(nbits, torch.randint(0, 2 ** nbits, (R, gs), generator=g, dtype=torch.uint8).cuda())
    s = (torch.rand(R, 1, generator=g) * 0.004 + 0.001).half().cuda()
    z = (torch.rand(R, 1, generator=g) * (2 ** nbits - 1)).round().half().cuda()
    return (P, s, z, None, N)
SH = {"o": ([4096], 4096, 36), "qkv": ([4096] * 3, 4096, 12), "gateup": ([11008] * 2, 4096, 7), "down": ([4096], 11008, 13)}
for name, (Ns, K, nl) in SH.items():
    groups = [[layer(N, K) for N in Ns] for _ in range(nl)]
    sub = all(ops.meta_scalable(L[1], L[2], L[4], K, gs, nbits) for G in groups for L in G)
    base = ops.OPT_META_SCALABLE if sub else 0
    for M in (8, 32):
        x = (torch.randn(M, K, generator=g) * 0.5).half().cuda()
        row = [f"sub={int(sub)}"]
        for ks in (0, 1, 2, 3, 4, 6, 8):
            try:
                f = lambda: [ops.gemv_grouped(x, G, K, gs, nbits, opts=base | (ks << 24)) for G in groups]
                f(); torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    f()
                for _ in range(2): gr.replay()
                torch.cuda.synchronize()
                best = 1e9
                for _ in range(4):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); gr.replay(); gr.replay(); e1.record(); torch.cuda.synchronize()
                    best = min(best, e0.elapsed_time(e1) * 1e3 / (2 * nl))
                row.append(f"{ks}:{best:.2f}")
            except Exception as e:
                row.append(f"{ks}:ERR")
        print(f"  {name} M={M}  " + "  ".join(row), flush=True)
    del groups
PY
for v in "" "$@"; do
  if [ -z "$v" ]; then echo "shipped:"; python /tmp/tsk2.py $R 2>&1 | grep -v amdgpu.ids; else echo "$v:"; HQQ_AMD_LIB=$R/tools/libhqq_hip_$v.so python /tmp/tsk2.py $R 2>&1 | grep -v amdgpu.ids; fi
done 2>&1 | tee gpurun_out/r3/lab_skinny2_nb${NB:-4}.txt
