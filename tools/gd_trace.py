#!/usr/bin/env python3
"""Phase timeline of the dense GEMM's main loop from the lab build (tools/build_variant.sh gdtrace gemm_dense.hip -DGD_TRACE):
    HQQ_AMD_LIB=tools/libhqq_hip_gdtrace.so python tools/gd_trace.py [M N K]
Per phase of K tiles 8..23, waves 0 / 4 / 5 of workgroup 0 (s_memtime ticks = shader cycles): wait = vmcnt wait, issue = DMA + read issue,
b1 = first barrier, mfma = 16 MFMAs issued, tail = reads returned + second barrier."""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from hqq_amd import _C, ops  # noqa: E402

M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (8192, 4096, 4096)
x = torch.randn(M, K, device="cuda").half()
W = (torch.randn(N, K, device="cuda") * 0.02).half()
y = torch.empty(M, N, device="cuda", dtype=torch.float16)
for _ in range(5):
    ops.gemm_dense(x, W, out=y)
torch.cuda.synchronize()
buf = np.zeros(3 * 16 * 4 * 5, dtype=np.uint64)
lib = _C.lib()
lib.hqq_hip_gd_trace_read.argtypes = [ctypes.c_void_p]
assert lib.hqq_hip_gd_trace_read(buf.ctypes.data) == 0
T = buf.reshape(3, 16 * 4, 5).astype(np.int64)
for w, name in enumerate(("wave 0 (wm 0)", "wave 4 (wm 1)", "wave 5 (wm 1)")):
    S = T[w]
    nxt = np.roll(S[:, 0], -1)
    seg = np.stack([S[:, 1] - S[:, 0], S[:, 2] - S[:, 1], S[:, 3] - S[:, 2], S[:, 4] - S[:, 3], nxt - S[:, 4]], axis=1)[:-1]
    ph = np.arange(len(seg)) % 4
    print(name, " K tile period", (S[-4, 0] - S[0, 0]) / 15.0, "cycles")
    for p in range(4):
        m = seg[ph == p].mean(axis=0)
        print(f"   phase {p}: wait {m[0]:6.0f}  issue {m[1]:6.0f}  b1 {m[2]:6.0f}  mfma {m[3]:6.0f}  tail {m[4]:6.0f}   sum {m.sum():6.0f}")
print("wave 0 vs wave 4 phase-0 start offset:", float((T[1, :, 0] - T[0, :, 0]).reshape(-1, 4)[:, 0].mean()))
