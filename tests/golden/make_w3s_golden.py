#!/usr/bin/env python3
"""Golden fixture of the 3-bit stream layout (hqq_amd/csrc/w3s.h): levels -> the reference's 3bit_32 container (reference BitPack.pack_3bit_32,
imported from /root/reference when present, else the oracle's restatement, which tests/test_oracle_golden.py pins to it) -> the stream layout
as oracle/hqq_oracle.py restates it.  Pins the layout across rounds: the kernels, the oracle and this file must agree byte for byte.
    python tests/golden/make_w3s_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import hqq_oracle as orc  # noqa: E402

orc.build()
rng = np.random.default_rng(20260925)
out = {}
for name, (N, K) in {"a": (8, 64), "b": (64, 128), "c": (34, 320)}.items():
    U = rng.integers(0, 8, size=(N * K // 64, 64), dtype=np.uint8)
    ref = None
    try:   # the reference's own packer, when the reference is mounted (this container)
        sys.path.insert(0, "/root/reference")
        sys.modules.setdefault("termcolor", type(sys)("termcolor")).colored = lambda t, *a, **k: t
        import torch
        from hqq.core.bitpack import BitPack
        ref = BitPack.pack_3bit_32(torch.from_numpy(U.astype(np.int32))).numpy().astype(np.int32)
    except Exception:
        pass
    mine = orc.pack(3, U)
    if ref is not None:
        assert np.array_equal(ref.view(np.uint32), mine.view(np.uint32)), "oracle pack != reference pack"
    w3s = orc.w3s_pack_np(mine, N, K)
    assert np.array_equal(orc.w3s_unpack_np(w3s, N, K).view(np.uint32), mine.view(np.uint32))
    out[f"{name}_levels"] = U
    out[f"{name}_ref"] = mine.view(np.int32)
    out[f"{name}_w3s"] = w3s.view(np.uint32)
    out[f"{name}_shape"] = np.array([N, K])
    out[f"{name}_from_reference"] = np.array([ref is not None])
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "w3s_layout.npz"), **out)
print("wrote tests/golden/w3s_layout.npz", {k: v.shape for k, v in out.items() if k.endswith("_w3s")}, "reference packer used:", bool(out["a_from_reference"][0]))
