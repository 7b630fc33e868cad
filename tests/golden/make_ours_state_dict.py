#!/usr/bin/env python3
"""Run ON THE GPU BOX: quantise the BASELINE configs[0] layer (nn.Linear(1024,1024,bias=True), seed 0, 4-bit gs=64) with the HIP
solver and write hqq_amd.HQQLinear.state_dict() — the wire format of hqq/core/quantize.py:617-680 as THIS build emits it — to an
.npz.  tests/test_wire_format_cpu.py loads that file into the REFERENCE's HQQLinear (where /root/reference is mounted) and checks
the reference's own forward on it.

    gpurun -- 'python tests/golden/make_ours_state_dict.py gpurun_out/ours_sd_cfg1_4b.npz'   # then copy into tests/golden/
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hqq_amd.core.quantize import BaseQuantizeConfig, HQQLinear  # noqa: E402

out = sys.argv[1]
torch.manual_seed(0)
lin = torch.nn.Linear(1024, 1024, bias=True)
layer = HQQLinear(lin, BaseQuantizeConfig(nbits=4, group_size=64, axis=1), compute_dtype=torch.float16, device="cuda")
sd = layer.state_dict()
arrs = {}
for k, v in sd.items():
    assert isinstance(v, torch.Tensor), (k, type(v))
    v = v.detach().cpu().contiguous()
    arrs["sd__" + k] = v.view(torch.uint16).numpy() if v.dtype == torch.bfloat16 else v.numpy()
    arrs["dt__" + k] = np.frombuffer(str(v.dtype).encode(), np.uint8)
torch.manual_seed(1)
x = torch.randn(3, 1024)
arrs["x_f32"] = x.numpy()
with torch.no_grad():
    arrs["y_f16"] = layer(x.cuda().half()).cpu().numpy()
np.savez_compressed(out, **arrs)
print("wrote", out, os.path.getsize(out), "bytes")
