"""CPU tests of the checkpoint wire format, both directions.

* tests/golden/ours_sd_cfg1_4b.npz is hqq_amd.HQQLinear.state_dict() of the BASELINE configs[0] layer, written ON AN MI355X by
  tests/golden/make_ours_state_dict.py (HIP solver + packer).  Where the reference is mounted (/root/reference — the authoring
  container, not the GPU box) it is loaded into the REFERENCE's HQQLinear and run through the reference's own forward.
* independent of the reference being present: the file carries exactly the keys / dtypes / shapes of the reference's own
  state_dict (tests/golden/refsd_cfg1_4b.npz, written by the reference), and — the solver being bit-exact — the same bytes.
"""
import os
import sys
import types

import numpy as np
import pytest

from conftest import load_golden

torch = pytest.importorskip("torch")
REF = os.environ.get("HQQ_REFERENCE", "/root/reference")


def _sd(g):
    sd = {}
    for k in g:
        if k.startswith("sd__"):
            name = k[4:]
            dt = getattr(torch, bytes(g["dt__" + name]).decode().split(".")[1])
            t = torch.from_numpy(np.array(g[k]))
            sd[name] = t.view(torch.bfloat16) if dt == torch.bfloat16 else t.to(dt)
    return sd


def test_our_state_dict_is_the_reference_schema_and_the_reference_bytes():
    ours, ref = _sd(load_golden("ours_sd_cfg1_4b")), _sd(load_golden("refsd_cfg1_4b"))
    assert set(ours) == set(ref)
    for k in ref:
        assert ours[k].dtype == ref[k].dtype and ours[k].shape == ref[k].shape, k
        assert torch.equal(ours[k], ref[k]), f"{k}: the HIP-written checkpoint differs from the reference-written one"


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference is mounted in the authoring container only")
def test_the_reference_loads_our_state_dict_and_its_own_forward_agrees():
    stub = types.ModuleType("termcolor")   # hard import at hqq/core/quantize.py:13, absent from the image
    stub.colored = lambda t, *a, **k: t
    sys.modules.setdefault("termcolor", stub)
    sys.path.insert(0, REF)
    try:
        from hqq.core.quantize import HQQBackend, HQQLinear
    finally:
        sys.path.remove(REF)
    HQQLinear.set_backend(HQQBackend.PYTORCH)
    g = load_golden("ours_sd_cfg1_4b")
    layer = HQQLinear(None, None, compute_dtype=torch.float16, device="cpu")
    layer.load_state_dict(_sd(g))
    x = torch.from_numpy(g["x_f32"]).half()
    with torch.no_grad():
        y = layer.forward(x)
    # the reference's CPU forward on OUR checkpoint vs our fused GPU forward on it, and vs the reference's forward on ITS checkpoint
    torch.testing.assert_close(y.float(), torch.from_numpy(g["y_f16"].astype(np.float32)), rtol=1e-3, atol=1e-3)
    # ... and, in THIS process, the reference's forward on its own checkpoint: same bytes in, same bits out.  (The stored y_f16 of
    # refsd_cfg1_4b.npz is compared with a tolerance only: torch's fp16 CPU matmul sums in a host-dependent order — core count,
    # ISA — and the fixture was written on another host.)
    gr = load_golden("refsd_cfg1_4b")
    layer_r = HQQLinear(None, None, compute_dtype=torch.float16, device="cpu")
    layer_r.load_state_dict(_sd(gr))
    with torch.no_grad():
        y_r = layer_r.forward(x)
    assert torch.equal(y, y_r), "same checkpoint bytes must give the reference the same output bits"
    torch.testing.assert_close(y.float(), torch.from_numpy(gr["y_f16"].astype(np.float32)), rtol=1e-3, atol=1e-3)
