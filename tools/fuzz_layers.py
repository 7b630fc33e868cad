#!/usr/bin/env python3
"""Randomised sweep of the drop-in surface: HQQLinear (HIP backend) vs its own PYTORCH-backend forward, HQQLinearHIP after
prepare_for_inference, grouped projections — random shapes, bit widths, group sizes, dtypes, batch shapes (development aid; MI355X).
    python tools/fuzz_layers.py [cases] [seed]"""
import random
import sys

import torch
from torch import nn

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from hqq_amd.backends.hip import HQQLinearHIP, group_projections, patch_hqq_to_hip  # noqa: E402
from hqq_amd.core.quantize import BaseQuantizeConfig, HQQBackend, HQQLinear  # noqa: E402


def close(y, ref, dt, what, mag):
    ulp = 2.0 ** (-10 if dt == torch.float16 else -7)
    err = (y.double() - ref.double()).abs()
    tol = 2 * ulp * ref.double().abs().clamp(min=2.0 ** -4) + 1e-6 * mag.double() + 2e-3
    ok = bool((err <= tol).all()) and bool(torch.isfinite(y).all())
    if not ok:
        print(f"FAIL {what}: max err {err.max().item():.3e} (ref {ref.flatten()[err.argmax()].item():.4f})")
    return ok


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    for it in range(cases):
        torch.manual_seed(it)
        nbits = rnd.choice([8, 4, 4, 4, 3, 2, 1])
        gs = rnd.choice([64, 64, 64, 32, 128, 16])
        K = gs * rnd.randint(1, 20) if rnd.random() < 0.5 else 256 * rnd.randint(1, 12)
        K = max(gs, (K // gs) * gs)
        N = 8 * rnd.randint(1, 120)
        if nbits == 3 and (N * (K // gs) + 9) // 10 < K // gs:
            continue
        dt = rnd.choice([torch.float16, torch.float16, torch.bfloat16])
        bias = rnd.random() < 0.5
        lin = nn.Linear(K, N, bias=bias)
        cfg = BaseQuantizeConfig(nbits=nbits, group_size=gs, axis=1, view_as_float=rnd.random() < 0.2)
        layer = HQQLinear(lin, cfg, compute_dtype=dt, device="cuda")
        shape = rnd.choice([(1,), (3,), (1, 1), (2, 5), (4, 8), (17,), (2, 32), (70,), (1, 4), (9,), (300,), (2, 1300)])
        x = torch.randn(*shape, K, device="cuda", dtype=dt)
        what = f"[{it}] int{nbits} {N}x{K} gs={gs} {str(dt)[6:]} bias={bias} vf={cfg['weight_quant_params']['view_as_float']} x{shape}"
        HQQLinear.set_backend(HQQBackend.PYTORCH)
        try:
            ref = layer(x)
        finally:
            HQQLinear.set_backend(HQQBackend.HIP)
        Wd = layer.dequantize()
        mag = x.reshape(-1, K).float().abs() @ Wd.float().abs().t()
        mag = mag.reshape(*shape, N)
        try:
            y = layer(x)
            bad += not close(y, ref, dt, what + " HQQLinear", mag)
            assert y.shape == ref.shape and y.dtype == ref.dtype
            if rnd.random() < 0.4:
                # HQQLinear.merge: the layer and one or two more on the same input held as ONE layer — the same weights bit for bit, outputs side by side (within the forward tolerance: a launch's K split may depend on its row count)
                more = [HQQLinear(nn.Linear(K, 8 * rnd.randint(1, 60), bias=bias), cfg, compute_dtype=dt, device="cuda") for _ in range(rnd.randint(1, 2))]
                if nbits != 3 or all((m_.out_features * (K // gs) + 9) // 10 >= K // gs for m_ in more):
                    merged = HQQLinear.merge([layer] + more)
                    if not torch.equal(merged.dequantize(), torch.cat([l_.dequantize() for l_ in [layer] + more], 0)):
                        bad += 1
                        print(f"FAIL {what} merged weights differ")
                    refm = torch.cat([ref] + [l_.forward_pytorch(x) for l_ in more], -1)
                    magm = torch.cat([mag] + [(x.reshape(-1, K).float().abs() @ l_.dequantize().float().abs().t()).reshape(*shape, -1) for l_ in more], -1)
                    bad += not close(merged(x), refm, dt, what + " merged", magm)
            fast = patch_hqq_to_hip(layer, None)
            y2 = fast(x)
            bad += not close(y2, ref, dt, what + f" {type(fast).__name__}", mag)
            if isinstance(fast, HQQLinearHIP) and rnd.random() < 0.5 and dt == torch.float16:
                # a second layer on the same input, grouped
                lin2 = nn.Linear(K, 8 * rnd.randint(1, 60), bias=rnd.random() < 0.5)
                l2 = patch_hqq_to_hip(HQQLinear(lin2, cfg, compute_dtype=dt, device="cuda"), None)
                if isinstance(l2, HQQLinearHIP):
                    parent = nn.Module()
                    parent.a, parent.b = fast, l2
                    r2 = l2(x)
                    if group_projections(parent, ["a", "b"]):
                        ya, yb = parent.a(x), parent.b(x)
                        bad += not close(ya, ref, dt, what + " grouped a", mag)
                        if not torch.equal(yb, r2) and x.numel() // K <= 4:
                            bad += 1
                            print(f"FAIL {what} grouped b differs from ungrouped")
        except Exception as e:  # noqa: BLE001
            bad += 1
            print(f"FAIL {what}: {type(e).__name__}: {str(e)[:160]}")
    print(f"{cases} cases, {bad} failures")


if __name__ == "__main__":
    main()
