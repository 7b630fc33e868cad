#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by running the REFERENCE itself (mobiusml/hqq,
imported read-only from /root/reference) on the CPU of the authoring container.

    python tests/golden/make_golden.py            # needs /root/reference; writes tests/golden/*.npz

The reference's own tests hold no known-answer vectors for this path (only
round-trip / view-invariance properties), so these files ARE the pin: every array below is an
output of hqq/core/{quantize,optimize,bitpack}.py, untouched.  /root/reference does not exist on
the GPU box, so nothing in tests/ imports it at run time; only this script does.

`termcolor` (hard import at hqq/core/quantize.py:13) is absent from the image; a 2-line stub is
put on sys.path for the import only.

Fixture families
  pack_<nbits>b.npz          BitPack.pack_*/unpack_* on random ints, ragged 3-bit row counts
  quant_<tag>.npz            Quantizer.quantize (CPU => float32 solver) -> packed W_q, scale, zero;
                             HQQLinear(...).dequantize() and .forward(x) for fp16 / bf16 / fp32
  cfg1_1024_<nbits>b.npz     BASELINE.json configs[0]: nn.Linear(1024,1024) seed 0, gs=64 axis=1;
                             W itself is regenerated from the seed by the tests (sha256 stored)
  cfg2_4096_<nbits>b.npz     BASELINE.json configs[1] at full size: 4096x4096, W ~ N(0, 0.02^2) fp16-valued, seed 0:
                             sha256 of the reference's packed W_q / zero / scale (+ heads), forward(x) for fp16
  quant_tensorwise_<tag>.npz Quantizer.quantize(channel_wise=False): one scale / zero for the whole tensor, packed in its own shape
  quant_axis0_<tag>.npz      Quantizer.quantize(axis=0) (groups down the rows of the [gs, numel/gs] view) + HQQLinear(axis=0)
                             dequantize / forward for fp16
  step_<tag>.npz             optimize_weights_proximal_legacy_step (optimize.py:201-206) on a grouped tensor with a given start
                             (scale, zero): W_r, W_q and the new zero-point of ONE solver step
  refsd_cfg1_4b.npz          the reference's own HQQLinear.state_dict() (encoded, quantize.py:617-680) of the configs[0]
                             layer at 4 bits, fp16 — the wire format hqq_amd.HQQLinear.load_state_dict must accept
"""
import hashlib
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("HQQ_REFERENCE", "/root/reference")


def _import_reference():
    if not os.path.isdir(REF):
        sys.exit(f"{REF} not found: golden vectors can only be regenerated where the reference is mounted")
    stub = types.ModuleType("termcolor")
    stub.colored = lambda t, *a, **k: t
    sys.modules.setdefault("termcolor", stub)
    sys.path.insert(0, REF)
    from hqq.core.quantize import Quantizer, HQQLinear, BaseQuantizeConfig, HQQBackend  # noqa
    from hqq.core.bitpack import BitPack  # noqa
    HQQLinear.set_backend(HQQBackend.PYTORCH)
    return Quantizer, HQQLinear, BaseQuantizeConfig, BitPack


def raw(t: torch.Tensor) -> np.ndarray:
    t = t.detach().contiguous()
    return t.view(torch.uint16).numpy() if t.dtype == torch.bfloat16 else t.numpy()


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  {name}.npz  {os.path.getsize(path) / 1024:.0f} KiB")


CD = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}


def main():
    Quantizer, HQQLinear, BaseQuantizeConfig, BitPack = _import_reference()
    torch.set_num_threads(os.cpu_count() or 1)
    manifest = {"torch": torch.__version__, "reference": "mobiusml/hqq v0.2.8.post1 @ /root/reference", "files": {}}

    # ---------------- BitPack (bitpack.py) ----------------
    PACK = {8: (BitPack.pack_8bit_u8, BitPack.unpack_8bit_u8), 4: (BitPack.pack_4bit_u8, BitPack.unpack_4bit_u8),
            3: (BitPack.pack_3bit_32, BitPack.unpack_3bit_32), 2: (BitPack.pack_2bit_u8, BitPack.unpack_2bit_u8),
            1: (BitPack.pack_1bit_u8, BitPack.unpack_1bit_u8)}
    torch.manual_seed(42)  # tests/test_bitpack.py:15
    for nbits, (pk, upk) in PACK.items():
        arrs = {}
        shapes = [(32, 32), (128, 256), (1024, 64)] if nbits != 3 else [(32, 32), (37, 64), (1001, 8), (1024, 64)]
        for i, shp in enumerate(shapes):
            W = torch.randint(0, 2 ** nbits, shp)
            P = pk(W)
            U = upk(P)
            arrs[f"U{i}"] = W.numpy().astype(np.uint8)
            arrs[f"P{i}"] = P.numpy()
            if nbits == 3:   # unpack returns the zero-padded 10*ceil(R/10) rows (bitpack.py:95-110)
                arrs[f"UP{i}"] = U.numpy().astype(np.uint8)
            else:
                assert torch.equal(U.to(W.dtype), W)
        save(f"pack_{nbits}b", **arrs)

    # ---------------- Quantizer + HQQLinear, small shapes with W stored ----------------
    def quant_case(tag, W, bias, nbits, gs, M, cds=("f16", "bf16", "f32")):
        N, K = W.shape
        arrs = {"W": W.numpy().astype(np.float32), "nbits": np.array(nbits), "gs": np.array(gs)}
        # raw Quantizer output on CPU: float32 scale (=1/scale) and zero, packed W_q  (quantize.py:75-180)
        Wq, meta = Quantizer.quantize(W.clone(), nbits=nbits, group_size=gs, axis=1, round_zero=(nbits == 4),
                                      optimize=True, device="cpu", compute_dtype=torch.float16)
        arrs["Wq_packed"] = Wq.numpy()
        arrs["scale_f32"] = meta["scale"].numpy()
        arrs["zero_f32"] = meta["zero"].numpy()
        Wq_raw, _ = Quantizer.quantize(W.clone(), nbits=nbits, group_size=gs, axis=1, round_zero=(nbits == 4),
                                       optimize=True, device="cpu", bitpack=False)
        arrs["Wq_unpacked"] = Wq_raw.numpy().astype(np.uint8)
        if bias is not None:
            arrs["bias_f32"] = bias.numpy().astype(np.float32)
        torch.manual_seed(1)
        x32 = torch.randn(M, K)
        arrs["x_f32"] = x32.numpy()
        for cdn in cds:
            cd = CD[cdn]
            lin = torch.nn.Linear(K, N, bias=bias is not None)
            lin.weight.data = W.clone()
            if bias is not None:
                lin.bias.data = bias.clone()
            layer = HQQLinear(lin, BaseQuantizeConfig(nbits=nbits, group_size=gs, axis=1), compute_dtype=cd, device="cpu")
            assert np.array_equal(layer.W_q.data.numpy(), arrs["Wq_packed"])
            arrs[f"scale_{cdn}"] = raw(layer.meta["scale"])
            arrs[f"zero_{cdn}"] = raw(layer.meta["zero"])
            arrs[f"Wdeq_{cdn}"] = raw(layer.dequantize())
            with torch.no_grad():
                y = layer.forward(x32.to(cd))
            arrs[f"y_{cdn}"] = raw(y)
        save(tag, **arrs)

    for nbits in (4, 3, 2, 8, 1):
        torch.manual_seed(0)
        lin = torch.nn.Linear(256, 192, bias=True)   # kaiming-uniform weights, like tests/test_quantize.py:22-24
        quant_case(f"quant_{nbits}b_192x256", lin.weight.data.clone(), lin.bias.data.clone(), nbits, 64, 3)

    # N(0, 0.02^2) fp16-valued weights (BASELINE.md §3 inputs), K > 1024 so rows span several groups
    torch.manual_seed(0)
    Wn = (torch.randn(64, 2048) * 0.02).half().float()
    for nbits in (4, 3, 2):
        quant_case(f"quant_{nbits}b_64x2048_normal", Wn, None, nbits, 64, 2, cds=("f16",))

    # edge cases: constant group (|max-min|<=1e-4 -> scale 1, quantize.py:128), tiny range (scale clamp 2e4, :129),
    # all-zero rows, one huge outlier, exact .5 ties
    torch.manual_seed(3)
    We = torch.randn(16, 128) * 0.05
    We[0, :64] = 0.125                 # constant group
    We[1, :64] = 0.0                   # zero group
    We[2, :64] = 1.0 + torch.arange(64) * 1e-6   # denom 6.3e-5 <= 1e-4 -> scale = 1
    We[3, :64] = torch.linspace(0, 3e-4, 64)     # denom 3e-4 -> 15/3e-4 = 5e4 -> clamped to 2e4
    We[4, 5] = 40.0                    # outlier
    We[5, :64] = torch.arange(64) * 0.5           # many exact ties before rounding
    for nbits in (4, 3, 2):
        quant_case(f"quant_{nbits}b_16x128_edge", We, None, nbits, 64, 1, cds=("f16", "f32"))

    # other group sizes the config accepts (multiples of 8, quantize.py:1088-1091)
    torch.manual_seed(5)
    Wg = torch.randn(32, 256) * 0.1
    for gs in (32, 128, 256):
        quant_case(f"quant_4b_32x256_gs{gs}", Wg, None, 4, gs, 1, cds=("f16",))

    # large groups (ATen's row sum starts cascading at 512 elements) and one group per row (group_size = in_features)
    torch.manual_seed(6)
    Wl = torch.randn(16, 4096) * 0.05
    for gs in (512, 1024, 4096):
        quant_case(f"quant_4b_16x4096_gs{gs}", Wl, None, 4, gs, 1, cds=("f16",))
    quant_case("quant_2b_16x4096_gs2048", Wl, None, 2, 2048, 1, cds=("f16",))

    # ---------------- BASELINE.json configs[0]: 1024x1024 on CPU ----------------
    for nbits in (4, 3, 2):
        torch.manual_seed(0)
        lin = torch.nn.Linear(1024, 1024, bias=False)
        W = lin.weight.data.clone()
        arrs = {"W_sha256": np.frombuffer(sha(W.numpy()).encode(), np.uint8), "W_head": W.numpy()[:2, :8].copy()}
        torch.manual_seed(1)
        x32 = torch.randn(1, 1024)
        arrs["x_f32"] = x32.numpy()
        for cdn in ("f16", "f32"):
            cd = CD[cdn]
            torch.manual_seed(0)
            lin = torch.nn.Linear(1024, 1024, bias=False)
            layer = HQQLinear(lin, BaseQuantizeConfig(nbits=nbits, group_size=64, axis=1), compute_dtype=cd, device="cpu")
            if cdn == "f16":
                arrs["Wq_packed"] = layer.W_q.data.numpy()
            else:
                assert np.array_equal(arrs["Wq_packed"], layer.W_q.data.numpy())
                arrs["scale_f32"] = raw(layer.meta["scale"])
                arrs["zero_f32"] = raw(layer.meta["zero"])
            arrs[f"scale_{cdn}"] = raw(layer.meta["scale"])
            arrs[f"zero_{cdn}"] = raw(layer.meta["zero"])
            with torch.no_grad():
                arrs[f"y_{cdn}"] = raw(layer.forward(x32.to(cd)))
            arrs[f"Wdeq_sha256_{cdn}"] = np.frombuffer(sha(raw(layer.dequantize())).encode(), np.uint8)
        save(f"cfg1_1024_{nbits}b", **arrs)

    # ---------------- BASELINE.json configs[1] at full size: hashes of what the reference produces ----------------
    torch.manual_seed(0)
    W2 = (torch.randn(4096, 4096) * 0.02).half().float()
    torch.manual_seed(1)
    x2 = torch.randn(1, 4096)
    for nbits in (4, 3, 2):
        Wq, meta = Quantizer.quantize(W2.clone(), nbits=nbits, group_size=64, axis=1, round_zero=(nbits == 4),
                                      optimize=True, device="cpu", compute_dtype=torch.float16)
        arrs = {"W_sha256": np.frombuffer(sha(W2.numpy()).encode(), np.uint8), "W_head": W2.numpy()[:2, :8].copy(),
                "Wq_sha256": np.frombuffer(sha(Wq.numpy()).encode(), np.uint8), "Wq_head": Wq.numpy()[:4, :16].copy(),
                "scale_sha256": np.frombuffer(sha(meta["scale"].float().numpy()).encode(), np.uint8),
                "zero_sha256": np.frombuffer(sha(meta["zero"].float().numpy()).encode(), np.uint8),
                "scale_head": meta["scale"].float().numpy().reshape(-1)[:16].copy(), "zero_head": meta["zero"].float().numpy().reshape(-1)[:16].copy(),
                "x_f32": x2.numpy()}
        lin = torch.nn.Linear(4096, 4096, bias=False)
        lin.weight.data = W2.clone()
        layer = HQQLinear(lin, BaseQuantizeConfig(nbits=nbits, group_size=64, axis=1), compute_dtype=torch.float16, device="cpu")
        assert np.array_equal(layer.W_q.data.numpy(), Wq.numpy())
        with torch.no_grad():
            arrs["y_f16"] = raw(layer.forward(x2.half()))
        arrs["Wdeq_sha256_f16"] = np.frombuffer(sha(raw(layer.dequantize())).encode(), np.uint8)
        save(f"cfg2_4096_{nbits}b", **arrs)

    # ---------------- axis = 0 (the reference's better-quality layout, Readme.md:28-31; quantize.py:104-116) ----------------
    def axis0_case(tag, W, nbits, gs, M):
        N, K = W.shape
        arrs = {"W": W.numpy().astype(np.float32), "nbits": np.array(nbits), "gs": np.array(gs)}
        Wq, meta = Quantizer.quantize(W.clone(), nbits=nbits, group_size=gs, axis=0, round_zero=(nbits == 4),
                                      optimize=True, device="cpu", compute_dtype=torch.float16)
        arrs["Wq_packed"] = Wq.numpy()
        arrs["scale_f32"] = meta["scale"].numpy()
        arrs["zero_f32"] = meta["zero"].numpy()
        Wq_raw, _ = Quantizer.quantize(W.clone(), nbits=nbits, group_size=gs, axis=0, round_zero=(nbits == 4),
                                       optimize=True, device="cpu", bitpack=False)
        arrs["Wq_unpacked"] = Wq_raw.numpy().astype(np.uint8)
        torch.manual_seed(1)
        x32 = torch.randn(M, K)
        arrs["x_f32"] = x32.numpy()
        lin = torch.nn.Linear(K, N, bias=False)
        lin.weight.data = W.clone()
        layer = HQQLinear(lin, BaseQuantizeConfig(nbits=nbits, group_size=gs, axis=0), compute_dtype=torch.float16, device="cpu")
        assert np.array_equal(layer.W_q.data.numpy(), arrs["Wq_packed"])
        arrs["scale_f16"] = raw(layer.meta["scale"])
        arrs["zero_f16"] = raw(layer.meta["zero"])
        arrs["Wdeq_f16"] = raw(layer.dequantize())
        with torch.no_grad():
            arrs["y_f16"] = raw(layer.forward(x32.half()))
        save(tag, **arrs)

    torch.manual_seed(11)
    Wa = torch.randn(128, 256) * 0.05
    for nbits in (4, 3, 2, 8):
        axis0_case(f"quant_axis0_{nbits}b_128x256", Wa, nbits, 64, 2)
    torch.manual_seed(12)
    axis0_case("quant_axis0_4b_32x80", torch.randn(32, 80) * 0.05, 4, 64, 1)        # 40 groups: the 8-column and scalar tails of ATen's outer sum
    axis0_case("quant_axis0_4b_96x72_gs8", torch.randn(96, 72) * 0.05, 4, 8, 1)      # 864 groups of 8
    axis0_case("quant_axis0_4b_256x256_gs128", torch.randn(256, 256) * 0.05, 4, 128, 1)

    # ---------------- channel_wise=False: one scale / zero for the tensor (quantize.py:114-116) ----------------
    def tensorwise_case(tag, W, nbits, round_zero):
        arrs = {"W": W.numpy().astype(np.float32), "nbits": np.array(nbits), "round_zero": np.array(int(round_zero))}
        Wq, meta = Quantizer.quantize(W.clone(), nbits=nbits, channel_wise=False, group_size=None, optimize=False, round_zero=round_zero,
                                      axis=1, device="cpu", compute_dtype=torch.float16)
        arrs["Wq_packed"] = Wq.numpy()
        arrs["scale_f32"] = meta["scale"].numpy()
        arrs["zero_f32"] = meta["zero"].numpy()
        Wq_raw, _ = Quantizer.quantize(W.clone(), nbits=nbits, channel_wise=False, group_size=None, optimize=False, round_zero=round_zero,
                                       axis=1, device="cpu", bitpack=False)
        arrs["Wq_unpacked"] = Wq_raw.numpy().astype(np.uint8)
        meta16 = dict(meta)
        meta16["scale"], meta16["zero"] = meta["scale"].half(), meta["zero"].half()
        meta16["compute_dtype"] = torch.float16
        if nbits != 3:   # (the reference's dequantize divides by group_size=None for 3-bit, quantize.py:190-195)
            arrs["Wdeq_f16"] = raw(Quantizer.dequantize(Wq, meta16))
        save(tag, **arrs)

    torch.manual_seed(13)
    Wt = torch.randn(160, 256) * 0.05
    for nbits in (8, 4, 3, 2, 1):
        tensorwise_case(f"quant_tensorwise_{nbits}b_160x256", Wt, nbits, nbits == 4)

    # ---------------- one solver step on its own: optimize_weights_proximal_legacy_step (optimize.py:201-206) ----------------
    from hqq.core.optimize import optimize_weights_proximal_legacy_step

    def step_case(tag, Wg, nbits, axis):
        max_v = round(2 ** nbits - 1)
        _min, _max = Wg.min(axis=axis, keepdim=True)[0], Wg.max(axis=axis, keepdim=True)[0]
        scale = (max_v / (_max - _min)).clamp(max=2e4)   # (the start Quantizer.quantize gives the solver, quantize.py:118-134)
        zero = -_min * scale
        if nbits == 4:
            zero = torch.round(zero)
        W_r, W_q, zero_out, scale_out = optimize_weights_proximal_legacy_step(Wg.clone(), scale.clone(), zero.clone(), [0, max_v], 1e1, 0.7, axis)
        assert torch.equal(scale_out, scale)
        save(tag, W=Wg.numpy(), scale_in=scale.numpy(), zero_in=zero.numpy(), axis=np.array(axis), max_v=np.array(max_v), beta=np.array(1e1),
             lp_norm=np.array(0.7), W_r=W_r.numpy(), W_q=W_q.numpy().astype(np.uint8), zero_out=zero_out.numpy())

    torch.manual_seed(17)
    step_case("step_4b_axis1_384x64", torch.randn(384, 64) * 0.03, 4, 1)
    step_case("step_2b_axis1_128x64", torch.randn(128, 64) * 0.5 + 0.1, 2, 1)
    step_case("step_4b_axis0_64x512", torch.randn(64, 512) * 0.03, 4, 0)

    # ---------------- the reference's state_dict (wire format) of the configs[0] layer ----------------
    torch.manual_seed(0)
    lin = torch.nn.Linear(1024, 1024, bias=True)
    layer = HQQLinear(lin, BaseQuantizeConfig(nbits=4, group_size=64, axis=1), compute_dtype=torch.float16, device="cpu")
    sd = layer.state_dict()
    arrs = {}
    for k, v in sd.items():
        assert isinstance(v, torch.Tensor), (k, type(v))
        arrs["sd__" + k] = raw(v)
        arrs["dt__" + k] = np.frombuffer(str(v.dtype).encode(), np.uint8)
    torch.manual_seed(1)
    x32 = torch.randn(3, 1024)
    arrs["x_f32"] = x32.numpy()
    with torch.no_grad():
        arrs["y_f16"] = raw(layer.forward(x32.half()))
    arrs["Wdeq_sha256_f16"] = np.frombuffer(sha(raw(layer.dequantize())).encode(), np.uint8)
    save("refsd_cfg1_4b", **arrs)

    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            manifest["files"][f] = hashlib.sha256(open(os.path.join(HERE, f), "rb").read()).hexdigest()
    with open(os.path.join(HERE, "MANIFEST.json"), "w") as fh:
        json.dump(manifest, fh, indent=1, sort_keys=True)
    print("wrote MANIFEST.json")


if __name__ == "__main__":
    main()
