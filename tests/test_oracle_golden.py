"""CPU tests: the oracle (oracle/hqq_oracle.c) against golden vectors produced by the reference
itself (tests/golden/make_golden.py).  This is what pins the oracle; the -m gpu tests then compare
the HIP kernels with the oracle and with the same golden files."""
import hashlib

import numpy as np
import pytest

from conftest import CD_CODE, load_golden

NB = [8, 4, 3, 2, 1]


@pytest.mark.parametrize("nbits", NB)
def test_bitpack_matches_reference(oracle, nbits):
    g = load_golden(f"pack_{nbits}b")
    i = 0
    while f"U{i}" in g:
        U, P = g[f"U{i}"], g[f"P{i}"]
        mine = oracle.pack(nbits, U)
        assert mine.dtype == P.dtype and np.array_equal(mine, P)           # bitpack.py pack_*: bit-exact
        assert np.array_equal(oracle.pack_np(nbits, U), P)                  # independent numpy restatement
        up = oracle.unpack(nbits, P)
        assert np.array_equal(up, oracle.unpack_np(nbits, P))
        if nbits == 3:
            assert np.array_equal(up, g[f"UP{i}"])                           # padded rows included
        assert np.array_equal(up[: len(U)], U)                               # tests/test_bitpack.py:24-34 property
        i += 1
    assert i >= 3


def test_bitpack_rejects_ragged_rows(oracle):
    # BitPack.pack_4bit_u8 on an odd row count raises in torch (slab shapes differ); the oracle reports it
    assert oracle.packed_rows(4, 7) < 0 and oracle.packed_rows(2, 6) < 0 and oracle.packed_rows(3, 7) == 1
    with pytest.raises(ValueError):
        oracle.pack(4, np.zeros((7, 8), np.uint8))


def test_aten_row_sum_order(oracle):
    # order restated from ATen SumKernel.cpp; spot values computed with torch 2.10 (x.sum(1)) at generation time
    g = load_golden("quant_4b_192x256")
    # the solver's `zero` equality below is the real check; here only the degenerate properties
    x = np.arange(64, dtype=np.float32)
    assert oracle.row_sum(x) == 2016.0
    x = np.zeros(64, np.float32); x[0] = 1e8; x[1] = 1.0; x[32] = -1e8
    assert oracle.row_sum(x) == 1.0      # x[0]+x[32] cancel first (lane pairing 32 apart), then +1 survives


QUANT_FILES = ([f"quant_4b_16x4096_gs{g}" for g in (512, 1024, 4096)] + ["quant_2b_16x4096_gs2048"] + [f"quant_{b}b_192x256" for b in NB] + [f"quant_{b}b_64x2048_normal" for b in (4, 3, 2)] +
               [f"quant_{b}b_16x128_edge" for b in (4, 3, 2)] + [f"quant_4b_32x256_gs{g}" for g in (32, 128, 256)])


@pytest.mark.parametrize("name", QUANT_FILES)
def test_quantize_matches_reference(oracle, name):
    g = load_golden(name)
    nbits, gs = int(g["nbits"]), int(g["gs"])
    W = g["W"]
    N, K = W.shape
    o = oracle.quantize(W, nbits=nbits, group_size=gs)
    # solver output: the reference's CPU float32 path
    assert np.array_equal(o["Wq"].reshape(-1), g["Wq_unpacked"].reshape(-1))
    assert np.array_equal(o["zero"], g["zero_f32"].reshape(-1, 1))
    assert np.array_equal(o["scale"], g["scale_f32"].reshape(-1, 1))
    packed = oracle.pack(nbits, o["Wq"])
    assert np.array_equal(packed, g["Wq_packed"])
    for cdn, code in CD_CODE.items():
        if f"Wdeq_{cdn}" not in g:
            continue
        s_cd, z_cd = oracle.to_cd(o["scale"], code), oracle.to_cd(o["zero"], code)
        vt = np.float32 if code == 0 else np.uint16
        assert np.array_equal(np.asarray(s_cd).view(vt).reshape(-1), g[f"scale_{cdn}"].view(vt).reshape(-1))
        assert np.array_equal(np.asarray(z_cd).view(vt).reshape(-1), g[f"zero_{cdn}"].view(vt).reshape(-1))
        Wd = oracle.dequantize(nbits, packed, s_cd, z_cd, N, K, gs, code)
        assert np.array_equal(np.asarray(Wd).view(vt), g[f"Wdeq_{cdn}"].view(vt))      # bit-exact dequantised weights
        x_cd = oracle.to_cd(g["x_f32"], code)
        b_cd = oracle.to_cd(g["bias_f32"], code) if "bias_f32" in g else None
        y, _ = oracle.matmul(x_cd, Wd, b_cd, code)
        yo, yr = oracle.from_cd(y, code), oracle.from_cd(g[f"y_{cdn}"], code)
        # BLAS accumulation order differs: 1e-3 (one fp16 ulp at |y|~1) absolute + relative
        tol = {"f32": 2e-6, "f16": 1e-3, "bf16": 8e-3}[cdn]
        np.testing.assert_allclose(yo, yr, rtol=tol, atol=tol)


@pytest.mark.parametrize("nbits", [4, 3, 2])
def test_config1_1024(oracle, nbits):
    """BASELINE.json configs[0]: nn.Linear(1024,1024) seed 0, gs=64 axis=1, quantised by the reference on CPU."""
    torch = pytest.importorskip("torch")
    g = load_golden(f"cfg1_1024_{nbits}b")
    torch.manual_seed(0)
    W = torch.nn.Linear(1024, 1024, bias=False).weight.data.numpy()
    if hashlib.sha256(W.tobytes()).hexdigest().encode() != g["W_sha256"].tobytes():
        pytest.skip("torch RNG stream differs from the one the fixture was generated with")
    o = oracle.quantize(W, nbits=nbits, group_size=64)
    packed = oracle.pack(nbits, o["Wq"])
    assert np.array_equal(packed, g["Wq_packed"])
    assert np.array_equal(o["scale"].reshape(-1), g["scale_f32"].reshape(-1))
    assert np.array_equal(o["zero"].reshape(-1), g["zero_f32"].reshape(-1))
    for cdn in ("f16", "f32"):
        code = CD_CODE[cdn]
        s_cd, z_cd = oracle.to_cd(o["scale"], code), oracle.to_cd(o["zero"], code)
        Wd = oracle.dequantize(nbits, packed, s_cd, z_cd, 1024, 1024, 64, code)
        assert hashlib.sha256(np.ascontiguousarray(Wd).tobytes()).hexdigest().encode() == g[f"Wdeq_sha256_{cdn}"].tobytes()
        y, _ = oracle.matmul(oracle.to_cd(g["x_f32"], code), Wd, None, code)
        tol = 1e-3 if cdn == "f16" else 2e-6
        np.testing.assert_allclose(oracle.from_cd(y, code), oracle.from_cd(g[f"y_{cdn}"], code), rtol=tol, atol=tol)


def test_half_conversions(oracle):
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(20000).astype(np.float32) * s for s in (1e-8, 1e-5, 1e-3, 1.0, 1e3, 7e4)])
    x = np.concatenate([x, np.array([0.0, -0.0, 65504.0, 65519.9, 65520.0, 2.0 ** -24, 2.0 ** -25, 3 * 2.0 ** -26, np.inf, -np.inf], np.float32)])
    with np.errstate(over="ignore"):
        want = x.astype(np.float16)
    got = oracle.to_cd(x, 1)
    assert np.array_equal(got.view(np.uint16), want.view(np.uint16))
    torch = pytest.importorskip("torch")
    wb = torch.from_numpy(x).to(torch.bfloat16).view(torch.uint16).numpy()
    assert np.array_equal(oracle.to_cd(x, 2), wb)


@pytest.mark.parametrize("nbits", [8, 4, 3, 2, 1])
def test_quantize_tensorwise_matches_reference(oracle, nbits):
    """Quantizer.quantize(channel_wise=False): one scale / zero from the tensor's min and max, levels packed in the tensor's shape"""
    g = load_golden(f"quant_tensorwise_{nbits}b_160x256")
    o = oracle.quantize_tensorwise(g["W"], nbits=nbits, round_zero=bool(g["round_zero"]))
    assert np.array_equal(o["Wq"], g["Wq_unpacked"])
    assert o["scale"].shape == g["scale_f32"].shape == () and np.array_equal(o["scale"].view(np.uint32), g["scale_f32"].view(np.uint32))
    assert np.array_equal(o["zero"].view(np.uint32), g["zero_f32"].view(np.uint32))
    assert np.array_equal(oracle.pack(nbits, o["Wq"]), g["Wq_packed"])
    if nbits != 3:
        z16, s16 = oracle.to_cd(o["zero"].reshape(1), 1), oracle.to_cd(o["scale"].reshape(1), 1)
        Wd = ((o["Wq"].astype(np.float16) - z16).astype(np.float16) * s16).astype(np.float16)
        assert np.array_equal(Wd.view(np.uint16), g["Wdeq_f16"].view(np.uint16))


AXIS0_FILES = [f"quant_axis0_{b}b_128x256" for b in (4, 3, 2, 8)] + ["quant_axis0_4b_32x80", "quant_axis0_4b_96x72_gs8", "quant_axis0_4b_256x256_gs128"]


@pytest.mark.parametrize("name", AXIS0_FILES)
def test_quantize_axis0_matches_reference(oracle, name):
    """Quantizer.quantize(axis=0): groups down the rows of the [gs, numel/gs] view; the solver's mean is ATen's OUTER-dimension
    float sum (32-column cascade blocks, 8-column and scalar tails) — levels, zero, scale and packed bytes bit-exact"""
    g = load_golden(name)
    nbits, gs = int(g["nbits"]), int(g["gs"])
    o = oracle.quantize_axis0(g["W"], nbits=nbits, group_size=gs)
    assert np.array_equal(o["Wq"], g["Wq_unpacked"])
    assert np.array_equal(o["zero"].view(np.uint32), g["zero_f32"].reshape(1, -1).view(np.uint32))
    assert np.array_equal(o["scale"].view(np.uint32), g["scale_f32"].reshape(1, -1).view(np.uint32))
    assert np.array_equal(oracle.pack(nbits, o["Wq"]), g["Wq_packed"])
    # (W_r - zero) * scale on the [gs, C] view, reshaped to [N, K]  (quantize.py:183-199)
    N, K = g["W"].shape
    z16, s16 = oracle.to_cd(o["zero"], 1), oracle.to_cd(o["scale"], 1)
    with np.errstate(over="ignore"):
        Wd = ((o["Wq"].astype(np.float16) - z16).astype(np.float16) * s16).astype(np.float16).reshape(N, K)
    assert np.array_equal(Wd.view(np.uint16), g["Wdeq_f16"].view(np.uint16))


def test_aten_outer_sum_order(oracle):
    """the restated order of torch.sum(dim=0) on a contiguous [n, C] float tensor (the mean of the axis-0 solver)"""
    torch = pytest.importorskip("torch")
    import ctypes
    L = oracle.lib()
    L.hqq_oracle_col_sum_f32.restype = ctypes.c_float
    L.hqq_oracle_col_sum_f32.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64]
    # (fewer than 8 columns take ATen's scalar outer-sum path, another order: the oracle and the HIP solver refuse C < 8)
    for (n, C) in ((64, 256), (64, 40), (128, 40), (32, 9), (8, 24), (256, 64), (64, 33), (100, 19), (64, 4096)):
        x = (torch.randn(n, C, generator=torch.Generator().manual_seed(n * C)) * 3).float()
        ref = x.sum(dim=0).numpy()
        xn = np.ascontiguousarray(x.numpy())
        got = np.array([L.hqq_oracle_col_sum_f32(xn[:, j:].ctypes.data_as(ctypes.c_void_p), C, n, j, C) for j in range(C)], np.float32)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (n, C)


@pytest.mark.parametrize("name", ["step_4b_axis1_384x64", "step_2b_axis1_128x64", "step_4b_axis0_64x512"])
def test_one_solver_step_formula_matches_reference(name):
    """optimize_weights_proximal_legacy_step (optimize.py:201-206) as the reference computed it (tests/golden/step_*.npz) against the
    op sequence the GPU test `test_one_proximal_step_on_its_own` evaluates with torch on the CPU — which pins that test's expectation
    (and hqq_amd.core.optimize.shrink_lp_op, host code) to the reference itself: W_r, W_q and the new zero-point bit for bit"""
    import torch
    from hqq_amd.core.optimize import shrink_lp_op
    g = load_golden(name)
    W, scale, zero = torch.from_numpy(g["W"]), torch.from_numpy(g["scale_in"]), torch.from_numpy(g["zero_in"])
    axis, max_v, beta, lp = int(g["axis"]), int(g["max_v"]), float(g["beta"]), float(g["lp_norm"])
    W_q = torch.round(W * scale + zero).clamp_(0, max_v)
    W_r = (W_q - zero) / scale
    W_e = shrink_lp_op(W - W_r, beta, lp)
    zero_new = torch.mean(W_q - (W - W_e) * scale, axis=axis, keepdim=True)
    assert np.array_equal(W_q.numpy().astype(np.uint8), g["W_q"])
    assert np.array_equal(W_r.numpy().view(np.uint32), g["W_r"].view(np.uint32))
    assert np.array_equal(zero_new.numpy().view(np.uint32), g["zero_out"].view(np.uint32))


def test_w3s_stream_layout_matches_the_golden_fixture(oracle):
    """the 3-bit stream layout (hqq_amd/csrc/w3s.h) as the oracle restates it, against tests/golden/w3s_layout.npz (written from the
    REFERENCE's pack_3bit_32 container by tests/golden/make_w3s_golden.py): container -> layout and back, byte for byte; and the layout's
    definition checked element by element on the smallest case (level (slab s, k) of a chunk where csrc/w3s.h says it is)"""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "w3s_layout.npz"))
    for name in ("a", "b", "c"):
        N, K = (int(v) for v in g[f"{name}_shape"])
        ref, want = g[f"{name}_ref"], g[f"{name}_w3s"]
        assert np.array_equal(oracle.pack(3, g[f"{name}_levels"]).view(np.int32), ref)
        got = oracle.w3s_pack_np(ref, N, K)
        assert got.shape == (N // 2, K // 16, 3) and np.array_equal(got.view(np.uint32), want)
        assert np.array_equal(oracle.w3s_unpack_np(want, N, K).view(np.int32), ref)
    N, K = (int(v) for v in g["a_shape"])
    L = g["a_levels"].reshape(N, K)
    D = g["a_w3s"]
    for p_ in range(N // 2):
        for c in range(K // 16):
            for s_ in range(2):
                for i in range(16):
                    d, b = oracle.w3s_pos(s_, i)
                    w = D[p_, c]
                    lv = (int(w[d]) >> b) & 7 if d >= 0 else sum(((int(w[t]) >> b) & 1) << t for t in range(3))
                    assert lv == int(L[p_ + s_ * (N // 2), 16 * c + i])
