"""GPU tests added in round 3 ("parity hardening" and the advisor's findings):
* one proximal step on the GPU against the step fixtures the REFERENCE computed (tests/golden/step_*.npz), loaded here directly;
* bf16 at the Llama-2-7B shapes against the oracle; the unsharded Llama-2-70B shapes and the pipelined GEMM at 8192 rows against the
  ORACLE (its own unpack / dequantise / double-accumulated matmul), not this repo's dequantise kernel;
* the measured distance of HQQ_OPT_FACTORED from the reference's outputs, asserted as measured;
* channel_wise=False with a group size left in the meta; stale option bits after load_state_dict; the per-device workspace across streams."""
import numpy as np
import pytest

from conftest import load_golden

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "gpu tests need a ROCm device"
    from hqq_amd import ops as o
    assert o.is_available(), "libhqq_hip.so must load on the GPU box (no fallback)"
    return o


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _qlayer(ops, N, K, nbits, seed, gs=64, std=0.02):
    W = (torch.randn(N, K, generator=torch.Generator().manual_seed(seed)) * std).half().cuda()
    Wq, s, z = ops.quantize(W, nbits=nbits, group_size=gs, round_zero=(nbits == 4))
    return Wq, s.half(), z.half()


# ------------------------------------------------------------------------------------------------
# one solver step, pinned to what the reference computed
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["step_4b_axis1_384x64", "step_2b_axis1_128x64", "step_4b_axis0_64x512"])
def test_one_proximal_step_equals_the_reference_fixture(ops, name):
    """optimize_weights_proximal_legacy_step (optimize.py:201-206) through hqq_hip_optimize with one iteration: W_r, W_q and the new
    zero-point are, bit for bit, what the REFERENCE returned for the same W / scale / zero (tests/golden/make_golden.py wrote the fixture
    by calling the reference's function on the CPU)."""
    from hqq_amd.core.optimize import optimize_weights_proximal_legacy_step
    g = load_golden(name)
    axis, max_v = int(g["axis"]), int(g["max_v"])
    W, s, z = torch.from_numpy(g["W"]), torch.from_numpy(g["scale_in"]), torch.from_numpy(g["zero_in"])
    got_r, got_q, got_z, got_s = optimize_weights_proximal_legacy_step(W.cuda(), s.cuda(), z.cuda(), [0, max_v], float(g["beta"]), float(g["lp_norm"]), axis)
    assert np.array_equal(got_q.cpu().numpy().astype(np.uint8), g["W_q"])
    assert np.array_equal(got_r.cpu().numpy().view(np.uint32), g["W_r"].view(np.uint32))
    assert np.array_equal(got_z.cpu().numpy().view(np.uint32), g["zero_out"].view(np.uint32))
    assert torch.equal(got_s.cpu(), s)


# ------------------------------------------------------------------------------------------------
# full-size parity against the oracle: bf16 at the 7B shapes, fp16 at the unsharded 70B shapes, the pipelined GEMM at 8192 rows
# ------------------------------------------------------------------------------------------------
def _synthetic(oracle, N, K, nbits, code, seed):
    R = N * K // 64
    rng = np.random.default_rng(seed)
    U = rng.integers(0, 2 ** nbits, size=(R, 64), dtype=np.uint8)
    s = oracle.to_cd(rng.random((R, 1), dtype=np.float32) * 0.004 + 0.001, code)
    z = oracle.to_cd(rng.random((R, 1), dtype=np.float32) * (2 ** nbits - 1) * 0.5 + 0.25 * (2 ** nbits - 1), code)
    return rng, oracle.pack(nbits, U), s, z


@pytest.mark.parametrize("nbits", [4, 2])
@pytest.mark.parametrize("NK", [(4096, 4096), (11008, 4096), (4096, 11008)])
def test_full_size_forward_bf16_against_the_oracle(ops, oracle, nbits, NK):
    """bf16 compute dtype at the Llama-2-7B shapes: dequantised weights bit-exact against the oracle's bf16 arithmetic, the fused forward at
    1 row (row-per-wave kernel) and 32 rows (skinny kernel) within one bf16 ulp of the double-accumulated oracle (2^-7 relative, 2e-3 absolute)"""
    N, K = NK
    rng, P, s, z = _synthetic(oracle, N, K, nbits, oracle.BF16, N + K + nbits)
    Pd = dev(P)
    sd, zd = dev(s).view(torch.bfloat16), dev(z).view(torch.bfloat16)
    Wd = oracle.dequantize(nbits, P, s, z, N, K, 64, oracle.BF16)
    got = ops.dequantize(Pd, sd.reshape(-1), zd.reshape(-1), N, K, 64, nbits)
    assert np.array_equal(got.view(torch.int16).cpu().numpy().view(np.uint16), Wd.view(np.uint16))
    for M in (1, 32):
        x = oracle.to_cd(rng.standard_normal((M, K), dtype=np.float32), oracle.BF16)
        yo, _ = oracle.matmul(x, Wd, None, oracle.BF16)
        y = ops.forward(dev(x).view(torch.bfloat16), Pd, sd, zd, None, N, K, 64, nbits)
        torch.testing.assert_close(y.float().cpu(), torch.from_numpy(oracle.from_cd(yo, oracle.BF16)), rtol=2 ** -7, atol=2e-3)


@pytest.mark.parametrize("NK", [(8192, 8192), (28672, 8192), (8192, 28672)])
def test_70b_shapes_forward_against_the_oracle(ops, oracle, NK):
    """the unsharded Llama-2-70B shapes (BASELINE.json configs[4]) at int4: packed bytes, dequantised weights (sha-free: every bit) and
    the fused forward at 1 and 32 rows against the oracle, 1e-3"""
    N, K = NK
    nbits = 4
    rng, P, s, z = _synthetic(oracle, N, K, nbits, oracle.F16, N + K)
    Pd, sd, zd = dev(P), dev(s), dev(z)
    Wd = oracle.dequantize(nbits, P, s, z, N, K, 64, oracle.F16)
    got = ops.dequantize(Pd, sd.reshape(-1), zd.reshape(-1), N, K, 64, nbits)
    assert np.array_equal(got.cpu().numpy().view(np.uint16), Wd.view(np.uint16))
    del got
    for M in (1, 32):
        x = oracle.to_cd(rng.standard_normal((M, K), dtype=np.float32), oracle.F16)
        yo, _ = oracle.matmul(x, Wd, None, oracle.F16)
        y = ops.forward(dev(x), Pd, sd, zd, None, N, K, 64, nbits)
        torch.testing.assert_close(y.float().cpu(), torch.from_numpy(yo.astype(np.float32)), rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("cd", ["f16", "bf16"])
def test_pipelined_gemm_at_8192_rows_against_the_oracle(ops, oracle, cd):
    """gemm_pipe.hip at M = 8192 on a 4096 x 4096 int4 layer (one of BASELINE configs[2]'s chunks of 8192 tokens): all 8192 rows are
    computed on the GPU, 192 of them (spread over every token tile) are compared with the oracle's dequantise + double-accumulated matmul"""
    N = K = 4096
    nbits = 4
    code = oracle.F16 if cd == "f16" else oracle.BF16
    tdt = torch.float16 if cd == "f16" else torch.bfloat16
    rng, P, s, z = _synthetic(oracle, N, K, nbits, code, 99)
    Pd = dev(P)
    sd, zd = dev(s).view(tdt), dev(z).view(tdt)
    Wd = oracle.dequantize(nbits, P, s, z, N, K, 64, code)
    M = 8192
    x = oracle.to_cd(rng.standard_normal((M, K), dtype=np.float32), code)
    y = ops.gemm(dev(x).view(tdt), Pd, sd, zd, None, N, K, 64, nbits)
    rows = np.unique(np.concatenate([np.arange(0, M, 64), np.arange(37, M, 128)]))[:192]
    yo, _ = oracle.matmul(np.ascontiguousarray(x[rows]), Wd, None, code)
    tol = dict(rtol=1e-3, atol=2e-3) if cd == "f16" else dict(rtol=2 ** -7, atol=4e-3)
    torch.testing.assert_close(y[torch.from_numpy(rows).cuda()].float().cpu(), torch.from_numpy(oracle.from_cd(yo, code)), **tol)


# ------------------------------------------------------------------------------------------------
# HQQ_OPT_FACTORED: how far it is from the reference, measured
# ------------------------------------------------------------------------------------------------
def _beyond(y, ref, tol=1e-3):
    d = (y.double() - ref.double()).abs()
    lim = tol + tol * ref.double().abs()
    return int((d > lim).sum()), float(d.max()), float((d / ref.double().abs().clamp_min(1e-6)).max())


@pytest.mark.parametrize("nbits", [4, 2])
def test_factored_arithmetic_against_the_reference_outputs(ops, nbits):
    """north_star's forward tolerance is 1e-3 (rtol = atol) against HQQBackend.PYTORCH.  On the reference's own y (fixtures cfg1 1024^2 and
    cfg2 4096^2, written by the reference) the EXACT mode meets it; FACTORED — no per-weight fp16 rounding — is measured here and must
    stay inside the bound recorded in DESIGN.md section 4 (tools/factored_measure.py prints the figures this test pins)."""
    for name, N in ((f"cfg1_1024_{nbits}b", 1024), (f"cfg2_4096_{nbits}b", 4096)):
        g = load_golden(name)
        K = N
        x = dev(g["x_f32"]).half()
        ref = torch.from_numpy(g["y_f16"].astype(np.float32)).cuda()
        if "Wq_packed" in g:
            Wq, s, z = dev(g["Wq_packed"]), dev(g["scale_f16"]), dev(g["zero_f16"])
        else:
            torch.manual_seed(0)
            W = (torch.randn(N, K) * 0.02).half().float()
            Wq, s, z = ops.quantize(W.cuda(), nbits=nbits, group_size=64, round_zero=(nbits == 4))
            s, z = s.half(), z.half()
        y_exact = ops.gemv(x, Wq, s, z, None, N, K, 64, nbits, opts=0)
        y_fact = ops.gemv(x, Wq, s, z, None, N, K, 64, nbits, opts=ops.OPT_FACTORED)
        n_e, abs_e, _ = _beyond(y_exact.float(), ref)
        n_f, abs_f, rel_f = _beyond(y_fact.float(), ref)
        assert n_e == 0, f"{name}: exact mode leaves {n_e} outputs beyond 1e-3 (max abs {abs_e})"
        # measured (round 3, MI355X, profiles/r03_factored_measure.txt): FACTORED is at most ONE fp16 ulp of the output away from the
        # reference's y (2^-8 for 2 <= |y| < 4) — its sums differ from the exact ones by the reference's own weight-rounding noise, which
        # moves a result across a rounding boundary now and then — and leaves 0-1 of these 1024 / 4096 outputs (up to 0.4 % on a
        # 4096 x 11008 layer) beyond rtol = atol = 1e-3, by up to 1.8x: NOT inside north_star's tolerance on every output, so it stays opt-in.
        assert abs_f <= 2 ** -8 + 1e-6 and n_f <= max(2, N // 200), f"{name}: factored mode: {n_f} outputs beyond 1e-3, max abs {abs_f}, max rel {rel_f}"


# ------------------------------------------------------------------------------------------------
# advisor findings (round 2)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("nbits", [8, 4, 3, 2])
def test_channel_wise_false_with_a_group_size_in_the_meta(ops, nbits):
    """Quantizer.quantize(channel_wise=False) leaves the caller's group_size (default 64) in the meta; dequantize must still take the
    tensor-wide branch (one scale / zero, levels packed in the tensor's own shape) — the reference broadcasts the 0-d pair whatever
    group_size says.  Same result as with group_size=None."""
    from hqq_amd.core.quantize import Quantizer
    W = (torch.randn(120, 256, generator=torch.Generator().manual_seed(nbits)) * 0.1).half().cuda()
    Wq_a, meta_a = Quantizer.quantize(W, nbits=nbits, channel_wise=False, group_size=64, axis=1, compute_dtype=torch.float16, device="cuda")
    Wq_b, meta_b = Quantizer.quantize(W, nbits=nbits, channel_wise=False, group_size=None, axis=1, compute_dtype=torch.float16, device="cuda")
    assert torch.equal(Wq_a, Wq_b) and meta_a["scale"].numel() == 1
    for m in (meta_a, meta_b):
        m["compute_dtype"] = torch.float16
    Wq_a, meta_a = Quantizer.cuda(Wq_a, meta_a, "cuda")
    Wq_b, meta_b = Quantizer.cuda(Wq_b, meta_b, "cuda")
    Da, Db = Quantizer.dequantize(Wq_a, meta_a), Quantizer.dequantize(Wq_b, meta_b)
    assert Da.shape == W.shape and torch.equal(Da, Db)
    lv = ops.unpack(nbits, Wq_a)[: W.shape[0]].half()
    want = ((lv - meta_a["zero"]) * meta_a["scale"]).reshape(W.shape)
    assert torch.equal(Da, want)


def test_dequantize_checks_the_meta_size(ops):
    Wq, s, z = _qlayer(ops, 64, 256, 4, seed=0)
    with pytest.raises(ValueError):
        ops.dequantize(Wq, s.reshape(-1)[:2], z.reshape(-1)[:2], 64, 256, 64, 4)


def test_loading_a_state_dict_refreshes_the_three_op_bit(ops):
    """HQQLinearHIP keeps scale / zero as persistent buffers: a checkpoint whose zero-points do not survive the power-of-two scaling must
    drop HQQ_OPT_META_SCALABLE when it is loaded (the stale bit made the rebuild inexact, silently)"""
    from hqq_amd.backends.hip import HQQLinearHIP, patch_hqq_to_hip
    from hqq_amd.core.quantize import BaseQuantizeConfig, HQQLinear
    lin = torch.nn.Linear(256, 128, bias=False)
    layer = patch_hqq_to_hip(HQQLinear(lin, BaseQuantizeConfig(nbits=4, group_size=64, axis=1), compute_dtype=torch.float16, device="cuda"))
    assert isinstance(layer, HQQLinearHIP) and layer.opts == ops.OPT_META_SCALABLE
    sd = {k: v.clone() for k, v in layer.state_dict().items()}
    sd["zero"][3] = torch.tensor(0x0001, dtype=torch.int16).view(torch.float16)   # the smallest subnormal: lost by any down-scaling
    layer.load_state_dict(sd)
    assert layer.opts == 0
    x = torch.randn(2, 256, device="cuda").half()
    torch.testing.assert_close(layer(x).float(), x.float() @ layer.dequantize().float().t(), rtol=1e-3, atol=1e-3)
    e = torch.zeros(1, 256, dtype=torch.float16, device="cuda"); e[0, 200] = 1.0
    assert torch.equal(layer(e)[0], layer.dequantize()[:, 200])


def test_set_gemv_mode_reaches_the_layers(ops):
    from hqq_amd.backends.hip import patch_hqq_to_hip
    from hqq_amd.core.quantize import BaseQuantizeConfig, HQQLinear
    lin = torch.nn.Linear(512, 256, bias=False)
    layer = patch_hqq_to_hip(HQQLinear(lin, BaseQuantizeConfig(nbits=4, group_size=64, axis=1), compute_dtype=torch.float16, device="cuda"))
    x = torch.randn(1, 512, device="cuda").half()
    y_exact = layer(x)
    want_f = ops.gemv(x, layer.W_q, layer.scale, layer.zero, None, 256, 512, 64, 4, opts=ops.OPT_FACTORED)
    try:
        ops.set_gemv_mode(ops.GEMV_FACTORED)
        assert torch.equal(layer(x), want_f)
    finally:
        ops.set_gemv_mode(ops.GEMV_EXACT)
    assert torch.equal(layer(x), y_exact)


def test_workspace_users_on_two_streams_are_serialised(ops):
    """the per-device decode workspace (arrival counters + parked fp32 tiles of the split-K launches) is one buffer: calls from two
    streams must not overlap on it.  Alternating streams, 30 rounds, against the single-stream result."""
    N, K = 4096, 4096
    Wq, s, z = _qlayer(ops, N, K, 4, seed=11)
    x = torch.randn(32, K, device="cuda").half()
    want = ops.gemv(x, Wq, s, z, None, N, K, 64, 4)
    torch.cuda.synchronize()
    st = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = []
    for i in range(30):
        with torch.cuda.stream(st[i % 2]):
            outs.append(ops.gemv(x, Wq, s, z, None, N, K, 64, 4))
    torch.cuda.synchronize()
    assert all(torch.equal(o, want) for o in outs)


# ------------------------------------------------------------------------------------------------
# the skinny GEMM's two tiles (csrc/skinny.hip compiled twice): small 4-bit launches run on 32-row panels, HQQ_OPT_SKINNY_WIDE forces 64
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cd", ["f16", "bf16"])
@pytest.mark.parametrize("nbits", [4, 2])
@pytest.mark.parametrize("N,K,bias", [(4096, 4096, False), (4096, 11008, True), (1000, 1024, True), (68, 512, False), (2048, 2048, False), (8192, 4096, False)])
def test_both_tiles_of_the_skinny_gemm_against_the_oracle(ops, oracle, N, K, bias, cd, nbits):
    """a launch of <= 2048 packed rows is served by the narrow tile (a shape rule, never M); the same call with OPT_SKINNY_WIDE by the wide
    one: both within 1e-3 (fp16) / one bf16 ulp of the double-accumulated oracle on the oracle's exact weights, ragged last panels,
    bias, 5..64 rows, and reproducible run to run; a row's bits do not depend on the batch it is computed in"""
    code = oracle.F16 if cd == "f16" else oracle.BF16
    tdt = torch.float16 if cd == "f16" else torch.bfloat16
    rng, P, s, z = _synthetic(oracle, N, K, nbits, code, N * 3 + K + nbits)
    Pd, sd, zd = dev(P), dev(s).view(tdt), dev(z).view(tdt)
    Wd = oracle.dequantize(nbits, P, s, z, N, K, 64, code)
    b = oracle.to_cd(rng.standard_normal(N, dtype=np.float32) * 0.1, code) if bias else None
    bd = dev(b).view(tdt) if bias else None
    rtol, atol = (1e-3, 2e-3) if cd == "f16" else (2 ** -7, 4e-3)
    xs64 = oracle.to_cd(rng.standard_normal((64, K), dtype=np.float32), code)
    for M in (5, 17, 33, 64):
        x = xs64[:M]
        yo, _ = oracle.matmul(x, Wd, b, code)
        want = torch.from_numpy(oracle.from_cd(yo, code))
        for name, o in (("rule", 0), ("wide", ops.OPT_SKINNY_WIDE)):
            y = ops.forward(dev(x).view(tdt), Pd, sd, zd, bd, N, K, 64, nbits, opts=o)
            y2 = ops.forward(dev(x).view(tdt), Pd, sd, zd, bd, N, K, 64, nbits, opts=o)
            assert torch.equal(y, y2), "reproducible"
            torch.testing.assert_close(y.float().cpu(), want, rtol=rtol, atol=atol)
    for name, o in (("rule", 0), ("wide", ops.OPT_SKINNY_WIDE)):
        y64 = ops.forward(dev(xs64).view(tdt), Pd, sd, zd, bd, N, K, 64, nbits, opts=o)
        y17 = ops.forward(dev(xs64[:17]).view(tdt), Pd, sd, zd, bd, N, K, 64, nbits, opts=o)
        assert torch.equal(y64[:17], y17), f"{name}: a row must not depend on the batch it is computed in"


def test_skinny_gemm_with_more_than_128_kib_of_lds(ops, oracle):
    """2-bit, 64 rows, one K split of sixteen chunks on the wide tile: 64 KiB of x stages + 66.5 KiB of group constants — above the
    128 KiB the kernel used to ask for, inside gfx950's 160 KiB"""
    N, K, nbits = 4 * 64 * 150, 4096, 2
    rng, P, s, z = _synthetic(oracle, N, K, nbits, oracle.F16, 99)
    Pd, sd, zd = dev(P), dev(s).view(torch.float16), dev(z).view(torch.float16)
    x = oracle.to_cd(rng.standard_normal((64, K), dtype=np.float32), oracle.F16)
    y = ops.forward(dev(x).view(torch.float16), Pd, sd, zd, None, N, K, 64, nbits, opts=ops.OPT_SKINNY_WIDE)
    rows = rng.integers(0, N, size=512)
    Wd = oracle.dequantize(nbits, P, s, z, N, K, 64, oracle.F16)[rows]
    yo, _ = oracle.matmul(x, Wd, None, oracle.F16)
    torch.testing.assert_close(y[:, torch.from_numpy(rows).cuda()].float().cpu(), torch.from_numpy(yo.astype(np.float32)), rtol=1e-3, atol=2e-3)
