"""GPU tests of the drop-in surface: HQQLinear / HQQBackend / prepare_for_inference / column shard on the HIP kernels,
against golden vectors produced by the reference's own HQQLinear (tests/golden/cfg1_1024_*.npz, BASELINE.json configs[0])."""
import hashlib

import numpy as np
import pytest

from conftest import load_golden

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from hqq_amd.core.quantize import BaseQuantizeConfig, HQQBackend, HQQLinear  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from hqq_amd import ops as o
    assert o.is_available()
    return o


def _cfg1_linear():
    torch.manual_seed(0)
    return torch.nn.Linear(1024, 1024, bias=False)


def _levels_differ(ops, nbits, Wq_a, Wq_b, R):
    a = ops.unpack(nbits, Wq_a)[:R].int()
    b = ops.unpack(nbits, Wq_b)[:R].int()
    return int((a != b).sum()), int((a - b).abs().max())


@pytest.mark.parametrize("nbits", [4, 3, 2])
def test_hqqlinear_config1_matches_reference(ops, nbits):
    g = load_golden(f"cfg1_1024_{nbits}b")
    lin = _cfg1_linear()
    if hashlib.sha256(lin.weight.data.numpy().tobytes()).hexdigest().encode() != g["W_sha256"].tobytes():
        pytest.skip("torch RNG stream differs from the one the fixture was generated with")
    layer = HQQLinear(lin, BaseQuantizeConfig(nbits=nbits, group_size=64, axis=1), compute_dtype=torch.float16, device="cuda")
    assert layer.ready and layer.in_gpu and HQQLinear.backend is HQQBackend.HIP
    assert isinstance(layer.W_q, torch.nn.Parameter) and not layer.W_q.requires_grad
    assert layer.W_q.dtype == (torch.int32 if nbits == 3 else torch.uint8)                # tests/test_quantize.py:36-39
    assert tuple(layer.W_q.shape) == tuple(g["Wq_packed"].shape) and tuple(layer.meta["shape"]) == (1024, 1024)
    assert layer.meta["scale"].dtype == torch.float16 and tuple(layer.meta["scale"].shape) == (1024 * 16, 1)
    assert (layer.in_features, layer.out_features) == (1024, 1024) and not hasattr(layer, "linear_layer")
    nbad, dmax = _levels_differ(ops, nbits, layer.W_q.data, torch.from_numpy(g["Wq_packed"]).cuda(), 1024 * 16)
    assert nbad == 0, f"{nbad} levels differ from the reference (max step {dmax})"         # bit-exact: tools/solver_probe.py, round 2
    x = torch.from_numpy(g["x_f32"]).cuda().half()
    want = torch.from_numpy(g["y_f16"].astype(np.float32))
    for backend in (HQQBackend.HIP, HQQBackend.PYTORCH, HQQBackend.PYTORCH_FORWARD, HQQBackend.ATEN_FORWARD):
        HQQLinear.set_backend(backend)
        try:
            y = layer(x)
        finally:
            HQQLinear.set_backend(HQQBackend.HIP)
        torch.testing.assert_close(y.float().cpu(), want, rtol=1e-3, atol=1e-3)
    # dequantize() == reference formula on the layer's own tensors, bit for bit (and pure: meta untouched)
    keys = set(layer.meta)
    Wd = layer.dequantize()
    assert set(layer.meta) == keys and tuple(Wd.shape) == (1024, 1024) and Wd.dtype == torch.float16
    U = layer.unpack()[: 1024 * 16]
    assert torch.equal(Wd, ((U - layer.meta["zero"]) * layer.meta["scale"]).reshape(1024, 1024))


def test_state_dict_round_trip_and_float_view(ops):
    lin = _cfg1_linear()
    lin.bias = torch.nn.Parameter(torch.randn(1024))
    cfg = BaseQuantizeConfig(nbits=4, group_size=64, axis=1)
    a = HQQLinear(lin, cfg, compute_dtype=torch.float16, device="cuda", del_orig=False)
    sd = a.state_dict()
    assert set(sd) <= a.state_dict_keys() and all(isinstance(v, torch.Tensor) for v in sd.values())   # safetensors-compatible
    assert sd["nbits"].dtype == torch.int32 and sd["packing"].dtype == torch.uint8 and sd["shape"].tolist() == [1024, 1024]
    b = HQQLinear(None, None, compute_dtype=torch.float16, device="cuda")
    b.load_state_dict({k: v.clone() if isinstance(v, torch.Tensor) else v for k, v in sd.items()})
    assert b.ready and torch.equal(b.W_q, a.W_q) and torch.equal(b.meta["scale"], a.meta["scale"]) and b.meta["packing"] == "4bit_u8"
    assert b.meta["compute_dtype"] == torch.float16 and b.quant_config["weight_quant_params"]["round_zero"] is True
    x = torch.randn(5, 1024, device="cuda", dtype=torch.float16)
    assert torch.equal(a(x), b(x))
    # nn.Module-style hierarchical load through _load_from_state_dict
    holder = torch.nn.Module()
    holder.proj = HQQLinear(None, None, compute_dtype=torch.float16, device="cuda")
    dest = {}
    a.state_dict(destination=dest, prefix="proj.")
    holder.load_state_dict(dest)
    assert torch.equal(holder.proj(x), a(x))
    # view_as_float: same bytes stored as compute dtype (tests/test_quantize.py:41-48, :163)
    cfg_f = BaseQuantizeConfig(nbits=4, group_size=64, axis=1, view_as_float=True)
    c = HQQLinear(lin, cfg_f, compute_dtype=torch.float16, device="cuda", del_orig=False)
    assert c.W_q.dtype == torch.float16 and torch.equal(c.W_q.data.view(torch.uint8), a.W_q.data)
    assert torch.equal(c(x), a(x)) and torch.equal(c.dequantize(), a.dequantize())


def test_reference_autograd_function_names(ops):
    """HQQMatmulNoCacheMul / HQQMatmulNoCacheDeq / HQQMatmulCachedDeq (quantize.py:289-385): same outputs and input gradients as
    x @ dequantize().t() + bias differentiated by torch, 2-D and 3-D inputs"""
    from hqq_amd.core.quantize import HQQMatmulCachedDeq, HQQMatmulNoCacheDeq, HQQMatmulNoCacheMul
    lin = torch.nn.Linear(512, 256, bias=True)
    layer = HQQLinear(lin, BaseQuantizeConfig(nbits=4, group_size=64), compute_dtype=torch.float16, device="cuda")
    W = layer.dequantize()
    for shape in ((5, 512), (2, 3, 512)):
        x0 = torch.randn(*shape, device="cuda", dtype=torch.float16)
        xr = x0.clone().requires_grad_(True)
        yr = torch.matmul(xr, W.t()) + layer.bias
        yr.float().square().sum().backward()
        for fn, arg in ((HQQMatmulNoCacheMul, layer.matmul), (HQQMatmulNoCacheDeq, layer.dequantize), (HQQMatmulCachedDeq, layer)):
            x = x0.clone().requires_grad_(True)
            y = fn.apply(x, arg, layer.bias)
            y.float().square().sum().backward()
            assert y.shape == yr.shape and torch.allclose(y.float(), yr.float(), rtol=2e-3, atol=2e-3), fn.__name__
            assert torch.allclose(x.grad.float(), xr.grad.float(), rtol=1e-2, atol=1e-2), fn.__name__


def test_backward_wrt_input_redequantises(ops):
    lin = _cfg1_linear()
    layer = HQQLinear(lin, BaseQuantizeConfig(nbits=4, group_size=64, axis=1), compute_dtype=torch.float16, device="cuda")
    x = torch.randn(4, 1024, device="cuda", dtype=torch.float16, requires_grad=True)
    y = layer(x)
    y.float().sum().backward()
    want = torch.ones(4, 1024, device="cuda", dtype=torch.float16) @ layer.dequantize()
    torch.testing.assert_close(x.grad, want, rtol=2e-3, atol=2e-2)


def test_prepare_for_inference_default_binds_the_inference_forward(ops):
    """backend="default" (patching.py:128-136): every HQQLinear gets the instance-bound inference forward — here the fused HIP one —
    same output bits as the class-wide forward, no autograd graph"""
    from hqq_amd.utils.patching import prepare_for_inference
    torch.manual_seed(5)
    model = torch.nn.Sequential(HQQLinear(torch.nn.Linear(256, 128, bias=True), BaseQuantizeConfig(nbits=4, group_size=64), compute_dtype=torch.float16, device="cuda"))
    x = torch.randn(3, 256, device="cuda", dtype=torch.float16)
    want = model(x)
    prepare_for_inference(model, backend="default")
    assert "forward" in vars(model[0]) and model[0].quant_config is not None
    got = model(x.clone().requires_grad_(True))
    assert torch.equal(got, want) and not got.requires_grad


def test_prepare_for_inference_swaps_layers(ops):
    from hqq_amd.backends.hip import HQQLinearHIP
    from hqq_amd.utils.patching import prepare_for_inference

    class Block(torch.nn.Module):
        def __init__(self):
            super().__init__()
            torch.manual_seed(3)
            cfg4 = BaseQuantizeConfig(nbits=4, group_size=64, axis=1)
            cfg3 = BaseQuantizeConfig(nbits=3, group_size=64, axis=1)
            self.q = HQQLinear(torch.nn.Linear(256, 512, bias=True), cfg4, compute_dtype=torch.float16, device="cuda")
            self.inner = torch.nn.Sequential(HQQLinear(torch.nn.Linear(512, 256, bias=False), cfg3, compute_dtype=torch.float16, device="cuda"))

        def forward(self, x):
            return self.inner(self.q(x))

    m = Block()
    x = torch.randn(2, 7, 256, device="cuda", dtype=torch.float16)
    before = m(x)
    wq = m.q.W_q.data.clone()
    prepare_for_inference(m, backend="hip")
    assert isinstance(m.q, HQQLinearHIP) and (m.q.in_features, m.q.out_features) == (256, 512) and m.q.bias is not None
    assert torch.equal(m.q.W_q.data, wq)                                   # packed bytes untouched (no repacking)
    assert isinstance(m.inner[0], HQQLinearHIP) and m.inner[0].nbits == 3   # 3-bit: fused decode kernel for <= 4 rows, dequant + GEMM beyond
    torch.testing.assert_close(m(x), before, rtol=1e-3, atol=1e-3)
    assert tuple(before.shape) == (2, 7, 256)
    x1 = x[:1, :1]
    torch.testing.assert_close(m(x1), before[:1, :1], rtol=1e-3, atol=2e-3)   # decode-sized call goes through the fused kernels
    assert torch.equal(m.q.dequantize().shape, torch.Size([512, 256])) if False else tuple(m.q.dequantize().shape) == (512, 256)


@pytest.mark.parametrize("nbits", [4, 2, 3])
@pytest.mark.parametrize("world", [2, 8])
def test_column_shard_emulated_on_one_gpu(ops, nbits, world):
    """every rank's packed slice run through the fused kernel, gathered and un-permuted == the full-layer forward.
    (gpurun boxes expose one GPU: the ranks run one after the other; the collective itself is covered by the gloo test.)"""
    from hqq_amd import shard
    N, K, gs, M = 640, 512, 64, 3
    g = torch.Generator().manual_seed(nbits)
    W = (torch.randn(N, K, generator=g) * 0.05).half().cuda()
    Wq, s, z = ops.quantize(W, nbits=nbits, group_size=gs, round_zero=(nbits == 4))
    s, z = s.half(), z.half()
    bias = torch.randn(N, generator=g).half().cuda()
    x = torch.randn(M, K, generator=g).half().cuda()

    def fwd(Wq_, s_, z_, b_, n_):
        return ops.forward(x, Wq_, s_, z_, b_, n_, K, gs, nbits, fused=True)   # the fused decode kernels, 3-bit included (gemv3*.hip)

    full = fwd(Wq, s, z, bias, N)
    parts = []
    for r in range(world):
        Wl, sl, zl, bl, n_loc = shard.shard_packed(Wq, s, z, bias, N, K, gs, nbits, r, world)
        assert n_loc == N // world
        parts.append(fwd(Wl.contiguous(), sl, zl, bl, n_loc))
    y = shard.unpermute(torch.stack(parts), N, nbits, world)
    if nbits == 3:   # a re-packed 3-bit shard lays its rows out over other slabs: same exact weights, another summation order
        torch.testing.assert_close(y.float(), full.float(), rtol=1e-3, atol=1e-3)
        Wd = ops.dequantize(Wq, s.reshape(-1), z.reshape(-1), N, K, gs, 3)
        torch.testing.assert_close(y.float(), x.float() @ Wd.float().t() + bias.float(), rtol=1e-3, atol=2e-3)
    else:
        assert torch.equal(y, full)   # same weights, same k order per output row -> bit-identical


@pytest.mark.parametrize("M", [1, 3, 32])
@pytest.mark.parametrize("world", [2, 8])
def test_sharded_3bit_layer_runs_in_the_stream_layout(ops, world, M):
    """ShardedHQQForward keeps a re-packed 3-bit shard in the 3-bit STREAM layout (csrc/w3s.h): the sharded layer then takes the same one-launch decode
    kernels as a patched unsharded one (missing #2).  One process stands in for every rank of a gloo group of size 1 per call is not
    possible, so the shard objects are built through the class's own constructor path with a stub process group of the right size."""
    from hqq_amd import shard
    N, K, gs = 1280, 1024, 64
    g = torch.Generator().manual_seed(world + M)
    W = (torch.randn(N, K, generator=g) * 0.05).half().cuda()
    Wq, s, z = ops.quantize(W, nbits=3, group_size=gs, round_zero=False)
    s, z = s.half(), z.half()
    x = torch.randn(M, K, generator=g).half().cuda()
    Wd = ops.dequantize(Wq, s.reshape(-1), z.reshape(-1), N, K, gs, 3)
    want = (x.double() @ Wd.double().t()).float()

    class _Dist:   # the two calls ShardedHQQForward.__init__ makes
        def __init__(self, r): self.r = r
        def get_world_size(self, group=None): return world
        def get_rank(self, group=None): return self.r
    parts = []
    for r in range(world):
        sh = shard.ShardedHQQForward.__new__(shard.ShardedHQQForward)
        import torch.distributed as real
        orig = (real.get_world_size, real.get_rank)
        real.get_world_size, real.get_rank = _Dist(r).get_world_size, _Dist(r).get_rank
        try:
            sh.__init__(Wq, s, z, None, N, K, gs, 3)
        finally:
            real.get_world_size, real.get_rank = orig
        assert sh.opts & ops.OPT_W3S and sh.Wq.shape == (N // world // 2, K // 16 * 3)
        parts.append(sh._local(x).reshape(M, -1))
    y = shard.unpermute(torch.stack(parts), N, 3, world)
    torch.testing.assert_close(y.float().cpu(), want.cpu(), rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("nbits,cd,bias", [(4, torch.float16, False), (4, torch.bfloat16, True), (3, torch.float16, False), (2, torch.float16, True), (8, torch.float16, False)])
def test_merged_layers_are_the_concatenation(ops, nbits, cd, bias):
    """HQQLinear.merge (q|k|v, gate|up as one layer): the same weights bit for bit, outputs = the layers' outputs side by side; state_dict round trip."""
    torch.manual_seed(7 + nbits)
    K, Ns = 512, (256, 128, 640)
    cfg = BaseQuantizeConfig(nbits=nbits, group_size=64, axis=1)
    layers = []
    for N in Ns:
        lin = torch.nn.Linear(K, N, bias=bias)
        layers.append(HQQLinear(lin, cfg, compute_dtype=cd, device="cuda"))
    merged = HQQLinear.merge(layers)
    assert (merged.in_features, merged.out_features) == (K, sum(Ns)) and tuple(merged.meta["shape"]) == (sum(Ns), K)
    assert merged.W_q.dtype == layers[0].W_q.dtype and merged.meta["scale"].shape[1:] == layers[0].meta["scale"].shape[1:]
    assert torch.equal(merged.dequantize(), torch.cat([l.dequantize() for l in layers], 0))                  # the same weights
    for l in layers:                                                                                         # the originals are untouched
        assert tuple(l.meta["shape"]) in [(N, K) for N in Ns]
    x1 = torch.randn(1, K, device="cuda").to(cd)
    for M in (1, 5, 128, 300):   # (outputs within the forward tolerance, not bit for bit: how a launch cuts K may depend on its row count)
        x = torch.randn(M, K, device="cuda").to(cd)
        want = torch.cat([l(x) for l in layers], -1).float()
        got = merged(x).float()
        torch.testing.assert_close(got, want, rtol=2e-2 if cd == torch.bfloat16 else 4e-3, atol=(2e-2 if cd == torch.bfloat16 else 4e-3) * float(want.abs().max()))
    for backend in (HQQBackend.PYTORCH, HQQBackend.PYTORCH_FORWARD):
        HQQLinear.set_backend(backend)
        try:
            y = merged(x1)
        finally:
            HQQLinear.set_backend(HQQBackend.HIP)
        torch.testing.assert_close(y.float(), merged(x1).float(), rtol=1e-2, atol=2e-2)
    fresh = HQQLinear(None, None, compute_dtype=cd, device="cuda", initialize=False)
    fresh.load_state_dict(merged.state_dict())
    assert torch.equal(fresh.W_q.data.view(torch.uint8), merged.W_q.data.view(torch.uint8)) and torch.equal(fresh.dequantize(), merged.dequantize())
    with pytest.raises(ValueError):
        HQQLinear.merge([layers[0], HQQLinear(torch.nn.Linear(256, 128, bias=bias), cfg, compute_dtype=cd, device="cuda")])   # another input width


def test_3bit_hqqlinear_hip_route_reads_the_stream_layout(ops):
    """set_backend(HQQBackend.HIP) on a 3-bit layer: the forward goes through the stream-layout copy (the same launches a patched layer makes: same bits), W_q / state_dict stay
    the reference's container, the copy follows the weights (load_state_dict, .cuda), and the class switch turns it off"""
    from hqq_amd.backends.hip import HQQLinearHIP
    torch.manual_seed(3)
    K, N = 1024, 768
    cfg = BaseQuantizeConfig(nbits=3, group_size=64, axis=1)
    layer = HQQLinear(torch.nn.Linear(K, N, bias=True), cfg, compute_dtype=torch.float16, device="cuda")
    sd = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in layer.state_dict().items()}
    xs = [torch.randn(M, K, device="cuda").half() for M in (1, 3, 32, 200)]
    ys = [layer(x) for x in xs]
    assert layer._w3s is not None and layer.W_q.dtype == torch.int32 and tuple(layer.W_q.shape) == tuple(sd["W_q"].shape) and torch.equal(layer.W_q.data, sd["W_q"])
    ref = layer.dequantize().float()
    for x, y in zip(xs, ys):
        torch.testing.assert_close(y.float(), x.float() @ ref.t() + layer.bias.float(), rtol=4e-3, atol=4e-3 * float(ref.abs().max()) * K ** 0.5)
    HQQLinear.stream_layout_3bit = False
    try:
        slow = [layer(x) for x in xs]
    finally:
        HQQLinear.stream_layout_3bit = True
    for y, y0 in zip(ys, slow):   # the same weights through the container's own kernels
        torch.testing.assert_close(y.float(), y0.float(), rtol=4e-3, atol=4e-3 * float(ref.abs().max()) * K ** 0.5)
    # another set of weights into the same module: the copy must follow
    other = HQQLinear(torch.nn.Linear(K, N, bias=True), cfg, compute_dtype=torch.float16, device="cuda")
    layer.load_state_dict(other.state_dict())
    assert torch.equal(layer(xs[0]), other(xs[0])) and not torch.equal(layer(xs[0]), ys[0])
    # the patched layer (which holds ONLY the stream layout) makes the same launches: bit for bit
    want = [other(x) for x in xs]
    fast = HQQLinearHIP(other)
    for x, w in zip(xs, want):
        assert torch.equal(fast(x), w)
    # under capture with no copy built yet: the container's path serves, nothing is allocated for the copy
    fresh = HQQLinear(torch.nn.Linear(K, N, bias=False), cfg, compute_dtype=torch.float16, device="cuda")
    HQQLinear.stream_layout_3bit = False
    try:
        fresh(xs[0])   # (warms the workspace outside the capture)
    finally:
        HQQLinear.stream_layout_3bit = True
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        yc = fresh(xs[0])
    g.replay()
    torch.cuda.synchronize()
    assert fresh._w3s is None
    torch.testing.assert_close(yc.float(), fresh(xs[0]).float(), rtol=4e-3, atol=4e-3 * float(ref.abs().max()) * K ** 0.5)
    assert fresh._w3s is not None


def test_3bit_layer_made_under_inference_mode_runs(ops):
    """tensors created under torch.inference_mode() carry no version counter (reading `_version` raises): a 3-bit layer quantised there
    must still build and key its stream-layout copy (round-5 advisor finding), and give the bits of a layer made outside it"""
    torch.manual_seed(5)
    K, N = 512, 256
    cfg = BaseQuantizeConfig(nbits=3, group_size=64, axis=1)
    lin = torch.nn.Linear(K, N, bias=False)
    x = torch.randn(2, K, device="cuda").half()
    with torch.inference_mode():
        layer = HQQLinear(lin, cfg, compute_dtype=torch.float16, device="cuda")
        assert layer.W_q.is_inference() or layer.meta["scale"].is_inference()
        y = layer(x)
        y2 = layer(x)   # the second call finds the copy by its key
        assert layer._w3s is not None and torch.equal(y, y2)
    ref = layer.dequantize().float()
    torch.testing.assert_close(y.float(), x.float() @ ref.t(), rtol=4e-3, atol=4e-3 * float(ref.abs().max()) * K ** 0.5)
