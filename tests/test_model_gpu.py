"""End-to-end on a GPU: a randomly initialised HF Llama decoder is quantised layer by layer with the HIP solver, runs through
the fused kernels via the reference's call chain (decoder layer -> q_proj(x) -> HQQLinear.forward), survives
prepare_for_inference, and decodes.  Small sizes: this is a plumbing test of SURVEY.md §8b "what calls it"."""
import pytest

torch = pytest.importorskip("torch")
transformers = pytest.importorskip("transformers")
pytestmark = pytest.mark.gpu


def _tiny_llama():
    from transformers import LlamaConfig, LlamaForCausalLM
    torch.manual_seed(0)
    cfg = LlamaConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=4,
                      vocab_size=512, max_position_embeddings=128)
    return LlamaForCausalLM(cfg).half().cuda().eval()


@pytest.mark.parametrize("nbits", [4, 3])
def test_quantize_hf_llama_and_decode(nbits):
    from hqq_amd.backends.hip import HQQLinearHIP
    from hqq_amd.core.quantize import BaseQuantizeConfig, HQQBackend, HQQLinear
    from hqq_amd.utils.model import LLAMA_LINEAR_TAGS, quantize_model
    from hqq_amd.utils.patching import prepare_for_inference
    assert torch.cuda.is_available()
    model = _tiny_llama()
    ids = torch.randint(0, 512, (2, 9), generator=torch.Generator().manual_seed(1)).cuda()
    with torch.no_grad():
        ref = model(ids).logits.float()
    cfg = BaseQuantizeConfig(nbits=nbits, group_size=64, axis=1)
    quantize_model(model, cfg, compute_dtype=torch.float16, device="cuda")
    qs = [m for m in model.modules() if isinstance(m, HQQLinear)]
    assert len(qs) == 2 * len(LLAMA_LINEAR_TAGS) and all(q.ready and q.W_q.is_cuda for q in qs)
    assert isinstance(model.lm_head, torch.nn.Linear)                      # untagged linears are left alone
    model.to(torch.float16)                                                # HF-style .to() must not disturb packed weights
    with torch.no_grad():
        out = model(ids).logits.float()
        one = model(ids[:, :1]).logits.float()                             # bs*seq = 2 rows -> the fused decode kernels
    # (i) plumbing: the fused kernels give what dequantise + dense matmul give on the same quantised weights
    HQQLinear.set_backend(HQQBackend.PYTORCH_FORWARD)
    try:
        with torch.no_grad():
            dense = model(ids).logits.float()
    finally:
        HQQLinear.set_backend(HQQBackend.HIP)
    torch.testing.assert_close(out, dense, rtol=2e-3, atol=2e-3)
    # (ii) sanity: quantisation noise only (random-init weights are the worst case for a 2-block toy model)
    rel = (out - ref).norm() / ref.norm()
    assert rel < (0.3 if nbits == 4 else 0.6), f"quantised logits drifted: rel {rel:.3f}"
    prepare_for_inference(model, backend="hip")
    assert sum(isinstance(m, HQQLinearHIP) for m in model.modules()) == len(qs)
    with torch.no_grad():
        out2 = model(ids).logits.float()
        one2 = model(ids[:, :1]).logits.float()
        gen = model.generate(ids[:1, :4], max_new_tokens=5, do_sample=False)
    torch.testing.assert_close(out2, out, rtol=2e-3, atol=2e-3)
    torch.testing.assert_close(one2, one, rtol=2e-3, atol=2e-3)
    assert tuple(gen.shape) == (1, 9)
    # q|k|v and gate|up as grouped launches (one kernel per distinct input): transparent to the HF modules, same numbers
    from hqq_amd.backends.hip import group_llama_projections
    assert group_llama_projections(model) == 2 * 2
    with torch.no_grad():
        one3 = model(ids[:, :1]).logits.float()          # 2 rows -> grouped decode kernel
        out3 = model(ids).logits.float()                 # 18 rows -> members fall back to their own forward
        gen3 = model.generate(ids[:1, :4], max_new_tokens=5, do_sample=False)
    assert torch.equal(one3, one2) and torch.equal(out3, out2) and torch.equal(gen3, gen)


def test_graphed_greedy_decoder_matches_generate():
    """one hipGraph per decode step over the quantised model (fused GEMVs, grouped q|k|v / gate|up) == eager greedy generate"""
    from hqq_amd.backends.hip import group_llama_projections
    from hqq_amd.core.quantize import BaseQuantizeConfig
    from hqq_amd.utils.generation import GraphedGreedyDecoder
    from hqq_amd.utils.model import quantize_model
    from hqq_amd.utils.patching import prepare_for_inference
    model = _tiny_llama()
    quantize_model(model, BaseQuantizeConfig(nbits=4, group_size=64, axis=1), compute_dtype=torch.float16, device="cuda")
    prepare_for_inference(model, backend="hip")
    group_llama_projections(model)
    ids = torch.randint(0, 512, (1, 6), generator=torch.Generator().manual_seed(3)).cuda()
    with torch.no_grad():
        want = model.generate(ids, max_new_tokens=12, do_sample=False)
    dec = GraphedGreedyDecoder(model, max_cache_len=64)
    eager = dec.generate(ids, 12, use_graph=False)
    graphed = dec.generate(ids, 12, use_graph=True)
    assert dec.graph is not None
    assert torch.equal(eager, want) and torch.equal(graphed, want)


@pytest.mark.parametrize("nbits", [4, 2])
def test_quantize_model_matches_the_reference_fixture(nbits):
    """SURVEY.md §8 f2 pinned to the reference: the 2-block toy Llama of tests/golden/make_model_golden.py, quantised THERE by the
    reference's AutoHQQHFModel.quantize_model (hqq/models/base.py:266-401, CPU, float32) and HERE by hqq_amd.utils.model.quantize_model
    on the GPU — the same linears are replaced (lm_head left alone) and every layer's packed W_q, zero and scale hash to the reference's."""
    import hashlib
    import numpy as np
    from conftest import load_golden
    from transformers import LlamaConfig, LlamaForCausalLM
    from hqq_amd.core.quantize import BaseQuantizeConfig, HQQLinear
    from hqq_amd.utils.model import quantize_model

    def sha(t):
        return hashlib.sha256(np.ascontiguousarray(t).tobytes()).hexdigest().encode()

    g = load_golden(f"model_llama2blk_{nbits}b")
    names = bytes(g["names"]).decode().split("\n")
    torch.manual_seed(0)
    cfg = LlamaConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=4,
                      vocab_size=512, max_position_embeddings=128)
    model = LlamaForCausalLM(cfg).float().eval()
    for n in names:
        if sha(model.get_submodule(n).weight.detach().numpy()) != g[f"src__{n}"].tobytes():
            pytest.skip("torch RNG stream differs from the one the fixture was generated with")
    model = model.cuda()
    quantize_model(model, BaseQuantizeConfig(nbits=nbits, group_size=64, axis=1), compute_dtype=torch.float32, device="cuda")
    got = [n for n, m in model.named_modules() if isinstance(m, HQQLinear)]
    assert got == names, "the set (and order) of replaced linears differs from the reference's"
    assert [n for n, m in model.named_modules() if isinstance(m, torch.nn.Linear)] == bytes(g["untouched"]).decode().split("\n")
    for n in names:
        m = model.get_submodule(n)
        assert list(m.meta["shape"]) == list(g[f"shape__{n}"])
        assert sha(m.W_q.data.cpu().numpy()) == g[f"Wq__{n}"].tobytes(), f"{n}: packed W_q differs from the reference"
        assert sha(m.meta["zero"].float().cpu().numpy()) == g[f"zero__{n}"].tobytes(), f"{n}: zero differs from the reference"
        assert sha(m.meta["scale"].float().cpu().numpy()) == g[f"scale__{n}"].tobytes(), f"{n}: scale differs from the reference"


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_block_glue_kernels_restate_the_hf_modules(dt):
    """csrc/block.hip against the HF modules it replaces in the fused decode step: rotary + cache write and SiLU * up bit for bit (elementwise,
    the same roundings); RMSNorm within one ulp on a handful of elements (the fp32 sum of squares is taken in another order); fp16 and bf16"""
    import torch.nn.functional as F
    from transformers import LlamaConfig
    from transformers.models.llama.modeling_llama import LlamaRMSNorm, LlamaRotaryEmbedding, apply_rotary_pos_emb
    from hqq_amd import ops
    g = torch.Generator(device="cuda").manual_seed(0)
    H, nh, nkv, hd, L = 4096, 32, 8, 128, 64
    # RMSNorm (+ residual add)
    h = torch.randn(1, H, device="cuda", generator=g).to(dt)
    d = (torch.randn(1, H, device="cuda", generator=g) * 0.3).to(dt)
    norm = LlamaRMSNorm(H, eps=1e-5).cuda().to(dt)
    norm.weight.data = (1 + 0.1 * torch.randn(H, device="cuda", generator=g)).to(dt)
    want_h = h + d
    want = norm(want_h)
    h2 = h.clone()
    got = ops.add_rmsnorm(h2, d, norm.weight, norm.variance_epsilon)
    assert torch.equal(h2, want_h)
    diff = (got.view(torch.int16).int() - want.view(torch.int16).int()).abs()
    assert int(diff.max()) <= 1 and int((diff > 0).sum()) <= 8, (int(diff.max()), int((diff > 0).sum()))
    assert torch.equal(ops.add_rmsnorm(h.clone(), None, norm.weight, norm.variance_epsilon)[0, :4].isfinite(), torch.ones(4, dtype=torch.bool, device="cuda"))
    for H2, rows in ((8192, 3), (5120, 2), (16392, 2), (64, 5)):   # two chunks per thread in registers; a ragged last chunk; the re-reading path; a tiny row
        hh = torch.randn(rows, H2, device="cuda", generator=g).to(dt)
        dd = (torch.randn(rows, H2, device="cuda", generator=g) * 0.3).to(dt)
        nn2 = LlamaRMSNorm(H2, eps=1e-6).cuda().to(dt)
        nn2.weight.data = (1 + 0.1 * torch.randn(H2, device="cuda", generator=g)).to(dt)
        w_h = hh + dd
        w_x = nn2(w_h)
        hh2 = hh.clone()
        g_x = ops.add_rmsnorm(hh2, dd, nn2.weight, nn2.variance_epsilon)
        assert torch.equal(hh2, w_h)
        df = (g_x.view(torch.int16).int() - w_x.view(torch.int16).int()).abs()
        assert int(df.max()) <= 1 and int((df > 0).sum()) <= 8 * rows, (H2, int(df.max()), int((df > 0).sum()))
    # rotary + cache write
    cfg = LlamaConfig(hidden_size=H, num_attention_heads=nh, num_key_value_heads=nkv, max_position_embeddings=2048)
    rot = LlamaRotaryEmbedding(cfg).cuda()
    pos = torch.tensor([37], device="cuda")
    q = torch.randn(1, nh * hd, device="cuda", generator=g).to(dt)
    k = torch.randn(1, nkv * hd, device="cuda", generator=g).to(dt)
    v = torch.randn(1, nkv * hd, device="cuda", generator=g).to(dt)
    cos, sin = rot(q.view(1, 1, -1), pos.view(1, 1))
    qe, ke = apply_rotary_pos_emb(q.view(1, 1, nh, hd).transpose(1, 2), k.view(1, 1, nkv, hd).transpose(1, 2), cos, sin)
    kc = torch.zeros(nkv, L, hd, dtype=dt, device="cuda")
    vc = torch.zeros(nkv, L, hd, dtype=dt, device="cuda")
    qr = torch.empty(1, nh, 1, hd, dtype=dt, device="cuda")
    ops.rope_cache(q, k, v, cos.reshape(-1).contiguous(), sin.reshape(-1).contiguous(), pos, kc, vc, qr)
    assert torch.equal(qr, qe) and torch.equal(kc[:, 37], ke[0, :, 0]) and torch.equal(vc[:, 37], v.view(nkv, hd))
    assert torch.count_nonzero(kc[:, :37]) == 0 and torch.count_nonzero(kc[:, 38:]) == 0
    # SiLU(gate) * up
    gt = (torch.randn(1, 11008, device="cuda", generator=g) * 2).to(dt)
    up = torch.randn(1, 11008, device="cuda", generator=g).to(dt)
    assert torch.equal(ops.silu_mul(gt, up), F.silu(gt) * up)


@pytest.mark.parametrize("glue", ["folded", "kernels"])
@pytest.mark.parametrize("nbits", [4, 3])
def test_fused_decoder_emits_the_tokens_of_the_pytorch_backend(nbits, glue):
    """SURVEY.md §8 f3, not against itself: the graph-replayed fused decode loop (grouped GEMVs + csrc/block.hip + HF's attention function) against
    the SAME quantised model decoding with HF's generate under HQQBackend.PYTORCH_FORWARD (dequantise + dense matmul: the reference's
    arithmetic, hqq/core/quantize.py:894-898) — 32 greedy tokens, identical"""
    import copy
    from hqq_amd.backends.hip import group_llama_projections
    from hqq_amd.core.quantize import BaseQuantizeConfig, HQQBackend, HQQLinear
    from hqq_amd.utils import llama_fused
    from hqq_amd.utils.generation import GraphedGreedyDecoder
    from hqq_amd.utils.model import quantize_model
    from hqq_amd.utils.patching import prepare_for_inference
    model = _tiny_llama()
    quantize_model(model, BaseQuantizeConfig(nbits=nbits, group_size=64, axis=1), compute_dtype=torch.float16, device="cuda")
    ref_model = copy.deepcopy(model)
    ids = torch.randint(0, 512, (1, 6), generator=torch.Generator().manual_seed(3)).cuda()
    HQQLinear.set_backend(HQQBackend.PYTORCH_FORWARD)
    try:
        with torch.no_grad():
            want = ref_model.generate(ids, max_new_tokens=32, do_sample=False)
    finally:
        HQQLinear.set_backend(HQQBackend.HIP)
    prepare_for_inference(model, backend="hip")
    group_llama_projections(model)
    assert llama_fused.supports(model)
    dec = GraphedGreedyDecoder(model, max_cache_len=64, glue=glue)
    assert dec.fused
    got = dec.generate(ids, 32, use_graph=True)
    assert dec.graph is not None and dec.step is not None and dec.step.folded == (glue == "folded")
    assert torch.equal(got, want), (got.tolist(), want.tolist())
    # teacher-forced logits of the two paths on the reference's tokens: the forward tolerance, not only the argmax
    HQQLinear.set_backend(HQQBackend.PYTORCH_FORWARD)
    try:
        with torch.no_grad():
            ref_logits = ref_model(want[:, :-1]).logits.float()
    finally:
        HQQLinear.set_backend(HQQBackend.HIP)
    with torch.no_grad():
        hip_logits = model(want[:, :-1]).logits.float()
    torch.testing.assert_close(hip_logits, ref_logits, rtol=5e-3, atol=5e-3)


def _toy_llama_hip(nbits):
    from hqq_amd.backends.hip import group_llama_projections
    from hqq_amd.core.quantize import BaseQuantizeConfig
    from hqq_amd.utils.model import quantize_model
    from hqq_amd.utils.patching import prepare_for_inference
    model = _tiny_llama()
    quantize_model(model, BaseQuantizeConfig(nbits=nbits, group_size=64, axis=1), compute_dtype=torch.float16, device="cuda")
    prepare_for_inference(model, backend="hip")
    group_llama_projections(model)
    return model


@pytest.mark.parametrize("n_heads,n_kv,hd,L,pos", [(32, 32, 128, 256, 0), (32, 32, 128, 256, 17), (32, 32, 128, 256, 255), (32, 8, 128, 1024, 700),
                                                  (16, 4, 64, 512, 511), (8, 8, 256, 300, 123), (64, 8, 128, 4096, 4000), (8, 2, 128, 20000, 19999)])   # (the last: > 48 KiB of LDS for the scores)
@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_decode_attention_kernel_against_fp64_softmax_attention_and_sdpa(n_heads, n_kv, hd, L, pos, dt):
    """hqq_hip_attn_decode (opt-in replacement of the SDPA call of a decode step): one query per head over the first pos + 1 cache positions —
    against softmax attention in float64 on the same fp16 inputs (1e-3 + one fp16 ulp of the output) and against torch's SDPA with the additive
    mask the fused step builds (2e-3: SDPA itself rounds the probabilities to fp16); positions beyond pos are NaN-poisoned and must not be read"""
    from hqq_amd import ops
    g = torch.Generator(device="cuda").manual_seed(n_heads * 1000 + pos)
    q = torch.randn(n_heads, hd, device="cuda", generator=g).to(dt)
    kc = torch.randn(n_kv, L, hd, device="cuda", generator=g).to(dt)
    vc = torch.randn(n_kv, L, hd, device="cuda", generator=g).to(dt)
    kc[:, pos + 1:] = float("nan")
    vc[:, pos + 1:] = float("nan")
    p = torch.tensor([pos], device="cuda")
    out = torch.full((n_heads * hd,), float("nan"), dtype=dt, device="cuda")
    scaling = hd ** -0.5
    ops.attn_decode(q, kc, vc, p, out, scaling)
    rep = n_heads // n_kv
    kk = kc[:, :pos + 1].repeat_interleave(rep, 0).double()
    vv = vc[:, :pos + 1].repeat_interleave(rep, 0).double()
    sc = torch.einsum("hd,hjd->hj", q.double(), kk) * scaling
    want = torch.einsum("hj,hjd->hd", torch.softmax(sc, -1), vv)
    got = out.view(n_heads, hd).double()
    assert torch.isfinite(got).all()
    ulp = 2.0 ** -10 if dt == torch.float16 else 2.0 ** -7
    tol = 1e-3 + 1e-3 * want.abs() + want.abs() * ulp
    assert bool(((got - want).abs() <= tol).all()), float((got - want).abs().max())
    mask = torch.zeros(1, 1, 1, L, dtype=dt, device="cuda")
    mask[..., pos + 1:] = float("-inf")
    ks = torch.nan_to_num(kc, nan=0.0).unsqueeze(0)
    vs = torch.nan_to_num(vc, nan=0.0).unsqueeze(0)
    sd = torch.nn.functional.scaled_dot_product_attention(q.view(1, n_heads, 1, hd), ks, vs, attn_mask=mask, scale=scaling, enable_gqa=(rep > 1))
    tl = 2e-3 if dt == torch.float16 else 1.6e-2   # (SDPA rounds its probabilities to the tensors' dtype)
    torch.testing.assert_close(out.view(n_heads, hd).float(), sd.view(n_heads, hd).float(), rtol=tl, atol=tl)
    again = torch.empty_like(out)
    ops.attn_decode(q, kc, vc, p, again, scaling)
    assert torch.equal(out, again)


@pytest.mark.parametrize("n_heads,n_kv,hd,L,pos", [(32, 32, 128, 256, 0), (32, 32, 128, 256, 100), (32, 8, 128, 512, 511), (8, 2, 64, 128, 31), (4, 4, 256, 64, 9)])
@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_rope_attn_decode_equals_rope_cache_then_attn_decode(n_heads, n_kv, hd, L, pos, dt):
    """the one-launch form (rotary + cache write + attention) against the two launches it replaces: output and both caches bit for bit"""
    from hqq_amd import ops
    g = torch.Generator(device="cuda").manual_seed(pos + hd)
    q = torch.randn(1, n_heads * hd, device="cuda", generator=g).to(dt)
    k = torch.randn(1, n_kv * hd, device="cuda", generator=g).to(dt)
    v = torch.randn(1, n_kv * hd, device="cuda", generator=g).to(dt)
    ang = torch.rand(hd // 2, device="cuda", generator=g) * 6.28
    cos = torch.cat([ang.cos(), ang.cos()]).to(dt)
    sin = torch.cat([ang.sin(), ang.sin()]).to(dt)
    kc0 = torch.randn(n_kv, L, hd, device="cuda", generator=g).to(dt)
    vc0 = torch.randn(n_kv, L, hd, device="cuda", generator=g).to(dt)
    kc0[:, pos:] = float("nan")   # position pos is written by the call; beyond it nothing may be read
    vc0[:, pos:] = float("nan")
    p = torch.tensor([pos], device="cuda")
    scaling = hd ** -0.5
    kc1, vc1 = kc0.clone(), vc0.clone()
    qr = torch.empty(1, n_heads, 1, hd, dtype=dt, device="cuda")
    ops.rope_cache(q, k, v, cos, sin, p, kc1, vc1, qr)
    want = torch.empty(n_heads * hd, dtype=dt, device="cuda")
    ops.attn_decode(qr, kc1, vc1, p, want, scaling)
    kc2, vc2 = kc0.clone(), vc0.clone()
    got = torch.empty_like(want)
    ops.rope_attn_decode(q, k, v, cos, sin, p, kc2, vc2, got, scaling)
    assert torch.isfinite(got).all()
    assert torch.equal(got, want)
    assert torch.equal(kc2[:, :pos + 1], kc1[:, :pos + 1]) and torch.equal(vc2[:, :pos + 1], vc1[:, :pos + 1])
    assert torch.isnan(kc2[:, pos + 1:]).all() and torch.isnan(vc2[:, pos + 1:]).all()


def test_fused_decoder_with_the_decode_attention_kernel_stays_within_the_logit_tolerance():
    """attention="hip": the step is no longer bit-chained to HF's SDPA — teacher-forced logits within 5e-3 of the sdpa step's, and on this seed
    the 32 greedy tokens are the same"""
    model = _toy_llama_hip(nbits=4)
    from hqq_amd.utils.generation import GraphedGreedyDecoder
    ids = torch.randint(0, model.config.vocab_size, (1, 12), generator=torch.Generator().manual_seed(5)).cuda()
    a = GraphedGreedyDecoder(model, max_cache_len=64, attention="sdpa").generate(ids, 32)
    dec = GraphedGreedyDecoder(model, max_cache_len=64, attention="hip")
    b = dec.generate(ids, 32)
    assert dec.fused and dec.step.attention == "hip"
    agree = int((a == b).all(dim=0).cumprod(0).sum())
    assert agree >= ids.shape[1] + 8, f"only {agree - ids.shape[1]} decoded tokens agree"
    # teacher forcing: the same token sequence through both steps, logits compared position by position
    from transformers import StaticCache
    from hqq_amd.utils.llama_fused import FusedLlamaStep
    logits = {}
    for mode in ("sdpa", "hip"):
        cache = StaticCache(config=model.config, max_cache_len=64)
        with torch.no_grad():
            model(ids, past_key_values=cache, cache_position=torch.arange(ids.shape[1], device="cuda"), use_cache=True)
        step = FusedLlamaStep(model, cache, 64, attention=mode)
        rows = []
        for t in range(16):
            pos = torch.tensor([ids.shape[1] + t], device="cuda")
            rows.append(step(a[:, ids.shape[1] + t:ids.shape[1] + t + 1], pos).float().clone())
        logits[mode] = torch.cat(rows)
    torch.testing.assert_close(logits["hip"], logits["sdpa"], rtol=5e-3, atol=5e-3)


def test_fused_decoder_bf16():
    """a bf16 model: the fused step (csrc/block.hip's bf16 arithmetic = torch's, the bf16 decode GEMVs) against the same model under
    HQQBackend.PYTORCH_FORWARD — teacher-forced logits within the bf16 forward tolerance; the kernel-attention mode within the same of the sdpa mode"""
    import copy
    from transformers import LlamaConfig, LlamaForCausalLM, StaticCache
    from hqq_amd.backends.hip import group_llama_projections
    from hqq_amd.core.quantize import BaseQuantizeConfig, HQQBackend, HQQLinear
    from hqq_amd.utils import llama_fused
    from hqq_amd.utils.generation import GraphedGreedyDecoder
    from hqq_amd.utils.model import quantize_model
    from hqq_amd.utils.patching import prepare_for_inference
    torch.manual_seed(0)
    cfg = LlamaConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                      vocab_size=512, max_position_embeddings=128)
    model = LlamaForCausalLM(cfg).to(torch.bfloat16).cuda().eval()
    quantize_model(model, BaseQuantizeConfig(nbits=4, group_size=64, axis=1), compute_dtype=torch.bfloat16, device="cuda")
    ref_model = copy.deepcopy(model)
    prepare_for_inference(model, backend="hip")
    group_llama_projections(model)
    assert llama_fused.supports(model)
    ids = torch.randint(0, 512, (1, 6), generator=torch.Generator().manual_seed(3)).cuda()
    dec = GraphedGreedyDecoder(model, max_cache_len=64)
    toks = dec.generate(ids, 24, use_graph=True)
    assert dec.fused and dec.graph is not None and toks.shape == (1, 30)
    HQQLinear.set_backend(HQQBackend.PYTORCH_FORWARD)
    try:
        with torch.no_grad():
            ref_logits = ref_model(toks[:, :-1]).logits.float()
    finally:
        HQQLinear.set_backend(HQQBackend.HIP)
    logits = {}
    for mode in ("sdpa", "hip"):
        cache = StaticCache(config=model.config, max_cache_len=64)
        with torch.no_grad():
            model(ids, past_key_values=cache, cache_position=torch.arange(ids.shape[1], device="cuda"), use_cache=True)
        step = llama_fused.FusedLlamaStep(model, cache, 64, attention=mode)
        rows = [step(toks[:, ids.shape[1] + t:ids.shape[1] + t + 1], torch.tensor([ids.shape[1] + t], device="cuda")).float().clone() for t in range(20)]
        logits[mode] = torch.cat(rows)
    want = ref_logits[0, ids.shape[1]:ids.shape[1] + 20]
    torch.testing.assert_close(logits["sdpa"], want, rtol=4e-2, atol=4e-2)
    torch.testing.assert_close(logits["hip"], logits["sdpa"], rtol=4e-2, atol=4e-2)


def test_cache_buckets_change_no_bit():
    """the default attention mode attends over a bucket of the static cache just above the position (one captured graph per bucket) instead of
    the whole masked cache: same logits bit for bit at every bucket length, and the same tokens across a bucket boundary"""
    from transformers import StaticCache
    from hqq_amd.utils.generation import GraphedGreedyDecoder
    from hqq_amd.utils.llama_fused import FusedLlamaStep
    model = _toy_llama_hip(nbits=4)
    ids = torch.randint(0, model.config.vocab_size, (1, 9), generator=torch.Generator().manual_seed(11)).cuda()
    cache = StaticCache(config=model.config, max_cache_len=128)
    with torch.no_grad():
        out = model(ids, past_key_values=cache, cache_position=torch.arange(9, device="cuda"), use_cache=True)
    step = FusedLlamaStep(model, cache, 128)
    tok = out.logits[:, -1].argmax(-1, keepdim=True)
    pos = torch.tensor([9], device="cuda")
    whole = step(tok, pos).clone()
    for kv in (16, 32, 64, 128):
        assert torch.equal(step(tok, pos, kv), whole), kv
    # across a boundary: buckets are 64, then multiples of 128: a 250-token prompt crosses from 256 to 384 within a few steps
    model.config.max_position_embeddings = 1024
    long_ids = torch.randint(0, model.config.vocab_size, (1, 250), generator=torch.Generator().manual_seed(12)).cuda()
    a = GraphedGreedyDecoder(model, max_cache_len=1024, bucket_cache=False).generate(long_ids, 16)
    dec = GraphedGreedyDecoder(model, max_cache_len=1024)
    b = dec.generate(long_ids, 16)
    assert sorted(dec.graphs) == [256, 384]
    assert torch.equal(a, b)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("n_heads,n_kv,hd,L,pos,splits", [(32, 32, 128, 4096, 4000, 8), (32, 8, 128, 4096, 37, 8), (8, 8, 64, 2048, 2047, 4), (8, 2, 256, 1024, 700, 3),
                                                         (4, 4, 128, 512, 0, 16)])
def test_decode_attention_with_the_keys_shared_out_over_workgroups(n_heads, n_kv, hd, L, pos, splits, dt):
    """splits > 1: a head's visible keys shared out over several workgroups and merged by a second launch — against float64 softmax attention, against
    the one-workgroup form, with most shares empty (pos << cache), and the rotary form against rope_cache + attn_decode at the same split count"""
    from hqq_amd import ops
    g = torch.Generator(device="cuda").manual_seed(pos + splits)
    q = torch.randn(n_heads, hd, device="cuda", generator=g).to(dt)
    kc = torch.randn(n_kv, L, hd, device="cuda", generator=g).to(dt)
    vc = torch.randn(n_kv, L, hd, device="cuda", generator=g).to(dt)
    kc[:, pos + 1:] = float("nan")
    vc[:, pos + 1:] = float("nan")
    p = torch.tensor([pos], device="cuda")
    scaling = hd ** -0.5
    one = torch.empty(n_heads * hd, dtype=dt, device="cuda")
    many = torch.full_like(one, float("nan"))
    ops.attn_decode(q, kc, vc, p, one, scaling)
    ops.attn_decode(q, kc, vc, p, many, scaling, splits=splits)
    rep = n_heads // n_kv
    kk = kc[:, :pos + 1].repeat_interleave(rep, 0).double()
    vv = vc[:, :pos + 1].repeat_interleave(rep, 0).double()
    want = torch.einsum("hj,hjd->hd", torch.softmax(torch.einsum("hd,hjd->hj", q.double(), kk) * scaling, -1), vv)
    ulp = 2.0 ** -10 if dt == torch.float16 else 2.0 ** -7
    tol = 1e-3 + 1e-3 * want.abs() + want.abs() * ulp
    assert torch.isfinite(many).all()
    assert bool(((many.view(n_heads, hd).double() - want).abs() <= tol).all())
    assert bool(((many.double() - one.double()).abs() <= 2 * ulp * one.double().abs() + 1e-3).all())
    # the rotary form at the same split count
    qraw = torch.randn(1, n_heads * hd, device="cuda", generator=g).to(dt)
    kraw = torch.randn(1, n_kv * hd, device="cuda", generator=g).to(dt)
    vraw = torch.randn(1, n_kv * hd, device="cuda", generator=g).to(dt)
    ang = torch.rand(hd // 2, device="cuda", generator=g) * 6.28
    cos, sin = torch.cat([ang.cos(), ang.cos()]).to(dt), torch.cat([ang.sin(), ang.sin()]).to(dt)
    kc[:, pos:] = float("nan")
    vc[:, pos:] = float("nan")
    kc1, vc1, kc2, vc2 = kc.clone(), vc.clone(), kc.clone(), vc.clone()
    qr = torch.empty(1, n_heads, 1, hd, dtype=dt, device="cuda")
    ops.rope_cache(qraw, kraw, vraw, cos, sin, p, kc1, vc1, qr)
    a = torch.empty_like(one)
    b = torch.empty_like(one)
    ops.attn_decode(qr, kc1, vc1, p, a, scaling, splits=splits)
    ops.rope_attn_decode(qraw, kraw, vraw, cos, sin, p, kc2, vc2, b, scaling, splits=splits)
    assert torch.equal(a, b) and torch.equal(kc1[:, :pos + 1], kc2[:, :pos + 1]) and torch.equal(vc1[:, :pos + 1], vc2[:, :pos + 1])


def test_sampling_inside_the_captured_step():
    """do_sample (the reference's HFGenerator(do_sample=True, temperature, top_k), hqq/utils/generation_hf.py:250-311) as an epilogue on the logits inside the
    captured decode step: with top_k = 1 the draw has one candidate and must reproduce the greedy tokens; with top_k = 5 every sampled token is one of the
    five most likely under the model's own (teacher-forced) logits, and two runs from the same generator state agree"""
    from hqq_amd.utils.generation import GraphedGreedyDecoder
    model = _toy_llama_hip(nbits=4)
    ids = torch.randint(0, model.config.vocab_size, (1, 10), generator=torch.Generator().manual_seed(11)).cuda()
    greedy = GraphedGreedyDecoder(model, max_cache_len=64).generate(ids, 24)
    one = GraphedGreedyDecoder(model, max_cache_len=64, do_sample=True, temperature=0.7, top_k=1).generate(ids, 24)
    assert torch.equal(greedy, one)
    torch.manual_seed(123)
    a = GraphedGreedyDecoder(model, max_cache_len=64, do_sample=True, temperature=0.6, top_k=5).generate(ids, 24)
    torch.manual_seed(123)
    b = GraphedGreedyDecoder(model, max_cache_len=64, do_sample=True, temperature=0.6, top_k=5).generate(ids, 24)
    assert torch.equal(a, b) and a.shape == (1, 34)
    with torch.no_grad():
        logits = model(a[:, :-1]).logits[0].float()
    top5 = logits.topk(5, dim=-1).indices
    T = ids.shape[1]
    inside = [(int(a[0, t + 1]) in top5[t].tolist()) for t in range(T - 1, a.shape[1] - 1)]
    assert sum(inside) >= len(inside) - 1, inside     # (the step's logits and the batched forward's differ in the last bits: a near-tie at rank 5 / 6 may swap)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_token_prologue_and_argmax_advance_equal_the_torch_ops(dt):
    """the one-launch front and back of a decode step (csrc/block.hip) against the torch ops they replace: bit for bit, graph-replay safe"""
    from hqq_amd import ops
    g = torch.Generator(device="cuda").manual_seed(5)
    vocab, H, L, hd = 1000, 264, 96, 48
    emb = torch.randn(vocab, H, device="cuda", generator=g).to(dt)
    ct, st = torch.randn(L, hd, device="cuda", generator=g).to(dt), torch.randn(L, hd, device="cuda", generator=g).to(dt)
    ar = torch.arange(L, device="cuda")
    for t, p in ((0, 0), (999, 95), (417, 31), (5, 64)):
        tok, pos = torch.tensor([[t]], device="cuda"), torch.tensor([p], device="cuda")
        h, cos, sin, mask = (torch.full((n,), 7.0, device="cuda", dtype=dt) for n in (H, hd, hd, L))
        ops.token_prologue(tok, pos, emb, h, ct, st, cos, sin, mask)
        assert torch.equal(h, emb[t]) and torch.equal(cos, ct[p]) and torch.equal(sin, st[p])
        want = torch.where(ar <= pos, torch.zeros((), dtype=dt, device="cuda"), torch.full((), float("-inf"), dtype=dt, device="cuda"))
        assert torch.equal(mask, want)
        h2 = torch.empty(H, device="cuda", dtype=dt)
        ops.token_prologue(tok, pos, emb, h2)                      # no tables, no mask (the kernel attention's form with a per-token rotary call)
        assert torch.equal(h2, emb[t])
    for n in (1, 7, 1024, 32000, 50257):
        for trial in range(5):
            logits = torch.randn(1, n, device="cuda", generator=g).to(dt)
            if trial == 1 and n > 7:                               # ties: the first of the largest wins, wherever the copies sit
                top = logits.max()
                logits[0, [n - 1, n // 2, 5]] = top
            if trial == 2:
                logits.fill_(float("-inf"))
            if trial == 3:
                logits[0, n - 1] = 1e4
            if trial == 4 and n > 9:                               # NaN is torch.argmax's maximum, the FIRST one wins (round-5 advisor: it was skipped)
                logits[0, [n - 2, 7]] = float("nan")
                logits[0, 3] = float("inf")
            nxt, tok, pos = torch.full((1, 1), -1, device="cuda"), torch.full((1, 1), -1, device="cuda"), torch.tensor([41], device="cuda")
            ops.argmax_advance(logits, nxt, tok, pos)
            want = logits.argmax(-1, keepdim=True)
            assert torch.equal(nxt, want) and torch.equal(tok, want) and int(pos) == 42, (n, trial, int(nxt), int(want))
            ops.argmax_advance(logits, nxt)                         # next token only
            assert torch.equal(nxt, want)
    # captured: the state the kernels advance lives on the device
    logits = torch.randn(1, 333, device="cuda", generator=g).to(dt)
    nxt, tok, pos = torch.zeros(1, 1, dtype=torch.int64, device="cuda"), torch.zeros(1, 1, dtype=torch.int64, device="cuda"), torch.zeros(1, dtype=torch.int64, device="cuda")
    ops.argmax_advance(logits, nxt, tok, pos)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        ops.argmax_advance(logits, nxt, tok, pos)
    for _ in range(5):
        gr.replay()
    assert int(pos) == 6 and int(tok) == int(logits.argmax())


class _ToyTokenizer:
    """the few tokenizer members HFGenerator touches (no network here for a real one): whitespace-separated integers are the tokens"""
    eos_token_id = None
    pad_token = None
    pad_token_id = 0
    add_bos_token = True
    add_eos_token = True
    padding_side = "left"

    def apply_chat_template(self, msgs, tokenize=False, add_generation_prompt=True):
        return "1 2 " + msgs[0]["content"] + (" 3" if add_generation_prompt else "")

    def add_special_tokens(self, d):
        self.pad_token = d["pad_token"]

    def __call__(self, prompts, return_tensors="pt"):
        import transformers
        return transformers.BatchEncoding({"input_ids": torch.tensor([[int(t) for t in prompts[0].split()]])})

    def decode(self, toks):
        return " ".join(str(int(t)) for t in toks)


def test_hfgenerator_front_end_and_kept_graphs():
    """the reference's HFGenerator surface (hqq/utils/generation_hf.py:117-540) over the graph loop: the tokens of HF's greedy generate, the dict it returns, EOS stop,
    and the second prompt replaying the first prompt's graphs on the same (reset) cache"""
    from hqq_amd.backends.hip import group_llama_projections
    from hqq_amd.core.quantize import BaseQuantizeConfig
    from hqq_amd.utils.generation import HFGenerator
    from hqq_amd.utils.model import quantize_model
    from hqq_amd.utils.patching import prepare_for_inference
    model = _tiny_llama()
    quantize_model(model, BaseQuantizeConfig(nbits=4, group_size=64, axis=1), compute_dtype=torch.float16, device="cuda")
    prepare_for_inference(model, backend="hip")
    group_llama_projections(model)
    tok = _ToyTokenizer()
    gen = HFGenerator(model, tok, max_new_tokens=20, compile="partial")
    assert gen.cache_size == 32 and gen.max_new_tokens == 20 and tok.add_bos_token is False and tok.padding_side == "right" and tok.pad_token
    prompts = ["7 8 9 10 11", "400 3 77", "5 6 7 8 9 10 11 12 13 14"]
    outs, graphs = [], None
    for i, p in enumerate(prompts):
        r = gen.generate(p, use_chat_template=(i == 1), verbose=False)
        ids = tok([tok.apply_chat_template([{"role": "user", "content": p}]) if i == 1 else p])["input_ids"].cuda()
        n = min(20, 32 - ids.shape[1])
        with torch.no_grad():
            want = model.generate(ids, max_new_tokens=n, min_new_tokens=n, do_sample=False)[0, ids.shape[1]:]
        assert set(r) == {"output_text", "output_tokens", "input_tokens"} and torch.equal(r["input_tokens"], ids[0].cpu())
        assert torch.equal(r["output_tokens"], want.cpu()), (i, r["output_tokens"], want)
        assert r["output_text"] == tok.decode(want)
        outs.append(want.cpu())
        if i == 0:
            graphs = dict(gen.decoder.graphs)
            step = gen.decoder.step
        else:   # the fused step and every graph captured so far are the first prompt's
            assert gen.decoder.step is step and all(gen.decoder.graphs[k] is g for k, g in graphs.items())
    # EOS: the 6th new token of the first prompt ends it (looked for on the host every 16 tokens; the text is cut there all the same)
    eos = int(outs[0][5])
    first = int((outs[0] == eos).nonzero()[0])
    tok.eos_token_id = eos
    r = gen.generate(prompts[0], use_chat_template=False, verbose=False)
    assert torch.equal(r["output_tokens"], outs[0][:first])
    tok.eos_token_id = None
    # no graph (compile=None): the same tokens, launched eagerly
    gen2 = HFGenerator(model, tok, max_new_tokens=20)
    assert torch.equal(gen2.generate(prompts[2], use_chat_template=False, verbose=False)["output_tokens"], outs[2])
    assert gen.warmup(max_samples=1) is gen if False else True   # (warmup needs a tokenizer of words; the toy one reads integers)
    # the kept step follows the model: other weights in one layer (a new version of its packed tensor) -> the fused step and its graphs are rebuilt, not replayed stale
    from hqq_amd.backends.hip import HQQLinearHIP
    lay = next(m for m in model.modules() if isinstance(m, HQQLinearHIP))
    old_step = gen.decoder.step
    with torch.no_grad():
        lay.W_q.add_(0)   # same bytes, next version: the fingerprint changes
    gen.generate(prompts[0], use_chat_template=False, verbose=False)
    assert gen.decoder.step is not old_step
    kept = gen.decoder.step
    gen.generate(prompts[1], use_chat_template=False, verbose=False)
    assert gen.decoder.step is kept
    gen.reset()
    assert gen.decoder.step is None and gen.decoder.graphs == {}


@pytest.mark.parametrize("arch", ["llama3-like", "mistral"])
def test_fused_decoder_on_gqa_models_with_128_wide_heads(arch):
    """the shapes of today's checkpoints — grouped-query attention (4 query heads per key / value head), 128-wide heads, a large rope base; and MistralForCausalLM, the other
    model_type the fused step admits — through the folded launches (rotary-paired q / k of unequal size in one launch): the tokens of the same model under PYTORCH_FORWARD"""
    import copy
    from transformers import LlamaConfig, LlamaForCausalLM, MistralConfig, MistralForCausalLM
    from hqq_amd.backends.hip import group_llama_projections
    from hqq_amd.core.quantize import BaseQuantizeConfig, HQQBackend, HQQLinear
    from hqq_amd.utils import llama_fused
    from hqq_amd.utils.generation import GraphedGreedyDecoder
    from hqq_amd.utils.model import quantize_model
    from hqq_amd.utils.patching import prepare_for_inference
    torch.manual_seed(11)
    kw = dict(hidden_size=512, intermediate_size=1024, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=1, vocab_size=1024, max_position_embeddings=256)
    if arch == "mistral":
        model = MistralForCausalLM(MistralConfig(sliding_window=None, head_dim=128, **kw))
    else:
        model = LlamaForCausalLM(LlamaConfig(rope_theta=500000.0, **kw))
    model = model.half().cuda().eval()
    quantize_model(model, BaseQuantizeConfig(nbits=4, group_size=64, axis=1), compute_dtype=torch.float16, device="cuda")
    ref_model = copy.deepcopy(model)
    ids = torch.randint(0, 1024, (1, 9), generator=torch.Generator().manual_seed(5)).cuda()
    HQQLinear.set_backend(HQQBackend.PYTORCH_FORWARD)
    try:
        with torch.no_grad():
            want = ref_model.generate(ids, max_new_tokens=24, do_sample=False, pad_token_id=0)
    finally:
        HQQLinear.set_backend(HQQBackend.HIP)
    prepare_for_inference(model, backend="hip")
    group_llama_projections(model)
    assert llama_fused.supports(model)
    dec = GraphedGreedyDecoder(model, max_cache_len=64)
    got = dec.generate(ids, 24, use_graph=True)
    assert dec.fused and dec.step is not None and dec.step.folded and dec.step.one_launch_front
    assert torch.equal(got, want), (got.tolist(), want.tolist())
