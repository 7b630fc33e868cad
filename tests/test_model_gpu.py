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
