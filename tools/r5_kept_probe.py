#!/usr/bin/env python3
"""lab: does a decoder that kept its cache / fused step / graphs (after benchmark()) emit what a fresh one emits?"""
import sys, torch
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from transformers import LlamaConfig, LlamaForCausalLM
from hqq_amd.backends.hip import group_llama_projections
from hqq_amd.core.quantize import BaseQuantizeConfig
from hqq_amd.utils.generation import GraphedGreedyDecoder
from hqq_amd.utils.model import quantize_model
from hqq_amd.utils.patching import prepare_for_inference
torch.manual_seed(0)
cfg = LlamaConfig(hidden_size=512, intermediate_size=1024, num_hidden_layers=4, num_attention_heads=8, num_key_value_heads=8, vocab_size=2048, max_position_embeddings=512)
model = LlamaForCausalLM(cfg).half().cuda().eval()
quantize_model(model, BaseQuantizeConfig(nbits=4, group_size=64, axis=1), compute_dtype=torch.float16, device="cuda")
prepare_for_inference(model, backend="hip")
group_llama_projections(model)
ids = torch.randint(0, 2048, (1, 16), generator=torch.Generator().manual_seed(3)).cuda()
for att in ("sdpa", "hip"):
    fresh = GraphedGreedyDecoder(model, max_cache_len=256, attention=att).generate(ids, 24)
    d = GraphedGreedyDecoder(model, max_cache_len=256, attention=att)
    a = d.generate(ids, 24)
    b = d.generate(ids, 24)
    d.benchmark(ids, new_tokens=64, warmup=8)
    c = d.generate(ids, 24)
    d2 = GraphedGreedyDecoder(model, max_cache_len=256, attention=att)
    d2.benchmark(ids, new_tokens=64, warmup=8)
    e = d2.generate(ids, 8)
    f = lambda x, y: int((x[0, 16:16 + min(x.shape[1], y.shape[1]) - 16] == y[0, 16:16 + min(x.shape[1], y.shape[1]) - 16]).to(torch.int32).cumprod(0).sum())
    print(att, "fresh==first", f(fresh, a), "second", f(fresh, b), "after benchmark", f(fresh, c), "benchmark then 8", f(fresh, e), flush=True)
