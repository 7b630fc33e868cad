#!/usr/bin/env python3
"""End-to-end decode rate of the fused loop against the static cache's length, both attention modes (development aid; needs an MI355X):
HF's attention function attends over the WHOLE static cache behind a mask (cost ~ cache length), the decode-attention kernel over pos + 1 keys.
    python tools/e2e_cache_len.py [cache_len ...] [prompt=N]     (prompt: tokens prefilled before the timed steps, default 16)"""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from transformers import LlamaConfig, LlamaForCausalLM  # noqa: E402

from hqq_amd.backends.hip import group_llama_projections  # noqa: E402
from hqq_amd.core.quantize import BaseQuantizeConfig  # noqa: E402
from hqq_amd.utils.generation import GraphedGreedyDecoder  # noqa: E402
from hqq_amd.utils.model import quantize_model  # noqa: E402
from hqq_amd.utils.patching import prepare_for_inference  # noqa: E402

PROMPT = next((int(v.split("=")[1]) for v in sys.argv[1:] if v.startswith("prompt=")), 16)
lens = [int(v) for v in sys.argv[1:] if not v.startswith("prompt=")] or [256, 1024, 4096]
torch.manual_seed(0)
cfg = LlamaConfig(hidden_size=4096, intermediate_size=11008, num_hidden_layers=32, num_attention_heads=32, num_key_value_heads=32, vocab_size=32000,
                  max_position_embeddings=8192)
torch.set_default_dtype(torch.float16)
with torch.device("cuda"):
    model = LlamaForCausalLM(cfg).eval()
torch.set_default_dtype(torch.float32)
quantize_model(model, BaseQuantizeConfig(nbits=4, group_size=64, axis=1), compute_dtype=torch.float16, device="cuda")
prepare_for_inference(model, backend="hip")
group_llama_projections(model)
ids = torch.randint(0, 32000, (1, PROMPT), device="cuda")
for L in lens:
    row = []
    for mode in ("sdpa", "hip"):
        dec = GraphedGreedyDecoder(model, max_cache_len=L, attention=mode)
        r = dec.benchmark(ids, new_tokens=48, warmup=6)
        row.append(f"{mode}: {r['tok_s']:7.1f} tok/s ({r['ms_per_token']:.3f} ms)")
        del dec
    print(f"static cache of {L:5d} positions, {PROMPT}-token prompt, positions {PROMPT + 8}..{PROMPT + 55} timed:  " + "   ".join(row))
