#!/usr/bin/env python3
"""The fused decode loop on a random Llama-shaped model with NB decoder blocks, T tokens, for rocprofv3 --kernel-trace (development aid; needs
an MI355X): the kernel count per token / per decoder block comes from runs that differ in T and in NB.  The trace does not list the kernels of a
hipGraph replay, so the step runs stream-ordered here (the same launches the graph captures).   python tools/e2e_kernels.py NB T [sdpa|hip]"""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from transformers import LlamaConfig, LlamaForCausalLM  # noqa: E402

from hqq_amd.core.quantize import BaseQuantizeConfig  # noqa: E402
from hqq_amd.utils.generation import GraphedGreedyDecoder  # noqa: E402
from hqq_amd.utils.model import quantize_model  # noqa: E402
from hqq_amd.utils.patching import prepare_for_inference  # noqa: E402

NB, T = int(sys.argv[1]), int(sys.argv[2])
ATT = sys.argv[3] if len(sys.argv) > 3 else "sdpa"
torch.manual_seed(0)
cfg = LlamaConfig(hidden_size=4096, intermediate_size=11008, num_hidden_layers=NB, num_attention_heads=32, num_key_value_heads=32, vocab_size=32000,
                  max_position_embeddings=512, torch_dtype=torch.float16)
torch.set_default_dtype(torch.float16)
with torch.device("cuda"):
    model = LlamaForCausalLM(cfg).eval()
torch.set_default_dtype(torch.float32)
quantize_model(model, BaseQuantizeConfig(nbits=4, group_size=64, axis=1), compute_dtype=torch.float16, device="cuda")
prepare_for_inference(model, backend="hip")
from hqq_amd.backends.hip import group_llama_projections  # noqa: E402
group_llama_projections(model)
dec = GraphedGreedyDecoder(model, max_cache_len=128, attention=ATT)
ids = torch.randint(0, 32000, (1, 16), device="cuda")
out = dec.generate(ids, max_new_tokens=T, use_graph=False)
torch.cuda.synchronize()
print("fused:", dec.fused, "tokens:", out.shape)
