"""Whole-model caller of the per-layer quantiser: replace the decoder linears of a torch model by HQQLinear.

Mirrors what `BaseHQQModel.quantize_model` + a `*Patch.get_linear_tags` class do in the reference
(hqq/models/base.py:266-401, hqq/models/hf/llama.py:9-64) without the model zoo: linears are found by the qualified-name
suffix ("tag"), each tag can carry its own quant_config, everything else in the model is left alone.  The solver is per layer
and needs no exchange, so with several GPUs the decoder blocks are simply dealt out over `devices` (layer-parallel quantise)."""
from __future__ import annotations

import re
from typing import Dict, Iterable, Optional, Sequence, Union

import torch
from torch import nn

from ..core.quantize import HQQLinear

# hqq/models/hf/llama.py:12-21 — the seven quantised linears of a Llama / Mistral style decoder block
LLAMA_LINEAR_TAGS = ["self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj",
                     "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj"]


def _block_index(name: str) -> int:
    m = re.search(r"\.layers\.(\d+)\.", "." + name)
    return int(m.group(1)) if m else 0


def quantize_model(model: nn.Module, quant_config: Union[dict, Dict[str, Optional[dict]]], compute_dtype: torch.dtype = torch.float16,
                   device: Union[str, Sequence[str]] = "cuda", linear_tags: Iterable[str] = LLAMA_LINEAR_TAGS, verbose: bool = False) -> nn.Module:
    """quant_config: one BaseQuantizeConfig dict for every tag, or {tag: config-or-None} (None = leave that linear alone).
    device: one device, or a list — decoder block i goes to devices[i % len(devices)]."""
    tags = list(linear_tags)
    per_tag = quant_config if (isinstance(quant_config, dict) and "weight_quant_params" not in quant_config) else {t: quant_config for t in tags}
    devices = [device] if isinstance(device, (str, torch.device)) else list(device)
    todo = []
    for name, mod in model.named_modules():
        if isinstance(mod, nn.Linear):
            tag = next((t for t in tags if name.endswith(t)), None)
            if tag is not None and per_tag.get(tag) is not None:
                todo.append((name, tag))
    for name, tag in todo:
        parent_name, _, child = name.rpartition(".")
        parent = model.get_submodule(parent_name) if parent_name else model
        lin = getattr(parent, child)
        dev = devices[_block_index(name) % len(devices)]
        q = HQQLinear(lin, per_tag[tag], compute_dtype=compute_dtype, device=dev, del_orig=True)
        q.name = name
        setattr(parent, child, q)
        if verbose:
            print(f"quantized {name} -> {dev}")
    model.hqq_quantized = True
    return model
