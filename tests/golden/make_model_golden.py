#!/usr/bin/env python3
"""Model-level fixture: a 2-block toy Llama quantised by the REFERENCE's own whole-model caller,
AutoHQQHFModel.quantize_model (hqq/models/base.py:266-401 through hqq/models/hf/base.py), on the CPU of the authoring container.

    python tests/golden/make_model_golden.py      # needs /root/reference; writes tests/golden/model_llama2blk_<nbits>b.npz

Stored per quantised linear: its qualified name, the sha256 of the packed W_q, of zero and of scale (float32, as the solver returns them
before the cast to the compute dtype), and the sha256 of the source weight (so that a test can tell an RNG difference from a mismatch).
tests/test_model_gpu.py rebuilds the same model from the same seed and checks hqq_amd.utils.model.quantize_model against it."""
import hashlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("HQQ_REFERENCE", "/root/reference")


def sha(t):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(t).tobytes()).hexdigest().encode(), dtype=np.uint8)


def tiny_llama():
    from transformers import LlamaConfig, LlamaForCausalLM
    torch.manual_seed(0)
    cfg = LlamaConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=4,
                      vocab_size=512, max_position_embeddings=128)
    return LlamaForCausalLM(cfg).float().eval()


def main():
    if not os.path.isdir(REF):
        sys.exit(f"{REF} not found")
    stub = types.ModuleType("termcolor")
    stub.colored = lambda t, *a, **k: t
    sys.modules.setdefault("termcolor", stub)
    sys.path.insert(0, REF)
    from hqq.core.quantize import BaseQuantizeConfig, HQQLinear
    from hqq.models.hf.base import AutoHQQHFModel
    for nbits in (4, 2):
        model = tiny_llama()
        src = {n: sha(m.weight.detach().numpy()) for n, m in model.named_modules() if isinstance(m, torch.nn.Linear)}
        cfg = BaseQuantizeConfig(nbits=nbits, group_size=64, axis=1)
        AutoHQQHFModel.quantize_model(model, quant_config=cfg, compute_dtype=torch.float32, device="cpu")
        out = {}
        names = []
        for n, m in model.named_modules():
            if isinstance(m, HQQLinear):
                names.append(n)
                out[f"Wq__{n}"] = sha(m.W_q.data.numpy())
                out[f"zero__{n}"] = sha(m.meta["zero"].float().numpy())
                out[f"scale__{n}"] = sha(m.meta["scale"].float().numpy())
                out[f"src__{n}"] = src[n]
                out[f"shape__{n}"] = np.array(m.meta["shape"], dtype=np.int64)
        out["names"] = np.frombuffer("\n".join(names).encode(), dtype=np.uint8)
        out["untouched"] = np.frombuffer("\n".join(n for n, m in model.named_modules() if isinstance(m, torch.nn.Linear)).encode(), dtype=np.uint8)
        path = os.path.join(HERE, f"model_llama2blk_{nbits}b.npz")
        np.savez_compressed(path, **out)
        print(path, len(names), "quantised linears; left alone:", bytes(out["untouched"]).decode().split("\n"))


if __name__ == "__main__":
    main()
