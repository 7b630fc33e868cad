"""One decode step of a Llama-type HF model whose decoder linears are HQQLinearHIP layers, with the steps either side of the fused
GEMVs fused too (the loop the reference's headline tok/s is measured on: hqq/utils/generation_hf.py:117-540, Readme.md:153).

HF's decoder block around the seven linears is ~25 eager kernels per block at batch 1 (RMSNorm 5-6, rotary 8, cache update 2, SiLU * up 2,
residual adds 2, ...): 79 % of a token once the linears are fused.  Here a block is
    add_rmsnorm -> q|k|v (one grouped GEMV) -> rope_cache -> attention (HF's own attention function on the static cache) -> o ->
    add_rmsnorm (the residual add of o rides in it) -> gate|up (one grouped GEMV) -> silu_mul -> down (its residual add rides in the next block's add_rmsnorm)
= 8 launches + the attention's (glue="kernels").  Round 5 folds the glue into the launches either side of it (glue="folded", the default where
csrc/gemv_block.hip covers the model): q|k|v with the RMSNorm in its prologue -> rope_cache -> attention -> o with the residual add in its epilogue ->
ONE paired gate|up layer (RMSNorm prologue, SiLU * up epilogue) -> down with the residual add in its epilogue; and with q / k in the rotary-paired row order
(ops.rotary_pair_layout) the rotary embedding and the cache write ride in the q|k|v launch's epilogue too = 4 launches + the attention's.  The three glue kernels (csrc/block.hip) restate the HF modules rounding for rounding and the attention is
HF's function on HF's cache tensors, so the step emits the same tokens as `model(...)` does on the same kernels — and as the same model
under HQQBackend.PYTORCH_FORWARD does on the reference's arithmetic (tests/test_model_gpu.py).

Only what the step needs is taken from the model: module weights and the HF StaticCache's tensors are used in place (nothing is copied).
"""
from __future__ import annotations

import torch
from torch import Tensor

from .. import ops
from ..backends.hip import HQQLinearHIP, _GroupedMember


def _hip(layer):
    return layer.layer if isinstance(layer, _GroupedMember) else layer


def arch_supported(model) -> bool:
    """The allow-list half of supports(): the step restates LlamaDecoderLayer's arithmetic (transformers models/llama, models/mistral) and nothing
    else.  Models that merely LOOK like it (same attribute names) would decode wrong tokens without an error: Qwen3 (per-head q_norm / k_norm),
    Granite (residual / embedding / logits / attention multipliers), Gemma (soft-capping, (1 + w) norms), Cohere, OLMo ..."""
    try:
        cfg = model.config
        if getattr(cfg, "model_type", None) not in ("llama", "mistral"):
            return False
        if getattr(cfg, "sliding_window", None) or getattr(cfg, "attn_logit_softcapping", None) or getattr(cfg, "final_logit_softcapping", None):
            return False
        if getattr(cfg, "attention_bias", False) or getattr(cfg, "mlp_bias", False):
            return False
        for odd in ("residual_multiplier", "embedding_multiplier", "logits_scaling", "attention_multiplier"):
            if getattr(cfg, odd, None) not in (None, 1, 1.0):
                return False
        for blk in model.model.layers:
            at = blk.self_attn
            if any(hasattr(at, n) for n in ("q_norm", "k_norm", "qk_norm", "sinks")) or getattr(at, "sliding_window", None):
                return False
            if type(getattr(blk.mlp, "act_fn", None)).__name__ not in ("SiLUActivation", "SiLU"):
                return False
        return True
    except AttributeError:
        return False


def supports(model) -> bool:
    """a LlamaForCausalLM-shaped model of an allow-listed architecture (arch_supported) — model.model.layers[*].self_attn.{q,k,v,o}_proj,
    .mlp.{gate,up,down}_proj, RMSNorm without bias —, fp16 or bf16, every decoder linear an HQQLinearHIP without bias whose group can share one launch"""
    if not arch_supported(model):
        return False
    try:
        inner = model.model
        dt = inner.norm.weight.dtype
        if dt not in (torch.float16, torch.bfloat16):
            return False
        layers = inner.layers
        if not hasattr(inner, "rotary_emb") or not hasattr(inner, "embed_tokens") or not hasattr(model, "lm_head"):
            return False
        for blk in layers:
            at, mlp = blk.self_attn, blk.mlp
            lin = [_hip(getattr(at, n)) for n in ("q_proj", "k_proj", "v_proj", "o_proj")] + [_hip(getattr(mlp, n)) for n in ("gate_proj", "up_proj", "down_proj")]
            if not all(isinstance(L, HQQLinearHIP) and L.bias is None and L.compute_dtype == dt and L.W_q.is_cuda for L in lin):
                return False
            if len({(L.nbits, L.group_size, L.w3s) for L in lin[:3]}) != 1 or len({(L.nbits, L.group_size, L.w3s) for L in lin[4:6]}) != 1:
                return False
            if not ops.decode_covers(dt, 1, lin[0].out_features, lin[0].in_features, lin[0].group_size, lin[0].nbits) and not lin[0].w3s:
                return False
            if type(getattr(mlp, "act_fn", None)).__name__ not in ("SiLUActivation", "SiLU"):
                return False
            for nrm in (blk.input_layernorm, blk.post_attention_layernorm):
                if nrm.weight.dtype != dt or nrm.weight.shape[0] % 8:
                    return False
        return True
    except AttributeError:
        return False


class FusedLlamaStep:
    """decode step t -> logits of token t + 1, on the model's own weights and an HF StaticCache that a prefill has filled"""

    def __init__(self, model, cache, max_cache_len: int, attention: str = "sdpa", glue: str = "auto"):
        """attention: "sdpa" — HF's own attention function on the cache tensors (the step then emits the tokens `model(...)` would);
        "hip" — csrc/block.hip's decode-attention kernel (one query per head, fp32 softmax): within rounding of SDPA, not bit-identical,
        3-4 us instead of 12-15 per block.
        glue: "folded" — RMSNorm in the q|k|v / gate|up launches' prologue, the residual adds in o's / down's epilogue, SiLU * up in the epilogue of ONE
        paired gate|up layer (csrc/gemv_block.hip: 4 launches + rotary / attention per block; costs a second copy of gate / up's packed levels in the
        paired layout); "kernels" — round 4's separate glue kernels (9 launches per block); "auto": folded where hqq_hip_gemv_block covers the model."""
        from transformers.modeling_utils import ALL_ATTENTION_FUNCTIONS
        from transformers.models.llama.modeling_llama import eager_attention_forward
        self.model = model
        inner = model.model
        self.inner = inner
        cfg = model.config
        self.device = inner.embed_tokens.weight.device
        self.dt = dt = inner.norm.weight.dtype   # fp16 or bf16 (supports())
        self.n_heads = cfg.num_attention_heads
        self.n_kv = getattr(cfg, "num_key_value_heads", None) or cfg.num_attention_heads
        self.hd = getattr(cfg, "head_dim", None) or cfg.hidden_size // cfg.num_attention_heads
        self.H = cfg.hidden_size
        self.L = max_cache_len
        self.attn_fn = ALL_ATTENTION_FUNCTIONS.get_interface(cfg._attn_implementation, eager_attention_forward)
        if attention not in ("sdpa", "hip"):
            raise ValueError("attention: 'sdpa' or 'hip'")
        if attention == "hip" and (self.hd not in (64, 128, 256) or getattr(cfg, "sliding_window", None) or getattr(cfg, "attn_logit_softcapping", None)
                                   or max_cache_len > 30000):
            raise ValueError("hqq_amd: the decode-attention kernel covers plain softmax attention with head_dim 64 / 128 / 256 and caches of <= 30000 positions")
        self.attention = attention
        if glue not in ("auto", "folded", "kernels"):
            raise ValueError("glue: 'auto', 'folded' or 'kernels'")
        lins = [_hip(getattr(b.self_attn, n)) for b in inner.layers for n in ("q_proj", "o_proj")] + [_hip(getattr(b.mlp, n)) for b in inner.layers for n in ("gate_proj", "down_proj")]
        can_fold = all(ops.block_covers(dt, L.in_features, L.group_size, L.nbits, L.w3s, norm=(i % 2 == 0)) for i, L in enumerate(lins)) and \
            all(_hip(b.mlp.gate_proj).out_features == _hip(b.mlp.up_proj).out_features for b in inner.layers) and not (ops._default_opts & ops.OPT_FACTORED)
        if glue == "folded" and not can_fold:
            raise ValueError("hqq_amd: glue='folded' needs fp16 / bf16 layers of 4 / 2 bits or the 3-bit stream layout, group_size 64, hidden size <= 8192")
        self.folded = can_fold and glue != "kernels"
        # The folded step keeps re-laid-out COPIES of q, k (rotary-paired rows) and of gate | up (one paired layer) beside the layers' own tensors: about
        # +60 % of the decoder's linear-weight bytes (7B at 4 bits: +1.9 GB).  glue="auto" takes them only when they fit with room to spare; a model that filled
        # the GPU before keeps round 4's separate glue kernels (no copy) instead of running out of memory here (round-5 advisor).  glue="folded" insists.
        self.extra_weight_bytes = 0
        if self.folded:
            def _nbytes(L):
                return L.W_q.numel() * L.W_q.element_size() + 2 * L.scale.numel() * L.scale.element_size()
            need = sum(_nbytes(_hip(getattr(b.self_attn, n))) for b in inner.layers for n in ("q_proj", "k_proj")) + \
                sum(_nbytes(_hip(getattr(b.mlp, n))) for b in inner.layers for n in ("gate_proj", "up_proj"))
            free_b = torch.cuda.mem_get_info(self.device)[0] if self.device.type == "cuda" else need * 4
            if glue == "auto" and free_b < need + need // 4 + (1 << 30):
                import warnings
                warnings.warn(f"hqq_amd: the folded decode step needs {need / 1e9:.2f} GB for its paired layer copies, {free_b / 1e9:.2f} GB are free: "
                              "falling back to the separate glue kernels (glue='kernels')")
                self.folded = False
            else:
                self.extra_weight_bytes = need
        self.blocks = []
        dev = self.device
        for li, blk in enumerate(inner.layers):
            at, mlp = blk.self_attn, blk.mlp
            q, k, v, o = (_hip(getattr(at, n)) for n in ("q_proj", "k_proj", "v_proj", "o_proj"))
            g, u, d = (_hip(getattr(mlp, n)) for n in ("gate_proj", "up_proj", "down_proj"))
            lay = cache.layers[li]
            if not getattr(lay, "is_initialized", False) or lay.keys.shape[0] != 1 or lay.keys.shape[2] != max_cache_len or not lay.keys.is_contiguous():
                raise ValueError("hqq_amd: the fused decode step needs an HF StaticCache that a batch-1 prefill has initialised")
            self.blocks.append({
                "attn": at, "n1": blk.input_layernorm, "n2": blk.post_attention_layernorm,
                "qkv": [(L.W_q, L.scale, L.zero, None, L.out_features) for L in (q, k, v)], "qkv_opts": self._gopts((q, k, v)), "qkv_nbits": q.nbits, "qkv_gs": q.group_size, "gu_gs": g.group_size,
                "o": o, "gu": [(L.W_q, L.scale, L.zero, None, L.out_features) for L in (g, u)], "gu_opts": self._gopts((g, u)), "gu_nbits": g.nbits, "d": d,
                "kc": lay.keys[0], "vc": lay.values[0], "len": lay.cumulative_length,
                # outputs of the launches (static addresses: the step is captured in a hipGraph)
                "q": torch.empty(1, q.out_features, dtype=dt, device=dev), "k": torch.empty(1, k.out_features, dtype=dt, device=dev),
                "v": torch.empty(1, v.out_features, dtype=dt, device=dev), "qr": torch.empty(1, self.n_heads, 1, self.hd, dtype=dt, device=dev),
                "g": torch.empty(1, g.out_features, dtype=dt, device=dev), "u": torch.empty(1, u.out_features, dtype=dt, device=dev),
                "a": torch.empty(1, g.out_features, dtype=dt, device=dev),
            })
            if self.folded and attention != "hip" and self.hd % 2 == 0:
                # q and k in the rotary-paired row order (ops.rotary_pair_layout): the q|k|v launch's epilogue applies the rotary embedding and writes the cache
                # (the kernel attention folds the rotary embedding into the attention launch instead: it keeps the natural order)
                def _sub_ok(t, L_):
                    if dt != torch.float16:
                        return False
                    return ops.w3s_meta_scalable(t[1], t[2], t[3], L_.in_features) if L_.w3s else ops.meta_scalable(t[1], t[2], t[3], L_.in_features, L_.group_size, L_.nbits)
                qp = ops.rotary_pair_layout((q.W_q, q.scale, q.zero, q.out_features), q.in_features, q.group_size, q.nbits, self.hd, w3s=q.w3s)
                kp = ops.rotary_pair_layout((k.W_q, k.scale, k.zero, k.out_features), k.in_features, k.group_size, k.nbits, self.hd, w3s=k.w3s)
                sub = _sub_ok(qp, q) and _sub_ok(kp, k) and bool(v.opts & ops.OPT_META_SCALABLE)   # (the permutation moves rows between slabs: checked again)
                self.blocks[-1]["qkv_rope"] = [qp, kp, (v.W_q, v.scale, v.zero, v.out_features)]
                self.blocks[-1]["qkv_rope_opts"] = ops.layer_opts((ops.OPT_META_SCALABLE if sub else 0) | (ops.OPT_W3S if q.w3s else 0))
            if self.folded:   # gate|up as ONE paired layer: a packed row holds gate row n and up row n (ops.pair_layers); the layers' own tensors stay as they are
                pair = ops.pair_layers((g.W_q, g.scale, g.zero, g.out_features), (u.W_q, u.scale, u.zero, u.out_features), g.in_features, g.group_size, g.nbits, w3s=g.w3s)
                # the three-op rebuild's condition depends on the slab a row sits in (J = 9 - the slab's bit offset), and the pairing moves rows between
                # slabs: checked again on the paired tensors, never inherited from the two layers
                if dt != torch.float16:
                    sub = False
                elif g.w3s:
                    sub = ops.w3s_meta_scalable(pair[1], pair[2], pair[3], g.in_features)
                else:
                    sub = ops.meta_scalable(pair[1], pair[2], pair[3], g.in_features, g.group_size, g.nbits)
                self.blocks[-1]["gu_pair"] = [pair]
                self.blocks[-1]["gu_pair_opts"] = ops.layer_opts((ops.OPT_META_SCALABLE if sub else 0) | (ops.OPT_W3S if g.w3s else 0))
        self.h = torch.empty(1, self.H, dtype=dt, device=dev)       # the residual stream
        self.xn = torch.empty(1, self.H, dtype=dt, device=dev)      # its normalised copy, input of the next linears
        self.delta = torch.empty(1, self.H, dtype=dt, device=dev)   # output of o / down, added by the next add_rmsnorm
        self.att = torch.empty(1, self.n_heads * self.hd, dtype=dt, device=dev)   # attention output (attention="hip")
        self.attn_ws = {}                                                          # splits -> record buffer of the split attention launches
        # the causal mask of one query over the static cache, in the additive form SDPA turns a boolean mask into on every call
        # (where(mask, 0, -inf) in the query dtype): built once per token here instead of once per decoder block inside the attention function
        self.mask = torch.zeros(1, 1, 1, max_cache_len, dtype=dt, device=dev)
        self.ar = torch.arange(max_cache_len, device=dev)
        # cos / sin of every cache position, from the model's own rotary module called once (elementwise in the position: the rows equal what a
        # per-token call returns); rope types whose frequencies depend on the sequence length ("dynamic", "longrope") keep the per-token call
        self.cos_tab = self.sin_tab = None
        if getattr(inner.rotary_emb, "rope_type", "default") in ("default", "linear", "llama3", "yarn") and \
                max_cache_len <= getattr(cfg, "max_position_embeddings", max_cache_len):
            with torch.no_grad():
                c, s_ = inner.rotary_emb(torch.empty(1, 1, self.H, dtype=dt, device=dev), self.ar.view(1, -1))
            self.cos_tab, self.sin_tab = c[0].contiguous(), s_[0].contiguous()   # [max_cache_len, hd]
        self.zero = torch.zeros((), dtype=dt, device=dev)
        self.ninf = torch.full((), float("-inf"), dtype=dt, device=dev)
        # the front of a step as one launch (ops.token_prologue) where it is a plain table lookup: an ordinary nn.Embedding in the compute dtype and precomputed rotary tables
        emb = inner.embed_tokens
        self.one_launch_front = bool(glue != "kernels" and self.cos_tab is not None and type(emb) is torch.nn.Embedding and emb.max_norm is None and emb.weight.dtype == dt
                                     and emb.weight.is_contiguous() and emb.weight.device == self.h.device and self.H % 8 == 0 and dt in (torch.float16, torch.bfloat16))
        self.cos_v = torch.empty(self.hd, dtype=dt, device=dev)
        self.sin_v = torch.empty(self.hd, dtype=dt, device=dev)

    @staticmethod
    def _gopts(Ls) -> int:
        lay = ops.OPT_W3S if Ls[0].w3s else 0
        return ops.layer_opts((ops.OPT_META_SCALABLE if all(L.opts & ops.OPT_META_SCALABLE for L in Ls) else 0) | lay)

    @torch.no_grad()
    def __call__(self, tok: Tensor, pos: Tensor, kv_len: int | None = None) -> Tensor:
        """tok [1, 1] int64, pos [1] int64 (its position; both on the device) -> logits [1, vocab] of the next token.
        kv_len (host integer > the position, default the whole cache): HF's attention function attends over the first kv_len cache positions only
        (masked beyond pos as before) — its cost follows the length it is given, so a caller that knows the position passes a bucket just above it"""
        inner = self.inner
        h = self.h
        # (the position itself is device memory — the step is graph-replayed —: the kernels that index the cache with it skip their
        #  writes beyond the cache's last slot, csrc/block.hip; callers that know the position on the host check it there, generation.py)
        if self.one_launch_front:   # embedding row, rotary table row and the causal mask in ONE launch (csrc/block.hip: copies and compares, the same bits as the ops below)
            ops.token_prologue(tok, pos, inner.embed_tokens.weight, h, self.cos_tab, self.sin_tab, self.cos_v, self.sin_v, None if self.attention == "hip" else self.mask.view(-1))
            cos, sin = self.cos_v, self.sin_v
        else:
            h.copy_(inner.embed_tokens(tok).view(1, self.H))
            if self.cos_tab is not None:
                cos, sin = self.cos_tab.index_select(0, pos).view(-1), self.sin_tab.index_select(0, pos).view(-1)
            else:
                cos, sin = inner.rotary_emb(h.view(1, 1, self.H), pos.view(1, 1))   # [1, 1, hd] each, the model's own rotary module
                cos, sin = cos.reshape(-1).contiguous(), sin.reshape(-1).contiguous()
            if self.attention != "hip":
                torch.where(self.ar <= pos, self.zero, self.ninf, out=self.mask.view(-1))   # the causal mask of one query at `pos` over the static cache
        kvl = self.L if kv_len is None else min(int(kv_len), self.L)
        mask = self.mask[..., :kvl]
        splits = ops.attn_splits(kvl) if self.attention == "hip" else 1   # (kernel attention: kv_len only picks how many workgroups share a head)
        if splits > 1 and splits not in self.attn_ws:
            self.attn_ws[splits] = ops.attn_workspace(self.device, self.n_heads, self.hd, splits)
        delta = None
        for b in self.blocks:
            at = b["attn"]
            K = self.H
            if self.folded and "qkv_rope" in b:   # RMSNorm in the prologue, rotary embedding + cache write in the epilogue: q|k|v lands rotated in qr / the caches
                ops.gemv_block(h, b["n1"].weight, b["n1"].variance_epsilon, b["qkv_rope"], K, b["qkv_gs"], b["qkv_nbits"], [b["qr"], b["kc"], b["vc"]],
                               ops.BLOCK_NORM | ops.BLOCK_ROPE, opts=b["qkv_rope_opts"], rope=(cos, sin, pos, self.hd, self.L))
            elif self.folded:   # RMSNorm in the launch's prologue: every workgroup normalises h itself while its first weights are in flight
                ops.gemv_block(h, b["n1"].weight, b["n1"].variance_epsilon, b["qkv"], K, b["qkv_gs"], b["qkv_nbits"], [b["q"], b["k"], b["v"]], ops.BLOCK_NORM, opts=b["qkv_opts"])
            else:
                ops.add_rmsnorm(h, delta, b["n1"].weight, b["n1"].variance_epsilon, out=self.xn)
                ops.gemv_grouped(self.xn, b["qkv"], K, b["qkv_gs"], b["qkv_nbits"], outs=[b["q"], b["k"], b["v"]], opts=b["qkv_opts"])
            if self.attention == "hip":   # rotary + cache write + attention: one launch
                att = ops.rope_attn_decode(b["q"], b["k"], b["v"], cos, sin, pos, b["kc"], b["vc"], self.att, at.scaling, splits=splits,
                                           workspace=self.attn_ws.get(splits))
            else:
                if not (self.folded and "qkv_rope" in b):
                    ops.rope_cache(b["q"], b["k"], b["v"], cos, sin, pos, b["kc"], b["vc"], b["qr"])
                att, _ = self.attn_fn(at, b["qr"], b["kc"][:, :kvl].unsqueeze(0), b["vc"][:, :kvl].unsqueeze(0), mask, dropout=0.0, scaling=at.scaling)
            o, d = b["o"], b["d"]
            if self.folded:
                # o: h += o(att) in the epilogue; gate|up: RMSNorm prologue + silu(gate) * up epilogue on the paired layer; down: h += down(a) in the epilogue
                ops.gemv_block(att.reshape(1, -1), None, 0.0, [(o.W_q, o.scale, o.zero, o.out_features)], o.in_features, o.group_size, o.nbits, [h], ops.BLOCK_RESID,
                               opts=ops.layer_opts(o.opts))
                ops.gemv_block(h, b["n2"].weight, b["n2"].variance_epsilon, b["gu_pair"], K, b["gu_gs"], b["gu_nbits"], [b["a"]], ops.BLOCK_NORM | ops.BLOCK_SILU, opts=b["gu_pair_opts"])
                ops.gemv_block(b["a"], None, 0.0, [(d.W_q, d.scale, d.zero, d.out_features)], d.in_features, d.group_size, d.nbits, [h], ops.BLOCK_RESID, opts=ops.layer_opts(d.opts))
                continue
            ops.gemv(att.reshape(1, -1), o.W_q, o.scale, o.zero, None, o.out_features, o.in_features, o.group_size, o.nbits, out=self.delta,
                     opts=ops.layer_opts(o.opts))
            ops.add_rmsnorm(h, self.delta, b["n2"].weight, b["n2"].variance_epsilon, out=self.xn)
            ops.gemv_grouped(self.xn, b["gu"], K, b["gu_gs"], b["gu_nbits"], outs=[b["g"], b["u"]], opts=b["gu_opts"])
            ops.silu_mul(b["g"], b["u"], out=b["a"])
            ops.gemv(b["a"], d.W_q, d.scale, d.zero, None, d.out_features, d.in_features, d.group_size, d.nbits, out=self.delta, opts=ops.layer_opts(d.opts))
            delta = self.delta
        ops.add_rmsnorm(h, delta, inner.norm.weight, inner.norm.variance_epsilon, out=self.xn)
        return self.model.lm_head(self.xn)

    def account_tokens(self, n: int) -> None:
        """StaticLayer.update's bookkeeping for the n tokens the fused steps appended (kept out of the captured step: one add per layer)"""
        for b in self.blocks:
            b["len"].add_(n)
