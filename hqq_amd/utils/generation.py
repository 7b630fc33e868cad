"""Greedy decode loop with a static KV cache and ONE captured hipGraph per decode step (the reference's
`HFGenerator`, hqq/utils/generation_hf.py:117-540, does the same with torch.compile + CUDA graphs).

At batch 1 every fused dequant-GEMV is a few microseconds, i.e. comparable to an eager launch from Python; replaying the whole
step — all decoder layers, attention, sampling argmax — as one graph removes the host from the loop.  The HIP kernels are
capturable by construction (no allocation, no host sync, stream passed in)."""
from __future__ import annotations

import torch
from torch import Tensor

from .. import ops


class GraphedGreedyDecoder:
    """fused=True (default): a Llama-shaped model whose decoder linears are HQQLinearHIP layers decodes through hqq_amd.utils.llama_fused —
    RMSNorm (+ the residual adds), rotary + KV-cache write and SiLU * up as one HIP kernel each around the grouped GEMVs, HF's own attention
    function on HF's cache: the same tokens in a third of the launches; since round 5 (glue="auto") the RMSNorms, the residual adds and SiLU * up
    ride inside the GEMV launches themselves: 5 launches + attention per decoder block.  Any other model, or fused=False: the model's own forward."""

    def __init__(self, model, max_cache_len: int = 512, fused: bool = True, attention: str = "sdpa", bucket_cache: bool = True, glue: str = "auto",
                 do_sample: bool = False, temperature: float = 0.6, top_k: int | None = 5):
        from transformers import StaticCache
        from . import llama_fused
        self.model = model.eval()
        self.fused = bool(fused) and llama_fused.supports(model)
        self._fused_mod = llama_fused
        self.attention = attention   # "sdpa": HF's attention function (token-identical to model(...)); "hip": the decode-attention kernel (faster, within rounding)
        self.glue = glue             # "auto" / "folded": RMSNorm, residual adds and SiLU * up inside the GEMV launches (csrc/gemv_block.hip); "kernels": round 4's separate glue kernels
        # sampling (the reference's HFGenerator(do_sample=True, temperature=0.6, top_k=5), hqq/utils/generation_hf.py:250-311): applied to the logits ON THE
        # DEVICE inside the captured step — no host round trip, the generator's Philox offset advances per replay —; greedy argmax otherwise
        self.do_sample, self.temperature, self.top_k = bool(do_sample), float(temperature), (None if top_k is None else int(top_k))
        self.bucket_cache = bucket_cache   # attention="sdpa": attend over a bucket of the static cache just above the position (False: all of it)
        self.step = None
        self.device = next(p.device for p in model.parameters() if p.device.type == "cuda")
        self.max_cache_len = max_cache_len
        self._StaticCache = StaticCache
        self.graph = None      # the graph of the last step taken
        self.graphs = {}       # attended cache length -> captured step

    def _kv_len(self, p: int) -> int:
        """how much of the static cache a step at position p attends over.  HF's attention function costs what it is given (the whole masked cache:
        304 tok/s at 1024 positions, 119 at 4096, against 500 at 256), so the fused step hands it a bucket just above the position (64, then
        multiples of 128 to 1024, then multiples of 512) and keeps one captured graph per bucket; the kernel attention reads pos + 1 keys by itself and uses the bucket only to decide
        how many workgroups share a head's keys (one up to 1024 keys)"""
        if self.step is None or not self.bucket_cache:
            return self.max_cache_len
        n = p + 1
        if self.attention == "hip":   # (the kernel attention only changes its launch shape beyond 1024 visible keys: powers of two from there)
            b = 1024
            while b < n:
                b *= 2
        elif n <= 64:                 # SDPA over 64 / 128 / 256 / 512 keys: 6.4 / 9.0 / 14.4 / 24.8 us per block
            b = 64
        elif n <= 1024:
            b = -(-n // 128) * 128
        else:
            b = -(-n // 512) * 512
        return min(b, self.max_cache_len)

    def _pick(self, logits: Tensor) -> Tensor:
        """logits [1, vocab] -> the next token [1, 1].  Greedy: argmax.  do_sample: temperature, then the top_k cut, then one draw from the softmax by the
        exponential-race form of a categorical draw (argmax of p / e, e ~ Exp(1)): elementwise kernels + two reductions, nothing leaves the device"""
        if not self.do_sample:
            return logits.argmax(-1, keepdim=True)
        z = logits.float() / max(self.temperature, 1e-5)
        if self.top_k is not None:
            kth = torch.topk(z, min(self.top_k, z.shape[-1])).values[..., -1:]
            z = torch.where(z < kth, torch.full_like(z, float("-inf")), z)
        p = torch.softmax(z, dim=-1)
        return (p / torch.empty_like(p).exponential_(1.0)).argmax(-1, keepdim=True)

    @torch.no_grad()
    def _decode_once(self, kv_len=None):
        """one whole transition: logits at self.pos -> next_tok, tok = next_tok, pos += 1 (all on the device, so the captured graph carries the loop state forward by itself)"""
        if self.step is not None:
            logits = self.step(self.tok, self.pos, kv_len)
            if not self.do_sample and self.glue != "kernels" and logits.dtype in (torch.float16, torch.bfloat16) and logits.is_contiguous():
                ops.argmax_advance(logits, self.next_tok, self.tok, self.pos)   # argmax + hand-over + position increment: one launch (csrc/block.hip)
                return
            self.next_tok.copy_(self._pick(logits))
        else:
            out = self.model(self.tok, past_key_values=self.cache, cache_position=self.pos, use_cache=True)
            self.next_tok.copy_(self._pick(out.logits[:, -1]))
        self.tok.copy_(self.next_tok)
        self.pos += 1

    @torch.no_grad()
    def generate(self, input_ids: Tensor, max_new_tokens: int, use_graph: bool = True) -> Tensor:
        """continuation of a single sequence [1, T] — greedy, or sampled when the decoder was built with do_sample=True; returns [1, T + max_new_tokens]"""
        assert input_ids.shape[0] == 1, "one sequence (the decode-shaped bs=1 path)"
        T = input_ids.shape[1]
        assert T + max_new_tokens <= self.max_cache_len
        ids = input_ids.to(self.device)
        self.cache = self._StaticCache(config=self.model.config, max_cache_len=self.max_cache_len)
        out = self.model(ids, past_key_values=self.cache, cache_position=torch.arange(T, device=self.device), use_cache=True)   # prefill
        self.tok = self._pick(out.logits[:, -1])
        self.next_tok = torch.empty_like(self.tok)
        self.pos = torch.tensor([T], device=self.device)
        self.step = None
        if self.fused:
            try:
                self.step = self._fused_mod.FusedLlamaStep(self.model, self.cache, self.max_cache_len, attention=self.attention, glue=self.glue)
            except ValueError:   # a cache layout / attention configuration the fused step does not restate: the model's own forward serves
                self.step = None
        toks = [self.tok.clone()]
        self.graph = None
        self.graphs = {}
        for i in range(max_new_tokens - 1):
            self._advance(T + i, use_graph and i >= 1)   # step 0 runs eagerly (lazy initialisation inside the model)
            toks.append(self.tok.clone())
        if self.step is not None:
            self.step.account_tokens(max_new_tokens - 1)
        return torch.cat([ids] + toks, dim=1)

    @torch.no_grad()
    def _advance(self, p: int, use_graph: bool) -> None:
        """one decode step at position p (the host's copy of self.pos): replay the graph of p's cache bucket, capturing it first if need be"""
        kv = self._kv_len(p)
        g = self.graphs.get(kv) if use_graph else None
        if use_graph and g is None:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                snap = (self.tok.clone(), self.pos.clone())
                self._decode_once(kv)                    # warm-up on the side stream (writes cache slot pos, re-written below)
                self.tok.copy_(snap[0]); self.pos.copy_(snap[1])
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._decode_once(kv)
            self.graphs[kv] = g                          # the capture itself does not execute: replay for this step
        if g is not None:
            g.replay()
            self.graph = g
        else:
            self._decode_once(kv)

    @torch.no_grad()
    def benchmark(self, input_ids: Tensor, new_tokens: int = 64, warmup: int = 8) -> dict:
        """end-to-end decode rate: prefill `input_ids`, capture the decode step, then time `new_tokens` replays of it (HIP events on
        the current stream; the argmax feeds the next step on the device, the host only replays).  Returns tok/s and ms per token."""
        assert input_ids.shape[0] == 1
        T = input_ids.shape[1]
        assert T + warmup + new_tokens + 4 <= self.max_cache_len
        self.generate(input_ids, 3, use_graph=True)          # prefill + eager step + captured step (leaves self.graphs, self.tok, self.pos)
        assert self.graph is not None
        p = T + 2                                             # the host's copy of self.pos
        for _ in range(warmup):
            self._advance(p, True)
            p += 1
        for q in range(p, p + new_tokens):                    # buckets the timed steps will enter: captured before the clock starts
            if self._kv_len(q) not in self.graphs:
                snap = (self.tok.clone(), self.pos.clone())
                self._advance(q, True)
                self.tok.copy_(snap[0]); self.pos.copy_(snap[1])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(new_tokens):
            self._advance(p, True)
            p += 1
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / new_tokens
        return {"ms_per_token": ms, "tok_s": 1e3 / ms, "new_tokens": new_tokens, "prompt_tokens": T}

