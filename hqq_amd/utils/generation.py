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
        self.cache = None      # HF StaticCache, kept between generate() calls (reset in place)
        self._state = None     # (tok, next_tok, pos): the device tensors the captured graphs read and write

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

    def _fingerprint(self):
        """what the kept state was built from: identity, storage and version of every quantised layer's packed weights and scale, and the forward HQQLinear is
        bound to (set_backend rebinds it class-wide).  A re-quantised / re-loaded / re-patched model gives another tuple (round-5 advisor: the kept step and
        graphs silently decoded with stale copies).  In-place edits through .data bypass the version counters: call reset() after those."""
        from ..core.quantize import HQQLinear
        fp = [getattr(HQQLinear, "backend", None)]
        for m in self.model.modules():
            W, meta = getattr(m, "W_q", None), getattr(m, "meta", None)
            if isinstance(W, Tensor):
                sc = meta.get("scale") if isinstance(meta, dict) else getattr(m, "scale", None)
                ver = lambda t: None if (t is None or t.is_inference()) else t._version   # noqa: E731
                fp.append((id(m), W.data_ptr(), ver(W), None if sc is None else sc.data_ptr(), ver(sc)))
        return tuple(fp)

    def reset(self) -> None:
        """drop what generate() keeps between calls (the static cache, the fused step with its re-laid-out layer copies, the captured graphs).  generate() calls it
        by itself when the model's quantised layers are no longer the ones the state was built from (_fingerprint)"""
        self.cache = None
        self.step = None
        self.graph = None
        self.graphs = {}
        self._state = None

    @torch.no_grad()
    def generate(self, input_ids: Tensor, max_new_tokens: int, use_graph: bool = True, eos_token_id: int | None = None, check_every: int = 16) -> Tensor:
        """continuation of a single sequence [1, T] — greedy, or sampled when the decoder was built with do_sample=True; returns [1, T + n], n = max_new_tokens, or fewer
        when eos_token_id is given and was produced (the sequence then ends with it; the host looks at the tokens every `check_every` steps, never per token).
        The static cache, the fused step (its paired / rotary-paired layer copies) and the captured graphs are KEPT between calls: the next prompt resets the cache in
        place (StaticCache.reset keeps the tensors) and replays the same graphs — hqq/utils/generation_hf.py:190-207, :313-327 keep theirs the same way; reset() drops them."""
        assert input_ids.shape[0] == 1, "one sequence (the decode-shaped bs=1 path)"
        T = input_ids.shape[1]
        assert T + max_new_tokens <= self.max_cache_len
        ids = input_ids.to(self.device)
        fp = self._fingerprint()
        if getattr(self, "_fp", None) != fp:   # other weights / layers / backend than the kept step and graphs were built from
            self.reset()
            self._fp = fp
        kept = getattr(self, "cache", None) is not None and getattr(self, "_state", None) is not None
        if kept:
            self.cache.reset()
        else:
            self.cache = self._StaticCache(config=self.model.config, max_cache_len=self.max_cache_len)
        out = self.model(ids, past_key_values=self.cache, cache_position=torch.arange(T, device=self.device), use_cache=True)   # prefill
        first = self._pick(out.logits[:, -1])
        if kept:
            self.tok, self.next_tok, self.pos = self._state   # the tensors the captured graphs read and write
            self.tok.copy_(first)
            self.pos.fill_(T)
        else:
            self.tok = first
            self.next_tok = torch.empty_like(self.tok)
            self.pos = torch.tensor([T], device=self.device)
            self._state = (self.tok, self.next_tok, self.pos)
            self.step = None
            if self.fused:
                try:
                    self.step = self._fused_mod.FusedLlamaStep(self.model, self.cache, self.max_cache_len, attention=self.attention, glue=self.glue)
                except ValueError:   # a cache layout / attention configuration the fused step does not restate: the model's own forward serves
                    self.step = None
            self.graph = None
            self.graphs = {}
        toks = [self.tok.clone()]
        done = 0        # tokens the host has looked at
        n = max_new_tokens
        for i in range(max_new_tokens - 1):
            self._advance(T + i, use_graph and (kept or i >= 1))   # (a fresh decoder's step 0 runs eagerly: lazy initialisation inside the model)
            toks.append(self.tok.clone())
            if eos_token_id is not None and (len(toks) - done >= check_every or i == max_new_tokens - 2):
                seen = torch.cat(toks[done:], dim=1)[0].tolist()   # one host read per check_every tokens
                if eos_token_id in seen:
                    n = done + seen.index(eos_token_id) + 1
                    break
                done = len(toks)
        if eos_token_id is not None and n == max_new_tokens and len(toks) == 1 and int(toks[0]) == eos_token_id:
            n = 1
        if self.step is not None:
            self.step.account_tokens(len(toks) - 1)
        return torch.cat([ids] + toks[:n], dim=1)

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



# prompts of different lengths for HFGenerator.warmup(): what matters is that the first cache buckets get their graphs captured, not what is asked
WARMUP_PROMPTS = ["Hello.", "Name three prime numbers and say why each one is prime.",
                  "Explain in two paragraphs how a key-value cache speeds up autoregressive decoding, and what it costs in memory."]


class HFGenerator:
    """The reference's generation front end (hqq/utils/generation_hf.py:117-540: HFGenerator(model, tokenizer, ...).generate(prompt) -> {"output_text", "output_tokens",
    "input_tokens"}) over GraphedGreedyDecoder: same constructor arguments, same methods a caller uses (warmup, generate, tokenize_prompt), same defaults
    (cache_size = the next power of two above max_new_tokens, greedy unless do_sample, temperature 0.6 / top_k 5, stop at the tokenizer's EOS).
    What `compile` means here: the reference compiles the decode step with torch.compile ("partial" / "full") and can wrap it in a CUDA graph; this loop has no tracing
    compiler — "partial" / "full" both select the captured-hipGraph step (one graph per token, kept across prompts), None the same step launched eagerly.
    `compile_options` / `patch_accelerate` are accepted and unused.  The decode step itself is hqq_amd.utils.llama_fused (HIP kernels) for Llama-shaped models whose
    linears went through prepare_for_inference(backend="hip"), the model's own forward otherwise.
    Differences a caller can see: EOS is looked for every 16 tokens on the host instead of after every token (no per-token synchronisation; the text returned is cut at
    the EOS all the same); "output_tokens" holds every generated token before the EOS (the reference's slice drops the last one it generated, generation_hf.py:493);
    a prompt that leaves less than max_new_tokens of cache generates what fits."""

    def __init__(self, model, tokenizer, max_new_tokens: int = 1000, cache_size: int | None = None, do_sample: bool = False, temperature: float = 0.6, top_k: int = 5,
                 compile: str | None = None, compile_options: dict | None = None, patch_accelerate: bool = True):
        if compile not in (None, "partial", "full"):
            raise ValueError("compile: None, 'partial' or 'full'")
        self.model, self.tokenizer = model, tokenizer
        self.device = next(p.device for p in model.parameters() if p.device.type == "cuda")
        self.do_sample = bool(do_sample)
        self.temperature = temperature if self.do_sample else None
        self.top_k = top_k if self.do_sample else None
        self.max_new_tokens = int(max_new_tokens)
        self.cache_size = self.next_multiple(self.max_new_tokens) if cache_size is None else int(cache_size)
        self.max_new_tokens = min(self.max_new_tokens, self.cache_size)
        self.is_compiled = compile is not None
        self.use_graph = compile is not None
        self.compile_options = compile_options
        self.decoder = GraphedGreedyDecoder(model, max_cache_len=self.cache_size, do_sample=self.do_sample, temperature=temperature, top_k=top_k)
        self.init()

    @staticmethod
    def next_multiple(val: int) -> int:
        """the next power of two above val, from 32 (generation_hf.py:233-236)"""
        n = 32
        while n <= val:
            n *= 2
        return n

    def init(self) -> None:
        """inference-mode settings of tokenizer and model (generation_hf.py:238-247)"""
        tk = self.tokenizer
        for name in ("add_bos_token", "add_eos_token"):
            if hasattr(tk, name):
                setattr(tk, name, False)
        if getattr(tk, "pad_token", None) in (None, "") and hasattr(tk, "add_special_tokens"):
            tk.add_special_tokens({"pad_token": "<<[PAD]>>"})
        if hasattr(tk, "padding_side"):
            tk.padding_side = "right"
        self.model.eval()
        # (the reference also sets model.generation_config.cache_implementation = "static" here: its loop shares the model's own generate() settings.  This loop owns its
        #  StaticCache; the setting would only make every LATER model.generate() call of the caller compile the model — generate_() asks for the static cache itself)
        self.model.config.use_cache = True

    def reset(self) -> None:
        """drop the decoder's kept static cache, fused step and captured graphs (GraphedGreedyDecoder.reset; generate() also does it by itself when the model's
        quantised layers changed — call this after an in-place edit of weights that bypasses torch's version counters)"""
        self.decoder.reset()

    def warmup(self, max_samples: int = -1):
        """a few prompts through the loop: the fused step is built and the graphs of the first cache buckets captured before a caller's clock starts"""
        for prompt in WARMUP_PROMPTS[:max_samples if max_samples > 0 else len(WARMUP_PROMPTS)]:
            self.generate(prompt, verbose=False, print_tokens=False)
        return self

    def tokenize_prompt(self, prompt: str, use_chat_template: bool = True):
        if use_chat_template:
            prompt = self.tokenizer.apply_chat_template([{"role": "user", "content": prompt}], tokenize=False, add_generation_prompt=True)
        return self.tokenizer([prompt], return_tensors="pt").to(device=self.device)

    @torch.no_grad()   # (not inference_mode: graph capture updates generator state tensors in place, which inference tensors refuse outside that mode)
    def generate(self, prompt: str, use_chat_template: bool = True, verbose: bool = True, print_tokens: bool = False) -> dict:
        inputs = self.tokenize_prompt(prompt, use_chat_template=use_chat_template)
        ids = inputs["input_ids"].to(torch.int64)
        T = ids.shape[1]
        n = min(self.max_new_tokens, self.cache_size - T)
        if n < 1:
            raise ValueError(f"hqq_amd: the prompt ({T} tokens) leaves no room in a cache of {self.cache_size}")
        eos = getattr(self.tokenizer, "eos_token_id", None)
        out = self.decoder.generate(ids, n, use_graph=self.use_graph, eos_token_id=eos)
        new = out[0, T:]
        if eos is not None and new.numel() and int(new[-1]) == eos:
            new = new[:-1]
        output_tokens = new.cpu()
        output_text = self.tokenizer.decode(output_tokens)
        if print_tokens:
            print(output_text, flush=True)
        return {"output_text": output_text, "output_tokens": output_tokens, "input_tokens": ids[0].cpu()}

    def generate_(self, prompt: str, use_chat_template: bool = True, verbose: bool = False, print_tokens: bool = False) -> dict:
        """HF's own generate with a static cache (generation_hf.py:515-527): the loop this class replaces, for comparison"""
        gen_out = self.model.generate(**self.tokenize_prompt(prompt, use_chat_template=use_chat_template), do_sample=self.do_sample, cache_implementation="static",
                                      max_new_tokens=self.max_new_tokens, pad_token_id=getattr(self.tokenizer, "pad_token_id", None),
                                      **({"temperature": self.temperature, "top_k": self.top_k} if self.do_sample else {}))[0]
        return {"output_text": self.tokenizer.decode(gen_out), "output_tokens": gen_out}
