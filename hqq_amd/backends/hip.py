"""`prepare_for_inference(model, backend="hip")` target: an inference-only layer that owns exactly what the fused kernels
read (packed W_q in the reference layout, flat fp16 scale/zero, bias) and launches one HIP kernel per call.

Follows the patch-function contract of the reference's optimised backends (hqq/backends/torchao.py:299-339,
bitblas.py:174-202, gemlite.py:10-34): unwrap HQQLinearLoRA, return the layer unchanged (with a "Skipping" note) when the
configuration is not covered, otherwise build the new module, drop the old tensors and return it.
"""
from __future__ import annotations

import torch
from torch import Tensor, nn

from .. import ops
from ..core.quantize import HQQLinear, Quantizer


class HQQLinearHIP(nn.Module):
    """y = x @ dequantize(W_q)^T (+ bias) through hqq_hip_forward.  Keeps the packed tensor bit-identical to the
    HQQLinear it was built from, so `state_dict()`-level round trips back to the reference stay possible.

    3-bit layers (group_size 64, even out_features) are RE-LAID OUT here, once: the reference's container (ten unrelated row slabs per
    int32, bitpack.py:69-91) becomes the 3-bit stream layout of csrc/w3s.h — what torchao / marlin patching does for their kernels
    (hqq/backends/torchao.py:202-241, marlin.py:74-123) — and the layer then runs through the 4-bit container's kernels (`self.w3s`).
    `state_dict()` still carries the reference's container (restored bit for bit by hqq_hip_w3s_unpack) and `load_state_dict` takes it."""

    def __init__(self, hqq_layer: HQQLinear):
        super().__init__()
        m = hqq_layer.meta
        self.out_features, self.in_features = (int(v) for v in m["shape"])
        self.nbits = Quantizer._packing_bits[m["packing"]]
        self.group_size = int(m["group_size"])
        self.axis = 1
        self.compute_dtype = hqq_layer.compute_dtype
        self.device = hqq_layer.device
        self.name = getattr(hqq_layer, "name", None)
        W_q = hqq_layer.W_q.data
        if m["view_as_float"]:
            W_q = W_q.view(m["unpack_view_dtype"])
        self.w3s = bool(self.nbits == 3 and ops.w3s_covers(self.out_features, self.in_features, self.group_size) and W_q.is_cuda)
        if self.w3s:
            W_q = ops.w3s_pack(W_q.contiguous(), self.out_features, self.in_features)
        self.W_q = nn.Parameter(W_q.contiguous(), requires_grad=False)
        self.register_buffer("scale", m["scale"].reshape(-1).contiguous(), persistent=True)
        self.register_buffer("zero", m["zero"].reshape(-1).contiguous(), persistent=True)
        self.bias = None if hqq_layer.bias is None else hqq_layer.bias.to(device=W_q.device, dtype=self.compute_dtype)
        self.refresh_opts()

    def refresh_opts(self) -> None:
        """May the exact weight rebuild use its three-op form on this layer's (zero, scale)?  (include/hqq_hip.h, hqq_hip_meta_check.)
        Checked when the layer is built and again whenever a state dict is loaded into it; call it after editing `scale` / `zero` in place."""
        if self.w3s:
            self.opts = ops.OPT_W3S | (ops.OPT_META_SCALABLE if (self.compute_dtype == torch.float16 and self.scale.is_cuda and
                                                                  ops.w3s_meta_scalable(self.scale, self.zero, self.out_features, self.in_features)) else 0)
            return
        self.opts = ops.OPT_META_SCALABLE if (self.compute_dtype == torch.float16 and self.nbits in (8, 4, 2, 1) and self.scale.is_cuda and
                                               ops.meta_scalable(self.scale, self.zero, self.out_features, self.in_features, self.group_size, self.nbits)) else 0

    def _container_numel(self) -> int:
        return ((self.out_features * self.in_features // 64 + 9) // 10) * 64

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        super()._save_to_state_dict(destination, prefix, keep_vars)
        if self.w3s:   # the reference's bytes leave the layer, whatever layout it computes on
            destination[prefix + "W_q"] = ops.w3s_unpack(self.W_q.data, self.out_features, self.in_features)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        key = prefix + "W_q"
        if self.w3s and key in state_dict and state_dict[key].numel() == self._container_numel() and state_dict[key].dtype == torch.int32:
            state_dict = dict(state_dict)   # (shallow: only this entry is replaced, the caller's dict stays as it was)
            state_dict[key] = ops.w3s_pack(state_dict[key].to(self.W_q.device).contiguous(), self.out_features, self.in_features)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
        self.refresh_opts()   # scale / zero are persistent buffers: a loaded checkpoint may not satisfy what the old values did

    @staticmethod
    def check(hqq_layer: HQQLinear) -> bool:
        """configurations the fused kernels cover (the rest keep HQQLinear.forward_hip's dequantise + GEMM)"""
        m = getattr(hqq_layer, "meta", None)
        if m is None or not m.get("packing"):
            return False
        gs = m["group_size"]
        N, K = m["shape"]
        dt = hqq_layer.compute_dtype
        covered = (dt == torch.float16 and (m["packing"] in ("4bit_u8", "2bit_u8", "8bit_u8", "1bit_u8") or (m["packing"] == "3bit_32" and gs == 64))) or \
                  (dt == torch.bfloat16 and (m["packing"] in ("4bit_u8", "2bit_u8") or (m["packing"] == "3bit_32" and ops.w3s_covers(N, K, gs))))
        return (m["axis"] == 1 and covered
                and bool(gs) and gs % 16 == 0 and K % gs == 0 and (m["packing"] == "3bit_32" or N % ops.PER[Quantizer._packing_bits[m["packing"]]] == 0)
                and not m.get("quant_scale") and not m.get("quant_zero") and hqq_layer.W_q.is_cuda)

    def extra_repr(self) -> str:
        return f"in_features={self.in_features}, out_features={self.out_features}, bias={self.bias is not None}, nbits={self.nbits}, group_size={self.group_size}"

    def dequantize(self) -> Tensor:
        W_q = ops.w3s_unpack(self.W_q.data, self.out_features, self.in_features) if self.w3s else self.W_q
        return ops.dequantize(W_q, self.scale, self.zero, self.out_features, self.in_features, self.group_size, self.nbits, 1)

    @torch.compiler.disable   # (a tracing compiler has nothing to see in a ctypes call into libhqq_hip.so: HF's static-cache generate() auto-compiles the model's forward)
    def forward(self, x: Tensor) -> Tensor:
        if x.dtype != self.compute_dtype:
            x = x.to(self.compute_dtype)
        rows = x.numel() // x.shape[-1]
        # every number of rows through ops.forward: decode kernels, skinny GEMM, pipelined GEMM, and beyond them (long prompts, shapes the fused
        # kernels do not cover) the dequantise kernel + the in-tree dense GEMM; a 3-bit layer in the stream layout says so in its option bits
        return ops.forward(x, self.W_q, self.scale, self.zero, self.bias, self.out_features, self.in_features, self.group_size, self.nbits,
                           opts=ops.layer_opts(self.opts))


def patch_hqq_to_hip(layer, patch_params=None):
    hqq_layer = layer if isinstance(layer, HQQLinear) else getattr(layer, "linear_layer", None)
    if not isinstance(hqq_layer, HQQLinear):
        return layer
    if not HQQLinearHIP.check(hqq_layer):
        print("Skipping HIP conversion for ", getattr(hqq_layer, "name", None))
        return layer
    new = HQQLinearHIP(hqq_layer)
    del hqq_layer.W_q, hqq_layer.meta, hqq_layer.bias
    if hqq_layer is layer:
        return new
    layer.linear_layer = new   # HQQLinearLoRA-style wrapper keeps its adapters
    return layer


class _GroupedMember(nn.Module):
    """One of several HQQLinearHIP layers that are always called with the same activation tensor (q/k/v, gate/up).
    The first member called with a new `x` launches ONE grouped kernel for the whole set (hqq_hip_gemv_grouped) and parks the
    siblings' outputs; the siblings then return theirs without a launch.  Falls back to the member's own forward for batches the
    grouped decode kernel does not cover.  Transparent to the calling code (HF attention / MLP modules)."""

    def __init__(self, layer: HQQLinearHIP, group: "_GroupState", index: int):
        super().__init__()
        self.layer, self._group, self._index = layer, group, index
        self.in_features, self.out_features, self.bias = layer.in_features, layer.out_features, layer.bias
        self.compute_dtype, self.device = layer.compute_dtype, layer.device

    def dequantize(self) -> Tensor:
        return self.layer.dequantize()

    @torch.compiler.disable   # (a tracing compiler has nothing to see in a ctypes call into libhqq_hip.so: HF's static-cache generate() auto-compiles the model's forward)
    def forward(self, x: Tensor) -> Tensor:
        g = self._group
        # inference tensors (torch.inference_mode) carry no version counter: identity alone keys the parked outputs there
        ver = None if x.is_inference() else x._version
        if g.x is x and g.version == ver and g.outs[self._index] is not None:
            out, g.outs[self._index] = g.outs[self._index], None
            if all(o is None for o in g.outs):
                g.x = None   # every sibling served: do not keep the activation alive
            return out
        rows = x.numel() // x.shape[-1]
        if x.dtype != torch.float16:
            return self.layer(x)
        layers = [m.layer for m in g.members]
        specs = [(L.W_q, L.scale, L.zero, L.bias, L.out_features) for L in layers]
        if rows <= g.max_rows:
            outs = ops.gemv_grouped(x, specs, self.in_features, layers[0].group_size, layers[0].nbits, opts=ops.layer_opts(g.opts))
        elif g.gemm_rows and rows <= g.gemm_rows and ops.gemm_grouped_covers(x.dtype, [L.out_features for L in layers], rows, self.in_features, layers[0].group_size,
                                                                             layers[0].nbits, ops.layer_opts(g.opts)):
            # batched decode / speculative verification / short prompts: the group through ONE launch of the pipelined fused GEMM (round 6)
            outs = ops.gemm_grouped(x, specs, self.in_features, layers[0].group_size, layers[0].nbits, opts=ops.layer_opts(g.opts))
        else:
            return self.layer(x)
        g.x, g.version, g.outs = x, ver, list(outs)
        out, g.outs[self._index] = g.outs[self._index], None
        return out


class _GroupState:
    def __init__(self):
        self.members, self.x, self.version, self.outs, self.max_rows, self.opts = [], None, -1, [], 4, 0
        self.gemm_rows = 0   # > 0: batches up to this many rows take the grouped fused GEMM (hqq_hip_gemm_grouped); beyond, every layer its own route


def group_projections(parent: nn.Module, names) -> bool:
    """Fuse `parent.<name>` for name in names (all HQQLinearHIP, same in_features / nbits / group_size, fp16) into one grouped
    launch per distinct input.  Returns False (and changes nothing) when the layers cannot be grouped."""
    layers = [getattr(parent, n, None) for n in names]
    if not all(isinstance(L, HQQLinearHIP) for L in layers) or not 2 <= len(layers) <= ops.GEMV_MAX_GROUP:
        return False
    L0 = layers[0]
    if any((L.in_features, L.nbits, L.group_size, L.compute_dtype, L.w3s) != (L0.in_features, L0.nbits, L0.group_size, torch.float16, L0.w3s) for L in layers):
        return False
    if L0.nbits == 3 and any(L.group_size != 64 for L in layers):
        return False
    state = _GroupState()
    state.max_rows = 4 if (L0.nbits == 3 or L0.in_features % 64) else ops.GEMV_MAX_M   # (5..16 rows need K % 64 == 0)
    if L0.nbits == 3 and not L0.w3s:   # long K: fewer rows of x fit the kernel's LDS staging (70B down_proj: 2)
        while state.max_rows and not all(ops.decode_covers(torch.float16, state.max_rows, L.out_features, L.in_features, 64, 3) for L in layers):
            state.max_rows -= 1
        if not state.max_rows:
            return False
    if all(ops.skinny_covers(torch.float16, ops.SKINNY_MAX_M, L.out_features, L.in_features, L.group_size, L.nbits, L.w3s) for L in layers):
        state.max_rows = ops.SKINNY_MAX_M   # decode with a batch: still one weight-streaming launch for the group
    state.opts = (ops.OPT_META_SCALABLE if all(L.opts & ops.OPT_META_SCALABLE for L in layers) else 0) | (ops.OPT_W3S if L0.w3s else 0)
    # up to the row count where ops.forward itself leaves the fused GEMM for dequantise + dense GEMM (hqq_hip_forward_prefers_fused: 2560)
    if ops.gemm_grouped_covers(torch.float16, [L.out_features for L in layers], 128, L0.in_features, L0.group_size, L0.nbits, state.opts):
        state.gemm_rows = ops.FUSED_GEMM_MAX_M
    for i, (n, L) in enumerate(zip(names, layers)):
        m = _GroupedMember(L, state, i)
        state.members.append(m)
        setattr(parent, n, m)
    return True


def group_llama_projections(model: nn.Module) -> int:
    """q|k|v and gate|up of every Llama-style decoder block -> grouped launches (what bench.py measures).  Returns the number
    of groups formed.  Call after prepare_for_inference(model, backend="hip")."""
    n = 0
    for mod in model.modules():
        if all(hasattr(mod, a) for a in ("q_proj", "k_proj", "v_proj")):
            n += int(group_projections(mod, ("q_proj", "k_proj", "v_proj")))
        if all(hasattr(mod, a) for a in ("gate_proj", "up_proj")):
            n += int(group_projections(mod, ("gate_proj", "up_proj")))
    return n
