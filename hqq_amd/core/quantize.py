"""HQQLinear / HQQBackend / Quantizer / BaseQuantizeConfig — the reference's layer surface (hqq/core/quantize.py) on the
MI355X kernels of libhqq_hip.so.

What is kept from the reference: names, signatures, attribute names, the `meta` dict and the flat state-dict schema
(quantize.py:617-787), the class-wide `set_backend` mechanism (an enum whose value is the name of a forward method,
:269-285 / :498-512) and the no-op `.to()/.half()/...` family (:585-613).  What is new: every tensor operation is a call
through the C ABI (hqq_amd.ops): the solver + packing (`hqq_hip_quantize`), dequantisation (`hqq_hip_dequantize`) and the
fused forward (`hqq_hip_forward`).  The extra enum member `HQQBackend.HIP = "forward_hip"` is the default backend.

There is no CPU compute path in this package: constructing / loading a layer needs `device="cuda"` (a ROCm GPU).
State-dict encode/decode and config handling are pure host logic and work anywhere.
"""
from __future__ import annotations

import copy
from enum import Enum
from typing import Union

import torch
from torch import Tensor, float16, int32, nn, uint8

from .. import ops
from .bitpack import BitPack
from .utils import decode_safetensor_type, encode_safetensor_type, is_divisible

# types of the non-tensor entries of `meta` / the state dict (quantize.py:15-32)
_META_TYPE = {
    "scale": torch.Tensor, "zero": torch.Tensor, "zero_scale": torch.Tensor, "compute_dtype": torch.dtype,
    "quant_zero": bool, "quant_scale": bool, "view_as_float": bool, "unpack_view_dtype": torch.dtype,
    "packing": str, "axis": int, "group_size": int, "nbits": int, "shape": torch.Size,
    "channel_wise": bool, "optimize": bool, "round_zero": bool,
}


from .optimize import optimize_weights_proximal


class Quantizer:
    """hqq/core/quantize.py:36-253.  quantize() runs the half-quadratic solver + bit-packing in one HIP call."""
    SUPPORTED_BITS = [8, 6, 5, 4, 3, 2, 1.58, 1]
    bit_to_packing = {8: "8bit_u8", 6: "8bit_u8", 5: "8bit_u8", 4: "4bit_u8", 3: "3bit_32", 2: "2bit_u8", 1.58: "2bit_u8", 1: "1bit_u8"}
    pack = {"8bit_u8": BitPack.pack_8bit_u8, "4bit_u8": BitPack.pack_4bit_u8, "3bit_32": BitPack.pack_3bit_32,
            "2bit_u8": BitPack.pack_2bit_u8, "1bit_u8": BitPack.pack_1bit_u8}
    unpack = {"8bit_u8": BitPack.unpack_8bit_u8, "4bit_u8": BitPack.unpack_4bit_u8, "3bit_32": BitPack.unpack_3bit_32,
              "2bit_u8": BitPack.unpack_2bit_u8, "1bit_u8": BitPack.unpack_1bit_u8}
    unpack_view_dtype = {"8bit_u8": uint8, "4bit_u8": uint8, "3bit_32": int32, "2bit_u8": uint8, "1bit_u8": uint8}
    _packing_bits = {"8bit_u8": 8, "4bit_u8": 4, "3bit_32": 3, "2bit_u8": 2, "1bit_u8": 1}
    optimize_weights = optimize_weights_proximal   # (quantize.py:72; quantize() below runs the same solver fused with the initialisation and the packing)

    @classmethod
    def quantize(cls, tensor: Tensor, nbits: float = 4, channel_wise: bool = True, group_size: int = 64, optimize: bool = True,
                 round_zero: bool = False, axis: int = 0, bitpack: bool = True, compute_dtype: Union[torch.dtype, None] = None,
                 view_as_float: bool = False, device: str = "cuda") -> tuple:
        assert nbits in Quantizer.SUPPORTED_BITS, "nbits=" + str(nbits) + " not supported."
        assert axis in [0, 1], "axis should be either 0 or 1"
        if group_size is not None:
            assert is_divisible(tensor.numel(), group_size), (
                "group_size should be divisble by the total tensor dimensions. shape: " + str(tensor.shape) + ", group_size: " + str(group_size))
        shape = tensor.shape
        if not channel_wise:   # one scale / zero for the whole tensor, no solver, levels in the tensor's own shape (quantize.py:114-116)
            W = tensor.to(device)
            W_q, scale, zero = ops.quantize_tensorwise(W, nbits=nbits, round_zero=round_zero)
            meta = {"nbits": nbits, "group_size": group_size, "shape": shape, "scale": scale, "zero": zero, "axis": axis,
                    "packing": Quantizer.bit_to_packing[nbits]}
            if not bitpack:
                W_q = ops.unpack(Quantizer._packing_bits[meta["packing"]], W_q, dtype=tensor.dtype if tensor.dtype in (torch.float16, torch.bfloat16, torch.float32) else torch.float32)[:shape[0]]
                meta["packing"] = None
                meta["unpack_view_dtype"], meta["view_as_float"] = None, False
                return W_q, meta
            meta["unpack_view_dtype"] = Quantizer.unpack_view_dtype[meta["packing"]]
            meta["view_as_float"] = view_as_float
            if view_as_float:
                W_q = W_q.view(torch.float32 if compute_dtype is None else compute_dtype)
            return W_q, meta
        # group_size None: one group per row (axis 1) / per column (axis 0) — HQQLinear.initialize resolves it the same way (quantize.py:434-439)
        gs = (tensor.shape[-1] if axis == 1 else tensor.shape[0]) if group_size is None else group_size
        W = tensor.to(device)
        W_q, scale, zero = ops.quantize(W, nbits=nbits, group_size=gs, round_zero=round_zero, optimize=optimize, axis=axis)
        meta = {"nbits": nbits, "group_size": group_size, "shape": shape, "scale": scale, "zero": zero, "axis": axis,
                "packing": Quantizer.bit_to_packing[nbits]}
        if not bitpack:   # the levels themselves, in the input dtype (quantize.py:169): unpack what the fused solver packed
            rows = (W.numel() // gs) if axis == 1 else gs
            W_q = ops.unpack(Quantizer._packing_bits[meta["packing"]], W_q, dtype=tensor.dtype if tensor.dtype in (torch.float16, torch.bfloat16, torch.float32) else torch.float32)[:rows]
            meta["packing"] = None
            meta["unpack_view_dtype"], meta["view_as_float"] = None, False
            return W_q, meta
        meta["unpack_view_dtype"] = Quantizer.unpack_view_dtype[meta["packing"]]
        meta["view_as_float"] = view_as_float
        if view_as_float:   # store the packed bytes reinterpreted as the compute dtype (quantize.py:170-173)
            W_q = W_q.view(torch.float32 if compute_dtype is None else compute_dtype)
        return W_q, meta

    @classmethod
    def dequantize(cls, W_q: Tensor, meta: dict) -> Tensor:
        """bit-unpack -> (W_q - zero) * scale -> reshape, two roundings in the compute dtype (quantize.py:183-199)"""
        if not meta["packing"]:   # bitpack=False: W_q holds the levels themselves (quantize.py:196) — two GPU elementwise ops, as the reference does it
            cd = meta["compute_dtype"] if ("compute_dtype" in meta) else float16
            return ((W_q.to(cd) - meta["zero"]) * meta["scale"]).reshape(meta["shape"])
        if meta["view_as_float"]:
            W_q = W_q.view(meta["unpack_view_dtype"])
        N, K = meta["shape"]
        gs_eff = meta["group_size"] if meta["group_size"] else (K if meta["axis"] == 1 else N)
        if meta["scale"].numel() == 1 and meta["zero"].numel() == 1 and (N * K) // gs_eff > 1:
            # channel_wise=False: one scale / zero for the tensor, levels packed in the tensor's own shape — every row is a group with the
            # same constants.  Recognised by the meta itself (one constant pair for more than one group), whatever group_size the caller
            # left in it: Quantizer.quantize(channel_wise=False) stores the argument's default, 64 (the reference broadcasts the 0-d pair)
            return ops.dequantize(W_q, meta["scale"].reshape(1).expand(N).contiguous(), meta["zero"].reshape(1).expand(N).contiguous(),
                                  N, K, K, Quantizer._packing_bits[meta["packing"]], 1)
        gs = meta["group_size"] if meta["group_size"] else (K if meta["axis"] == 1 else N)
        return ops.dequantize(W_q, meta["scale"].reshape(-1), meta["zero"].reshape(-1), N, K, gs,
                              Quantizer._packing_bits[meta["packing"]], meta["axis"])

    @classmethod
    def to_inplace(cls, W_q, meta: dict, device) -> tuple:
        compute_dtype = meta["compute_dtype"] if ("compute_dtype" in meta) else float16
        if W_q is not None:
            W_q = W_q.to(device).contiguous()
        for key, val in meta.items():
            if isinstance(val, torch.Tensor):
                meta[key] = (val.to(compute_dtype) if torch.is_floating_point(val) else val).to(device).contiguous()
        return W_q, meta

    @classmethod
    def to_ooplace(cls, W_q, meta: dict, device) -> tuple:
        W_q_c, meta_c = Quantizer.to_inplace(None if W_q is None else W_q.clone(), dict(meta), device)
        return W_q_c, meta_c

    @classmethod
    def cuda(cls, W_q, meta: dict, device) -> tuple:
        return Quantizer.to_inplace(W_q, meta, device=device)

    @classmethod
    def cpu(cls, W_q, meta: dict) -> tuple:
        return Quantizer.to_ooplace(W_q, meta, device="cpu")


class HQQBackend(Enum):
    """Value = name of the HQQLinear forward method (quantize.py:269-285).  Reference members kept verbatim; HIP is new."""
    PYTORCH = "forward_pytorch_backprop"
    PYTORCH_COMPILE = "forward_pytorch_backprop_compile"
    ATEN = "forward_aten_backprop"
    PYTORCH_BACKPROP = "forward_pytorch_backprop"
    PYTORCH_BACKPROP_COMPILE = "forward_pytorch_backprop_compile"
    ATEN_BACKPROP = "forward_aten_backprop"
    PYTORCH_FORWARD = "forward_pytorch"
    PYTORCH_FORWARD_COMPILE = "forward_pytorch_compile"
    ATEN_FORWARD = "forward_aten"
    ATEN_FORWARD_INT8 = "forward_aten_int8"
    # fused unpack -> dequantize -> GEMV / MFMA GEMM in one HIP launch (inference; backward wrt x re-dequantises)
    HIP = "forward_hip"


class _MatmulNoCache(torch.autograd.Function):
    """y = matmul(x) (+ bias); backward wrt x calls matmul(grad, transpose=False) — the re-dequantising scheme of
    HQQMatmulNoCacheMul (quantize.py:322-352).  Weight gradients are undefined for frozen quantised weights."""
    @staticmethod
    def forward(x, matmul, bias):
        out = matmul(x, transpose=True)
        if bias is not None:
            out += bias
        return out

    @staticmethod
    def setup_context(ctx, inputs, outputs):
        x, matmul, bias = inputs
        ctx.save_for_backward(x, bias)
        ctx.matmul = matmul

    @staticmethod
    def backward(ctx, grad_output):
        x, bias = ctx.saved_tensors
        grad_input = ctx.matmul(grad_output, transpose=False) if ctx.needs_input_grad[0] else None
        grad_bias = grad_output.reshape(-1, grad_output.shape[-1]).sum(0) if (bias is not None and ctx.needs_input_grad[2]) else None
        return grad_input, None, grad_bias


# the reference's public names for the autograd functions of the path (quantize.py:289-385; used by HQQLinear itself and by
# code written against it, e.g. the PEFT wrappers).  Compute stays in the HIP kernels behind `dequantize` / `matmul`.
HQQMatmulNoCacheMul = _MatmulNoCache


class HQQMatmulNoCacheDeq(torch.autograd.Function):
    """y = x @ dequantize().t() (+ bias), the weight re-dequantised in backward instead of cached (quantize.py:289-319)"""
    @staticmethod
    def forward(x, dequantize, bias):
        out = torch.matmul(x, dequantize().t())
        if bias is not None:
            out += bias
        return out

    @staticmethod
    def setup_context(ctx, inputs, outputs):
        x, dequantize, bias = inputs
        ctx.save_for_backward(x, bias)
        ctx.dequantize = dequantize

    @staticmethod
    def backward(ctx, grad_output):
        x, bias = ctx.saved_tensors
        grad_input = torch.matmul(grad_output, ctx.dequantize()) if ctx.needs_input_grad[0] else None
        grad_bias = grad_output.reshape(-1, grad_output.shape[-1]).sum(0) if (bias is not None and ctx.needs_input_grad[2]) else None
        return grad_input, None, grad_bias


class HQQMatmulCachedDeq(torch.autograd.Function):
    """same product with the dequantised weight kept for backward: faster, one fp16 copy of W more (quantize.py:356-385)"""
    @staticmethod
    def forward(ctx, x, hqq_layer, bias):
        weight_tmp = hqq_layer.dequantize()
        out = torch.matmul(x, weight_tmp.t())
        if bias is not None:
            out += bias
        ctx.save_for_backward(x, bias, weight_tmp)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        x, bias, weight_tmp = ctx.saved_tensors
        grad_input = torch.matmul(grad_output, weight_tmp) if ctx.needs_input_grad[0] else None
        grad_bias = grad_output.reshape(-1, grad_output.shape[-1]).sum(0) if (bias is not None and ctx.needs_input_grad[2]) else None
        return grad_input, None, grad_bias


class HQQLinear(nn.Module):
    backend = HQQBackend.HIP   # class-wide default (the reference's is PYTORCH, quantize.py:389)

    def __init__(self, linear_layer: Union[nn.Module, None], quant_config: Union[dict, None], del_orig: bool = True,
                 compute_dtype: torch.dtype = float16, device: str = "cuda", initialize: bool = True):
        super().__init__()
        self.ready = False
        self.in_gpu = False
        self.bias = None
        self.axis = None
        self.channel_wise = None
        self.device = device
        self.compute_dtype = compute_dtype
        self.quant_config = copy.deepcopy(quant_config)
        self.del_orig = del_orig
        self.offload_meta = self.quant_config.pop("offload_meta") if (self.quant_config is not None) else None
        self.set_backend(HQQLinear.backend)
        self.linear_layer = linear_layer
        self.W_q = None
        self.meta = None
        self.encoded_state_dict = True   # state_dict() is safetensors-compatible by default
        if initialize:
            self.initialize()

    def is_initialized(self):
        return not (self.W_q is None or self.meta is None)

    def initialize(self):
        if self.linear_layer is None:
            return
        # quantised scale/zero and meta offloading are deprecated in the reference and ignored (quantize.py:434-439)
        self.quant_config["scale_quant_params"] = None
        self.quant_config["zero_quant_params"] = None
        wq = self.quant_config["weight_quant_params"]
        if wq["group_size"] is None:
            wq["group_size"] = self.linear_layer.in_features if (wq["axis"] == 1) else self.linear_layer.out_features
        self.quantize(self.linear_layer.weight.data, **self.quant_config)
        self.bias = None if (self.linear_layer.bias is None) else self.linear_layer.bias.clone().to(device=self.device, dtype=self.compute_dtype)
        if self.del_orig:
            for name, _ in list(self.linear_layer.named_parameters()):
                setattr(self.linear_layer, name, None)
            del self.linear_layer
            torch.cuda.empty_cache()

    @classmethod
    def from_weights(cls, weight: Tensor, bias: Union[Tensor, None], quant_config: dict, compute_dtype: torch.dtype = float16,
                     device: str = "cuda", del_orig: bool = True):
        shell = nn.Linear(1, 1, bias=False)
        shell.in_features, shell.out_features = weight.shape[1], weight.shape[0]
        shell.weight.data = weight
        # nn.Module refuses a plain Tensor in a registered parameter slot (the reference assigns it and raises, quantize.py:478)
        shell.bias = None if bias is None else nn.Parameter(bias, requires_grad=False)
        return cls(shell, quant_config=quant_config, compute_dtype=compute_dtype, device=device, del_orig=del_orig)

    @classmethod
    def merge(cls, layers):
        """ONE HQQLinear holding the rows of `layers` in order — modules that read the same input (q_proj | k_proj | v_proj, gate_proj | up_proj):
        merged(x) = torch.cat([l(x) for l in layers], -1) (the same weights bit for bit; outputs within the forward tolerance — how a launch cuts K may depend on its row count), from the layers' own levels, scale and zero (hqq_amd.ops.merge_layers: stacked and packed
        again, nothing re-quantised).  One launch over sum(out_features) rows instead of one per layer — what pays at 65..2560 activation rows
        (batched decode, speculative verification, short prompts).  The originals are untouched.  The reference has no such
        helper (its vLLM integration meets merged modules already merged: hqq/utils/vllm.py); the result is an ordinary HQQLinear: state_dict(),
        dequantize() and every backend work on it."""
        layers = list(layers)
        if not layers or any((not l.ready) or l.meta is None for l in layers):
            raise ValueError("hqq_amd: HQQLinear.merge takes quantised layers")
        first = layers[0]
        m0 = first.meta
        K = int(m0["shape"][1])
        for l in layers:
            m = l.meta
            if (m["axis"] != 1 or not m["packing"] or m["packing"] != m0["packing"] or m["group_size"] != m0["group_size"] or not m["group_size"]
                    or int(m["shape"][1]) != K or l.compute_dtype != first.compute_dtype or m.get("quant_scale") or m.get("quant_zero")
                    or (l.bias is None) != (first.bias is None)):
                raise ValueError("hqq_amd: HQQLinear.merge needs layers quantised along axis 1 with one packing, group size, input width and compute "
                                 "dtype, and either all or none with a bias")
        nb = Quantizer._packing_bits[m0["packing"]]
        parts = []
        for l in layers:
            W = l.W_q.data.view(l.meta["unpack_view_dtype"]) if l.meta["view_as_float"] else l.W_q.data
            parts.append((W, l.meta["scale"], l.meta["zero"], int(l.meta["shape"][0])))
        W, s, z, N = ops.merge_layers(parts, K, int(m0["group_size"]), nb)
        out = cls(None, quant_config=dict(copy.deepcopy(first.quant_config), offload_meta=first.offload_meta), compute_dtype=first.compute_dtype,
                  device=first.device, initialize=False)
        meta = {k: v for k, v in m0.items() if not isinstance(v, torch.Tensor)}
        meta.update({"shape": torch.Size([N, K]), "scale": s.reshape(-1, *m0["scale"].shape[1:]), "zero": z.reshape(-1, *m0["zero"].shape[1:])})
        if m0["view_as_float"]:
            W = W.view(first.W_q.dtype)
        out.W_q, out.meta = W, meta
        out.bias = None if first.bias is None else torch.cat([l.bias.reshape(-1) for l in layers])
        out.in_features, out.out_features = K, N
        out.axis, out.channel_wise = first.axis, first.channel_wise
        out.encoded_state_dict = first.encoded_state_dict
        out.cuda(first.device)
        out.ready = True
        return out

    def extra_repr(self) -> str:
        if getattr(self, "meta", None) is None:
            return ""
        in_features, out_features = self.meta["shape"][::-1]
        return f"in_features={in_features}, out_features={out_features}, bias={self.bias is not None}"

    @classmethod
    def set_backend(cls, backend: HQQBackend):
        """Rebinds HQQLinear.forward class-wide to the method the enum value names (quantize.py:498-512)."""
        HQQLinear.backend = backend
        cls.forward = getattr(cls, backend.value)

    def cuda(self, device):
        """Move the packed weights and the meta tensors (cast to compute_dtype) to `device` (quantize.py:515-583)."""
        self.meta["compute_dtype"] = self.compute_dtype
        W_q = self.W_q.data if isinstance(self.W_q, nn.Parameter) else self.W_q
        W_q, self.meta = Quantizer.cuda(W_q, self.meta, device)
        if self.bias is not None:
            self.bias = self.bias.to(device=device, dtype=self.compute_dtype)
        self.W_q = nn.Parameter(W_q, requires_grad=False)
        self.device = device
        self.in_gpu = True
        self._hip_opts = self._meta_opts()
        self._w3s = None   # (the 3-bit stream-layout copy of _matmul_hip: rebuilt on the next forward)
        return self

    def _meta_opts(self) -> int:
        """per-call option bits of the fused forward that depend on this layer's meta only (checked once, when it lands on the GPU)"""
        m = self.meta
        try:
            if (m["axis"] == 1 and m["scale"].dtype == float16 and m["packing"] in ("8bit_u8", "4bit_u8", "3bit_32", "2bit_u8", "1bit_u8")
                    and m["scale"].is_cuda and bool(m["group_size"])):
                N, K = m["shape"]
                if ops.meta_scalable(m["scale"].reshape(-1), m["zero"].reshape(-1), N, K, m["group_size"], Quantizer._packing_bits[m["packing"]]):
                    return ops.OPT_META_SCALABLE
        except (KeyError, TypeError, AttributeError):
            pass
        return 0

    # HF calls .to()/.half()/... on whole models; packed weights must not be touched (quantize.py:585-613)
    def to(self, *args, **kwargs):
        return self

    def type(self, dst_type):
        return self

    def half(self, *args, **kwargs):
        return self

    def bfloat16(self, *args, **kwargs):
        return self

    def float(self, *args, **kwargs):
        return self

    def double(self, *args, **kwargs):
        return self

    def cpu(self):
        return self

    # ---- state dict: flat, every non-tensor encoded as a small tensor (quantize.py:617-787) ----
    def state_dict_keys(self):
        return {"W_q", "nbits", "group_size", "shape", "scale", "zero", "axis", "packing", "unpack_view_dtype", "view_as_float",
                "quant_scale", "quant_zero", "compute_dtype", "bias", "offload_meta", "encoded_state_dict", "stores_quant_config",
                "channel_wise", "optimize", "round_zero"}

    def state_dict(self, *args, **kwargs):
        if not self.is_initialized():
            return {k: None for k in self.state_dict_keys()}
        enc = encode_safetensor_type if self.encoded_state_dict else (lambda z: z)
        state = {"W_q": self.W_q}
        state.update({k: enc(v) for k, v in self.meta.items()})
        if self.bias is not None:
            state["bias"] = self.bias
        state["offload_meta"] = enc(bool(self.offload_meta))
        if self.encoded_state_dict:
            state["encoded_state_dict"] = enc(True)
        state["stores_quant_config"] = enc(True)
        for k, v in self.quant_config["weight_quant_params"].items():
            state[k] = enc(v)
        if "destination" in kwargs and "prefix" in kwargs:
            for key, value in state.items():
                kwargs["destination"][kwargs["prefix"] + key] = value
        return state

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        layer_sd = {}
        for key in self.state_dict_keys():
            if prefix + key in state_dict:
                layer_sd[key] = state_dict.pop(prefix + key)
            elif key != "bias":
                missing_keys.append(prefix + key)
        if "W_q" in layer_sd:
            layer_sd["W_q"] = nn.Parameter(layer_sd["W_q"], requires_grad=False)
            self.load_state_dict(layer_sd, strict=strict)

    def load_state_dict(self, state_dict, strict=True, assign=False):
        state_dict = dict(state_dict)
        encoded = state_dict.pop("encoded_state_dict", None) is not None
        dec = decode_safetensor_type if encoded else (lambda z, t: z)
        if state_dict.pop("stores_quant_config", False) is not False:
            self.quant_config = {"weight_quant_params": {k: dec(state_dict[k], _META_TYPE[k]) for k in
                                                         ["nbits", "channel_wise", "group_size", "optimize", "round_zero", "axis", "view_as_float"]},
                                 "scale_quant_params": state_dict.pop("scale_quant_params", None),
                                 "zero_quant_params": state_dict.pop("zero_quant_params", None)}
        self.W_q = state_dict.pop("W_q")
        self.bias = state_dict.pop("bias", None)
        om = state_dict.pop("offload_meta", False)
        self.offload_meta = bool(dec(om, bool)) if isinstance(om, torch.Tensor) else bool(om)
        if "meta" in state_dict:
            self.meta = state_dict["meta"]   # pre-safetensors checkpoints
        else:
            self.meta = {k: dec(v, _META_TYPE[k]) for k, v in state_dict.items() if k in _META_TYPE}
        if "unpack_view_dtype" not in self.meta:
            self.meta["unpack_view_dtype"] = Quantizer.unpack_view_dtype[self.meta["packing"]]
        self.meta.setdefault("view_as_float", False)
        self.meta.setdefault("quant_scale", False)
        self.meta.setdefault("quant_zero", False)
        if self.meta["quant_scale"] or self.meta["quant_zero"]:
            raise NotImplementedError("hqq_amd: checkpoints with quantised scale/zero (deprecated in the reference) are not covered")
        self.cuda(self.device)
        self.ready = True
        self.in_features, self.out_features = self.meta["shape"][::-1]

    # ---- quantise / dequantise ----
    def quantize(self, W: Tensor, weight_quant_params: dict, scale_quant_params: Union[dict, None] = None,
                 zero_quant_params: Union[dict, None] = None) -> None:
        self.in_features, self.out_features = W.t().shape
        W_q, meta = Quantizer.quantize(W, device=self.device, compute_dtype=self.compute_dtype, **weight_quant_params)
        meta.update({"quant_scale": False, "quant_zero": False})
        self.W_q, self.meta = W_q, meta
        self.axis = weight_quant_params.get("axis")
        self.channel_wise = weight_quant_params.get("channel_wise")
        self.cuda(self.device)
        self.ready = True

    def unpack(self, reshape=False, dtype=None):
        if not self.ready or not self.meta["packing"]:
            return None
        W_q = self.W_q.view(self.meta["unpack_view_dtype"]) if self.meta["view_as_float"] else self.W_q
        W_r = Quantizer.unpack[self.meta["packing"]](W_q, dtype=dtype if (dtype is not None) else self.compute_dtype)
        return W_r.view(self.meta["shape"]) if reshape else W_r

    def dequantize(self):
        assert self.ready, "model was not quantized"
        return Quantizer.dequantize(self.W_q, self.meta)   # pure: meta is not mutated (the reference's is not re-entrant, :851-877)

    def matmul(self, x: Tensor, transpose: bool = True) -> Tensor:
        weight = self.dequantize()
        return torch.matmul(x, weight.t() if transpose else weight)

    # ---- forward methods named by HQQBackend ----
    def _fused_ok(self, x: Tensor) -> bool:
        """fp16 / bf16 layers quantised along axis 1 go through hqq_amd.ops.forward for every number of rows: the fused kernels where they
        cover the shape, the HIP dequantise kernel + the in-tree dense MFMA GEMM elsewhere (ops.forward decides).  Anything else —
        fp32 compute dtype, axis 0, group sizes that are not multiples of 16, quantised meta — keeps the reference's dequantise + matmul"""
        m = self.meta
        return (m["axis"] == 1 and bool(m["group_size"]) and m["group_size"] % 16 == 0 and x.dtype == m["scale"].dtype
                and x.dtype in (float16, torch.bfloat16) and m["packing"] in ("8bit_u8", "4bit_u8", "3bit_32", "2bit_u8", "1bit_u8")
                and x.is_cuda and self.W_q.is_cuda)

    def forward_hip(self, x: Tensor) -> Tensor:
        """Fused unpack -> dequantize -> GEMV / GEMM (one launch); shapes and prompt lengths the fused kernels do not cover run the HIP
        dequantise kernel + the in-tree dense GEMM (hqq_amd.ops.forward) — entirely on the GPU, never a CPU fallback."""
        if torch.is_grad_enabled() and x.requires_grad:
            return _MatmulNoCache.apply(x, self._matmul_hip, self.bias)
        return self._matmul_hip(x, transpose=True, bias=self.bias)

    # 3-bit layers on the set_backend(HQQBackend.HIP) route: the decode / GEMM kernels read the 3-bit STREAM layout (csrc/w3s.h; 2.4x the rate of the
    # reference container's two-launch path at one row).  W_q stays the reference's container — state_dict(), unpack(), dequantize() are untouched — and
    # the layer keeps a re-laid-out COPY beside it (3 bits per weight more), built on the first forward and rebuilt when W_q / scale / zero change.
    # HQQLinear.stream_layout_3bit = False keeps the single copy (and the slower path); prepare_for_inference(backend="hip") holds ONLY the stream layout.
    stream_layout_3bit = True

    def _w3s_copy(self, W_q: Tensor):
        m = self.meta
        N, K = m["shape"]
        # (tensors made under torch.inference_mode() carry no version counter: identity alone keys the copy there, as in backends/hip.py)
        ver = lambda t: None if t.is_inference() else t._version
        key = (W_q.data_ptr(), ver(W_q), m["scale"].data_ptr(), ver(m["scale"]), m["zero"].data_ptr(), ver(m["zero"]))
        c = getattr(self, "_w3s", None)
        if c is None or c[0] != key:
            if torch.cuda.is_current_stream_capturing():   # never built inside a capture (it allocates): the container's own path serves that step
                return None
            Ws = ops.w3s_pack(W_q.contiguous(), int(N), int(K))
            o = ops.OPT_W3S | (ops.OPT_META_SCALABLE if (m["scale"].dtype == float16 and ops.w3s_meta_scalable(m["scale"].reshape(-1), m["zero"].reshape(-1), int(N), int(K))) else 0)
            c = self._w3s = (key, Ws, o)
        return c[1], c[2]   # (in-place edits through W_q.data / meta[...].data bypass the version counters: call .cuda(device) again — it drops the copy — after such an edit)

    def _matmul_hip(self, x: Tensor, transpose: bool = True, bias=None) -> Tensor:
        if transpose and self._fused_ok(x):
            m = self.meta
            N, K = m["shape"]
            W_q = self.W_q.view(m["unpack_view_dtype"]) if m["view_as_float"] else self.W_q
            if m["packing"] == "3bit_32" and HQQLinear.stream_layout_3bit and ops.w3s_covers(int(N), int(K), m["group_size"]):
                got = self._w3s_copy(W_q)
                if got is not None:
                    return ops.forward(x, got[0], m["scale"], m["zero"], bias, N, K, m["group_size"], 3, opts=ops.layer_opts(got[1]))
            return ops.forward(x, W_q, m["scale"], m["zero"], bias, N, K, m["group_size"], Quantizer._packing_bits[m["packing"]],
                               opts=ops.layer_opts(getattr(self, "_hip_opts", 0)))
        out = self.matmul(x, transpose=transpose)
        if bias is not None:
            out += bias
        return out

    def forward_pytorch_backprop(self, x: Tensor) -> Tensor:
        return _MatmulNoCache.apply(x, self.matmul, self.bias)

    def forward_pytorch(self, x: Tensor) -> Tensor:
        out = torch.matmul(x, self.dequantize().t())
        if self.bias is not None:
            out += self.bias
        return out

    # the reference's *_compile variants wrap the same math in torch.compile and its ATen variants call the hqq_aten
    # dequantise kernel; both collapse onto the HIP dequantise kernel here (same arithmetic, both axes supported)
    forward_pytorch_backprop_compile = forward_pytorch_backprop
    forward_pytorch_compile = forward_pytorch
    forward_aten_backprop = forward_pytorch_backprop
    forward_aten = forward_pytorch
    matmul_compile = matmul                      # (quantize.py:884-886: torch.compile of the same call)
    dequantize_aten = dequantize                 # (quantize.py:899-976: the hqq_aten dequantise kernels; here one HIP kernel, both axes)
    dequantize_aten_with_streams = dequantize

    def forward_aten_int8(self, x: Tensor) -> Tensor:
        raise NotImplementedError("hqq_amd: the experimental int8-activation path (quantize.py:1034-1073) is not covered")

    forward = forward_hip


def hqq_base_quant_config(nbits: int = 4, group_size: int = 64, quant_zero: bool = False, quant_scale: bool = False,
                          offload_meta: bool = False, view_as_float: bool = False, axis: int = 1):
    """quantize.py:1076-1151.  quant_zero / quant_scale / offload_meta are deprecated there and ignored at initialize()."""
    assert nbits in Quantizer.SUPPORTED_BITS, "nbits value not supported. Check Quantizer.SUPPORTED_BITS."
    if group_size is not None:
        assert is_divisible(group_size, 8), "Invalid group_size param: the value should be a multiple of 8."
    weight_quant_params = {"nbits": nbits, "channel_wise": True, "group_size": group_size, "optimize": True,
                           "round_zero": True if nbits == 4 else False, "axis": axis, "view_as_float": view_as_float}
    meta8 = {"nbits": 8, "channel_wise": True, "group_size": 128, "optimize": False}
    scale_quant_params = dict(meta8) if quant_scale else None
    if offload_meta:
        zero_quant_params = dict(meta8) if quant_zero else None
    else:
        zero_quant_params = {"nbits": 8, "channel_wise": False, "group_size": None, "optimize": False} if quant_zero else None
    return {"weight_quant_params": weight_quant_params, "scale_quant_params": scale_quant_params,
            "zero_quant_params": zero_quant_params, "offload_meta": offload_meta}


BaseQuantizeConfig = hqq_base_quant_config
