"""Model-level patching, mirror of hqq/utils/patching.py:26-177 for the HIP backend."""
from __future__ import annotations

from ..backends.hip import HQQLinearHIP, patch_hqq_to_hip
from ..core.quantize import BaseQuantizeConfig, HQQLinear
from ..core.utils import cleanup


def _is_hqq(layer) -> bool:
    # HQQLinearLoRA-style wrappers expose the quantised layer as `.linear_layer` (hqq/core/peft.py:150-165)
    return isinstance(layer, HQQLinear) or isinstance(getattr(layer, "linear_layer", None), (HQQLinear, HQQLinearHIP))


def patch_linearlayers(model, fct, patch_param=None, verbose=False):
    """calls fct(layer, patch_param) on every HQQLinear(-wrapping) child and installs the returned module (patching.py:26-35)"""
    for name, layer in model.named_children():
        if _is_hqq(layer):
            setattr(model, name, fct(layer, patch_param))
        else:
            patch_linearlayers(layer, fct, patch_param, verbose)


def patch_add_quant_config(layer, patch_param):
    """rebuilds a missing quant_config from meta (patching.py:39-61)"""
    target = layer if isinstance(layer, HQQLinear) else getattr(layer, "linear_layer", None)
    if isinstance(target, HQQLinear):
        if patch_param is not None:
            target.quant_config = patch_param
        if target.quant_config is None:
            m = target.meta
            target.quant_config = BaseQuantizeConfig(nbits=m["nbits"], group_size=m["group_size"], axis=m["axis"])
    return layer


def autoset_quant_config(hqq_layer, patch_param=None):
    """patching.py:38-51: install `patch_param` as the layer's quant_config, or rebuild a missing one from meta"""
    if patch_param is not None:
        hqq_layer.quant_config = patch_param
    if hqq_layer.quant_config is None:
        m = hqq_layer.meta
        hqq_layer.quant_config = BaseQuantizeConfig(nbits=m["nbits"], group_size=m["group_size"], axis=m["axis"],
                                                    quant_scale=m.get("quant_scale", False), quant_zero=m.get("quant_zero", False))
    return hqq_layer


def patch_add_weight_param(layer, patch_param):
    """patching.py:62-78: a dummy `weight` parameter for code that asks a linear layer for `.weight.device` / `.weight.dtype`"""
    import torch
    if not hasattr(layer, "weight"):
        if hasattr(layer, "device"):
            device_ = layer.device
        else:
            params = list(layer.parameters())
            device_ = params[0].device if params else patch_param["device"]
        fp = [p for p in layer.parameters() if p.is_floating_point()]
        dtype_ = fp[0].dtype if fp else patch_param["dtype"]
        layer.weight = torch.nn.Parameter(torch.zeros((1,), device=device_, dtype=dtype_), requires_grad=False)
    return layer


def patch_hqq_inference(layer, patch_param=None):
    """patching.py:81-98: bind an inference-only forward to the instance.  The reference's is `x @ dequantize().T + bias` with the
    note "TODO GEMV use-case" (:84); here it is that use-case — HQQLinear.forward_hip, one fused launch — without the autograd wrapper."""
    def forward_hqq_inference(self, x):
        return self._matmul_hip(x.to(self.device), transpose=True, bias=self.bias)

    target = layer if type(layer) is HQQLinear else getattr(layer, "linear_layer", None)
    if type(target) is HQQLinear:
        target.forward = lambda x, _t=target: forward_hqq_inference(_t, x)
    return layer


def patch_lora_inference(layer, patch_param=None):
    """patching.py:101-109: the low-rank branch of an HQQLinearLoRA-style wrapper as two plain matmuls (host code; no kernel of this path)"""
    import torch
    if all(hasattr(layer, a) for a in ("lora_A", "lora_B", "scaling", "linear_layer")):
        layer.forward_lora = lambda x, _l=layer: torch.matmul(torch.matmul(x, _l.lora_A), _l.lora_B) * _l.scaling
    return layer


def prepare_for_inference(model, allow_merge=False, backend="hip", verbose=False):
    """backend "hip" (default here) swaps every covered HQQLinear for HQQLinearHIP; "default" only makes sure the class-wide
    forward is the fused HIP one.  The reference's external backends (torchao_int4 / gemlite / bitblas / marlin) are CUDA
    packages and are rejected loudly."""
    patch_linearlayers(model, patch_add_quant_config, patch_param=None)
    if backend in ("hip", "hqq_hip"):
        patch_linearlayers(model, patch_hqq_to_hip, verbose=verbose)
    elif backend == "default":
        patch_linearlayers(model, patch_hqq_inference)   # patching.py:133-134
        patch_linearlayers(model, patch_lora_inference)
    else:
        raise RuntimeError(f"hqq_amd: backend '{backend}' is a CUDA package of the reference and is not available here; use backend='hip'")
    cleanup()
    return model
