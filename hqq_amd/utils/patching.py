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


def prepare_for_inference(model, allow_merge=False, backend="hip", verbose=False):
    """backend "hip" (default here) swaps every covered HQQLinear for HQQLinearHIP; "default" only makes sure the class-wide
    forward is the fused HIP one.  The reference's external backends (torchao_int4 / gemlite / bitblas / marlin) are CUDA
    packages and are rejected loudly."""
    patch_linearlayers(model, patch_add_quant_config, patch_param=None)
    if backend in ("hip", "hqq_hip"):
        patch_linearlayers(model, patch_hqq_to_hip, verbose=verbose)
    elif backend != "default":
        raise RuntimeError(f"hqq_amd: backend '{backend}' is a CUDA package of the reference and is not available here; use backend='hip'")
    cleanup()
    return model
