"""Host mirror of hqq/core/optimize.py for the proximal solver on its own (the solver normally runs inside Quantizer.quantize, fused with
the min/max initialisation and the bit-packing: hqq_amd.ops.quantize).  Compute is the HIP solver (csrc/quantize.hip) — the reference's
CPU float32 arithmetic, bit for bit; there is no CPU path."""
from typing import Union

import torch
from torch import Tensor

from .. import ops


def shrink_lp_op(x: Tensor, beta: float, lp_norm: float) -> Tensor:
    """optimize.py:96-108 (generalised soft-thresholding); elementwise torch ops on the caller's device — the solver kernels fuse it"""
    if lp_norm == 1:
        return torch.sign(x) * torch.nn.functional.relu(torch.abs(x) - 1.0 / beta)
    return torch.sign(x) * torch.nn.functional.relu(torch.abs(x) - (1.0 / beta) * torch.pow(torch.abs(x), lp_norm - 1))


def optimize_weights_proximal_legacy(tensor: Tensor, scale: Tensor, zero: Tensor, min_max: list, axis: int = 0, device: Union[str, None] = None,
                                     opt_params: dict = {"lp_norm": 0.7, "beta": 1e1, "kappa": 1.01, "iters": 20}, verbose: bool = False) -> tuple:
    """optimize.py:208-255.  tensor: the grouped view Quantizer.quantize builds — [groups, group_size] with axis=1, [group_size, groups] with
    axis=0 —, scale / zero: one value per group ([groups, 1] / [1, groups]).  Returns (W_q levels as a float tensor, scale unchanged, zero float32)."""
    if tensor.dim() != 2 or axis not in (0, 1):
        raise ValueError("hqq_amd: optimize_weights_proximal_legacy takes the 2-D grouped view and axis 0 or 1")
    if min_max[0] != 0:
        raise NotImplementedError("hqq_amd: levels start at 0 (min_max[0] == 0), as Quantizer.quantize sets them")
    dev = tensor.device if device is None else torch.device(device)
    W = tensor.to(dev)
    W_q, zero_new = ops.optimize(W, scale.to(dev).float(), zero.to(dev).float(), int(min_max[1]), axis=axis, iters=int(opt_params["iters"]),
                                 beta=float(opt_params["beta"]), lp_norm=float(opt_params["lp_norm"]))
    return W_q.to(torch.float32 if tensor.dtype not in (torch.float16, torch.bfloat16, torch.float32) else tensor.dtype).to(tensor.device), scale.to(tensor.device), zero_new.to(tensor.device)


def optimize_weights_proximal_legacy_step(W_f: Tensor, scale: Tensor, zero: Tensor, min_max: list, beta: float, lp_norm: float, axis: int) -> tuple:
    """optimize.py:201-206, one half-quadratic step: returns (W_r, W_q, new zero, scale).  The new zero-point — the only reduction of
    the step — comes from the HIP solver run for one iteration (ATen's CPU float32 summation order, bit for bit); W_q and W_r are the
    step's elementwise expressions evaluated on the device (round / clamp / subtract / divide: one IEEE operation each, no contraction)."""
    if W_f.dim() != 2 or axis not in (0, 1):
        raise ValueError("hqq_amd: optimize_weights_proximal_legacy_step takes the 2-D grouped view and axis 0 or 1")
    if min_max[0] != 0:
        raise NotImplementedError("hqq_amd: levels start at 0 (min_max[0] == 0), as Quantizer.quantize sets them")
    if not W_f.is_cuda:
        raise RuntimeError("hqq_amd: optimize_weights_proximal_legacy_step has no CPU path (tensor on %s)" % W_f.device)
    scale_f, zero_f = scale.to(W_f.device).float(), zero.to(W_f.device).float()
    Wf = W_f.float()
    W_q = torch.round(Wf * scale_f + zero_f).clamp_(min_max[0], min_max[1])
    W_r = (W_q - zero_f) / scale_f
    _, zero_new = ops.optimize(W_f, scale_f, zero_f, int(min_max[1]), axis=axis, iters=1, beta=float(beta), lp_norm=float(lp_norm))
    return W_r, W_q, zero_new, scale


# the reference's aliases (optimize.py:259)
optimize_weights_proximal = optimize_weights_proximal_legacy
