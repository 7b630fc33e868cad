"""torch.library registrations of the two tensor-level entry points, so that torch.compile / export graphs keep them as
single opaque nodes (the reference does the same for its ATen kernel: `hqq::hqq_aten_dequantize` with a register_fake,
hqq/core/quantize.py:257-263).  Importing this module is optional; the eager path calls hqq_amd.ops directly."""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor

from . import ops


@torch.library.custom_op("hqq_hip::dequantize", mutates_args=())
def hqq_hip_dequantize(W_q: Tensor, scale: Tensor, zero: Tensor, N: int, K: int, group_size: int, nbits: int, axis: int) -> Tensor:
    return ops.dequantize(W_q, scale, zero, N, K, group_size, nbits, axis)


@hqq_hip_dequantize.register_fake
def _(W_q, scale, zero, N, K, group_size, nbits, axis):
    return torch.empty((N, K), device=W_q.device, dtype=scale.dtype)


@torch.library.custom_op("hqq_hip::forward", mutates_args=())
def hqq_hip_forward(x: Tensor, W_q: Tensor, scale: Tensor, zero: Tensor, bias: Optional[Tensor], N: int, K: int,
                    group_size: int, nbits: int) -> Tensor:
    return ops.forward(x, W_q, scale, zero, bias, N, K, group_size, nbits)


@hqq_hip_forward.register_fake
def _(x, W_q, scale, zero, bias, N, K, group_size, nbits):
    return torch.empty((*x.shape[:-1], N), device=x.device, dtype=x.dtype)
