"""BitPack — the reference's packing API (hqq/core/bitpack.py:9-144) on the gfx950 pack/unpack kernels.

Same method names and tensor contracts; the work is done by hqq_hip_pack / hqq_hip_unpack (csrc/bitpack.hip)
through the C ABI.  Tensors must live on the GPU: there is no CPU implementation in this package.
"""
from __future__ import annotations

import torch
from torch import Tensor, uint8

from .. import ops


def _packer(nbits: int):
    def pack(W_q: Tensor) -> Tensor:
        return ops.pack(nbits, W_q)
    pack.__doc__ = f"[rows, cols] integer levels -> packed {nbits}-bit container (bitpack.py pack_{nbits}bit_*)"
    return staticmethod(pack)


def _unpacker(nbits: int):
    def unpack(W_q: Tensor, dtype: torch.dtype = uint8) -> Tensor:
        return ops.unpack(nbits, W_q, dtype)
    unpack.__doc__ = f"packed {nbits}-bit container -> [per*rows, cols] levels of `dtype` (bitpack.py unpack_{nbits}bit_*)"
    return staticmethod(unpack)


class BitPack:
    pack_8bit_u8, unpack_8bit_u8 = _packer(8), _unpacker(8)
    pack_4bit_u8, unpack_4bit_u8 = _packer(4), _unpacker(4)
    pack_2bit_u8, unpack_2bit_u8 = _packer(2), _unpacker(2)
    pack_1bit_u8, unpack_1bit_u8 = _packer(1), _unpacker(1)
    # 3-bit: 10 levels per int32, rows zero-padded to a multiple of 10 (bitpack.py:69-110)
    pack_3bit_32, unpack_3bit_32 = _packer(3), _unpacker(3)
