"""Small host helpers shared by the HQQLinear surface (mirror of hqq/core/utils.py:10-70)."""
from __future__ import annotations

import gc

import torch


def cleanup() -> None:
    if torch.cuda.is_available():
        torch.cuda.empty_cache()
    gc.collect()


def is_divisible(val1: int, val2: int) -> bool:
    return val2 != 0 and val1 % val2 == 0


# ---- state-dict scalar <-> tensor coding (hqq/core/utils.py:37-70): safetensors can only hold tensors ----------
def encode_safetensor_type(data):
    """bool -> uint8 scalar, int -> int32 scalar, float -> float32 scalar, str / torch.dtype -> uint8 char codes,
    torch.Size -> int64 vector; tensors pass through.  Anything else (e.g. None) encodes to None, as in the reference."""
    if isinstance(data, torch.Tensor):
        return data
    if isinstance(data, torch.Size):
        return torch.tensor(data)
    if isinstance(data, torch.dtype):
        data = str(data)
    if isinstance(data, bool):          # before int: bool is an int
        return torch.tensor(int(data), dtype=torch.uint8)
    if isinstance(data, int):
        return torch.tensor(data, dtype=torch.int32)
    if isinstance(data, float):
        return torch.tensor(data, dtype=torch.float32)
    if isinstance(data, str):
        return torch.tensor([ord(c) for c in data], dtype=torch.uint8)
    return None


_DTYPE_BY_NAME = {str(d): d for d in (torch.float16, torch.bfloat16, torch.float32, torch.float64, torch.uint8, torch.int8,
                                      torch.int16, torch.int32, torch.int64, torch.bool)}


def decode_safetensor_type(data, data_type):
    if data_type in (torch.Tensor, torch.nn.Parameter):
        return data
    if data_type is torch.Size:
        return torch.Size(int(v) for v in data)
    if data_type is bool:
        return bool(data.item())
    if data_type is int:
        return int(data.item())
    if data_type is float:
        return float(data.item())
    text = "".join(chr(int(c)) for c in data)
    if data_type is str:
        return text
    if data_type is torch.dtype:
        # the reference eval()s the string (utils.py:69-70); a table lookup decodes the same strings without eval
        try:
            return _DTYPE_BY_NAME[text]
        except KeyError:
            raise ValueError(f"unknown dtype string in state_dict: {text!r}") from None
    raise TypeError(f"cannot decode to {data_type}")


def zero_pad_row(tensor: torch.Tensor, num_rows: int, dtype: Union[torch.dtype, None] = None) -> torch.Tensor:
    """hqq/core/utils.py:22-32: `tensor` on top of a zero matrix of `num_rows` rows (the 3-bit packer pads its rows to a multiple of ten with it)"""
    out = torch.zeros([num_rows, tensor.shape[1]], device=tensor.device, dtype=tensor.dtype if (dtype is None) else dtype)
    out[: len(tensor)] = tensor
    return out
