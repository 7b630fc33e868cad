"""numpy front-end of the CPU oracle (oracle/hqq_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (hqq_amd/) never does.  See the header of hqq_oracle.c for what is restated, the
reference file:line of every function and how the restatement is pinned to the reference.

Arrays: float32 = np.float32, float16 = np.float16, bfloat16 = raw np.uint16 bit patterns.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libhqq_oracle.so")
_lib = None

F32, F16, BF16 = 0, 1, 2

# Quantizer.bit_to_packing / max value, hqq/core/quantize.py:40-49, :121
PER = {8: 1, 4: 2, 2: 4, 1: 8, 3: 10}


def build(force: bool = False) -> str:
    """Compile the C oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "hqq_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        i64, i32, vp, f32 = ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_float
        L.hqq_oracle_version.restype = i32
        L.hqq_oracle_packed_rows.restype = i64
        L.hqq_oracle_packed_rows.argtypes = [i32, i64]
        L.hqq_oracle_pack.argtypes = [i32, vp, i64, i64, vp]
        L.hqq_oracle_unpack.argtypes = [i32, vp, i64, i64, vp]
        L.hqq_oracle_row_sum_f32.restype = f32
        L.hqq_oracle_row_sum_f32.argtypes = [vp, i64]
        L.hqq_oracle_quantize.argtypes = [vp, i64, i32, i32, i32, i32, i32, f32, f32, vp, vp, vp, vp]
        L.hqq_oracle_quantize_axis0.argtypes = [vp, i64, i32, i32, i32, i32, i32, f32, f32, vp, vp, vp, vp]
        L.hqq_oracle_dequantize.argtypes = [i32, vp, vp, vp, vp, i64, i64, i32, i32]
        L.hqq_oracle_matmul.argtypes = [vp, vp, vp, vp, vp, i64, i64, i64, i32]
        L.hqq_oracle_forward.argtypes = [i32, vp, vp, vp, vp, vp, vp, i64, i64, i64, i32, i32]
        L.hqq_oracle_f32_to_f16.argtypes = [vp, vp, i64]
        L.hqq_oracle_f16_to_f32.argtypes = [vp, vp, i64]
        L.hqq_oracle_f32_to_bf16.argtypes = [vp, vp, i64]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dtype=None):
    return np.ascontiguousarray(a, dtype=dtype)


def dtype_code(np_dtype_or_name) -> int:
    n = str(np_dtype_or_name)
    if "bfloat16" in n or n == "bf16":
        return BF16
    if "float16" in n or n == "f16":
        return F16
    if "float32" in n or n == "f32":
        return F32
    raise ValueError(n)


def to_cd(a32: np.ndarray, code: int) -> np.ndarray:
    """float32 -> compute dtype (one RNE rounding).  bf16 comes back as raw uint16."""
    a32 = _c(a32, np.float32)
    if code == F32:
        return a32.copy()
    out = np.empty(a32.shape, np.uint16)
    (lib().hqq_oracle_f32_to_f16 if code == F16 else lib().hqq_oracle_f32_to_bf16)(_p(a32), _p(out), a32.size)
    return out.view(np.float16) if code == F16 else out


def from_cd(a: np.ndarray, code: int) -> np.ndarray:
    """compute dtype -> float32 (exact)."""
    if code == F32:
        return _c(a, np.float32)
    if code == F16:
        return _c(a).view(np.float16).astype(np.float32)
    return (_c(a).view(np.uint16).astype(np.uint32) << 16).view(np.float32)


# ---------------------------------------------------------------------------------------------
# BitPack — C restatement plus an independent pure-numpy one (they must agree; tests check it)
# ---------------------------------------------------------------------------------------------
def packed_rows(nbits: int, R: int) -> int:
    return int(lib().hqq_oracle_packed_rows(int(nbits), int(R)))


def pack(nbits: int, U: np.ndarray) -> np.ndarray:
    U = _c(U, np.uint8)
    R, C = U.shape
    P = packed_rows(nbits, R)
    if P < 0:
        raise ValueError(f"cannot pack {R} rows at {nbits} bits")
    out = np.empty((P, C), np.int32 if nbits == 3 else np.uint8)
    rc = lib().hqq_oracle_pack(nbits, _p(U), R, C, _p(out))
    assert rc == 0, rc
    return out


def unpack(nbits: int, Pk: np.ndarray) -> np.ndarray:
    Pk = _c(Pk, np.int32 if nbits == 3 else np.uint8)
    P, C = Pk.shape
    out = np.empty((PER[nbits] * P, C), np.uint8)
    rc = lib().hqq_oracle_unpack(nbits, _p(Pk), P, C, _p(out))
    assert rc == 0, rc
    return out


def pack_np(nbits: int, U: np.ndarray) -> np.ndarray:
    """pure numpy, follows hqq/core/bitpack.py:24-28, 43-52, 69-91, 115-128 slab by slab."""
    U = np.asarray(U, np.uint8)
    per = PER[nbits]
    if nbits == 8:
        return U.copy()
    if nbits == 3:
        step = -(-U.shape[0] // 10)
        Z = np.zeros((10 * step, U.shape[1]), np.int32)
        Z[: U.shape[0]] = U
        out = np.zeros((step, U.shape[1]), np.int32)
        for s in range(10):
            out |= Z[s * step:(s + 1) * step] << (27 - 3 * s)
        return out
    step = U.shape[0] // per
    out = np.zeros((step, U.shape[1]), np.uint8)
    for s in range(per):
        out |= (U[s * step:(s + 1) * step] << (nbits * (per - 1 - s))).astype(np.uint8)
    return out


def unpack_np(nbits: int, Pk: np.ndarray) -> np.ndarray:
    per = PER[nbits]
    if nbits == 8:
        return np.asarray(Pk, np.uint8).copy()
    if nbits == 3:
        Pk = np.asarray(Pk, np.int32)
        return np.concatenate([((Pk >> (27 - 3 * s)) & 7).astype(np.uint8) for s in range(10)], 0)
    Pk = np.asarray(Pk, np.uint8)
    m = (1 << nbits) - 1
    return np.concatenate([((Pk >> (nbits * (per - 1 - s))) & m).astype(np.uint8) for s in range(per)], 0)


# ---------------------------------------------------------------------------------------------
# The 3-bit STREAM layout of this build (hqq_amd/csrc/w3s.h) — no reference counterpart: like the reference's optimised backends
# (hqq/backends/torchao.py:202-241, marlin.py:74-123) HQQLinearHIP re-lays the levels out when a layer is patched.  Restated here,
# independently of the HIP code, as the checker of hqq_hip_w3s_pack / hqq_hip_w3s_unpack: [N/2, K/16, 3] uint32, packed row p = output
# rows p and p + N/2, chunk c = their k 16c..16c+15; a dword holds 5 pair fields (bits [3f, 3f+3) = the pair's even k, [16+3f, +3) its odd k).
# ---------------------------------------------------------------------------------------------
def w3s_pos(s: int, i: int):
    """(dword, bit offset) of level (slab s, i = k % 16) of a chunk; dword -1: the scattered pair — bit t of the level is bit `offset` of dword t"""
    j, h = i >> 1, (i & 1) * 16
    if s == 0:
        return (0, 3 * j + h) if j < 5 else (2, 3 * (j - 5) + h)
    if j < 5:
        return (1, 3 * j + h)
    if j < 7:
        return (2, 3 * (j - 2) + h)
    return (-1, 15 + h)


def w3s_from_levels(L: np.ndarray) -> np.ndarray:
    """levels [N, K] (0..7) -> stream layout [N/2, K/16, 3] uint32"""
    L = np.asarray(L, np.uint8)
    N, K = L.shape
    assert N % 2 == 0 and K % 64 == 0
    out = np.zeros((N // 2, K // 16, 3), np.uint32)
    for s in range(2):
        Ls = L[s * (N // 2):(s + 1) * (N // 2)].reshape(N // 2, K // 16, 16).astype(np.uint32)
        for i in range(16):
            d, b = w3s_pos(s, i)
            if d >= 0:
                out[:, :, d] |= Ls[:, :, i] << np.uint32(b)
            else:
                for t in range(3):
                    out[:, :, t] |= ((Ls[:, :, i] >> np.uint32(t)) & np.uint32(1)) << np.uint32(b)
    return out


def w3s_to_levels(D: np.ndarray, N: int, K: int) -> np.ndarray:
    D = np.asarray(D, np.uint32).reshape(N // 2, K // 16, 3)
    L = np.zeros((N, K), np.uint8)
    for s in range(2):
        Ls = np.zeros((N // 2, K // 16, 16), np.uint32)
        for i in range(16):
            d, b = w3s_pos(s, i)
            if d >= 0:
                Ls[:, :, i] = (D[:, :, d] >> np.uint32(b)) & np.uint32(7)
            else:
                for t in range(3):
                    Ls[:, :, i] |= ((D[:, :, t] >> np.uint32(b)) & np.uint32(1)) << np.uint32(t)
        L[s * (N // 2):(s + 1) * (N // 2)] = Ls.reshape(N // 2, K).astype(np.uint8)
    return L


def w3s_pack_np(ref_container: np.ndarray, N: int, K: int) -> np.ndarray:
    """the reference's 3-bit container ([ceil(R/10), 64] int32, bitpack.py:69-91) -> stream layout"""
    R = N * K // 64
    return w3s_from_levels(unpack_np(3, ref_container)[:R].reshape(N, K))


def w3s_unpack_np(w3s: np.ndarray, N: int, K: int) -> np.ndarray:
    """stream layout -> the reference's container, bit for bit (zero padding rows included)"""
    return pack_np(3, w3s_to_levels(w3s, N, K).reshape(N * K // 64, 64))


# ---------------------------------------------------------------------------------------------
# Quantizer.quantize (axis=1) with the legacy proximal solver, float32
# ---------------------------------------------------------------------------------------------
def max_v_of(nbits) -> int:
    return int(round(2 ** nbits - 1))  # quantize.py:121


def quantize(W: np.ndarray, nbits=4, group_size: int = 64, round_zero=None, optimize: bool = True,
             iters: int = 20, beta: float = 10.0, lp_norm: float = 0.7):
    """returns dict(Wq [R,gs] uint8, scale [R,1] f32 (= 1/scale), zero [R,1] f32, iters_run, err_hist)."""
    W = _c(W, np.float32)
    if round_zero is None:
        round_zero = nbits == 4  # hqq_base_quant_config, quantize.py:1097
    R = W.size // group_size
    Wq = np.empty((R, group_size), np.uint8)
    sc = np.empty((R, 1), np.float32)
    ze = np.empty((R, 1), np.float32)
    err = np.full((iters,), np.nan, np.float64)
    rc = lib().hqq_oracle_quantize(_p(W), W.size, group_size, max_v_of(nbits), int(bool(round_zero)), int(bool(optimize)),
                                   iters, beta, lp_norm, _p(Wq), _p(sc), _p(ze), _p(err))
    if rc < 0:
        raise ValueError(f"hqq_oracle_quantize rc={rc}")
    return {"Wq": Wq, "scale": sc, "zero": ze, "iters_run": rc, "err_hist": err}


def quantize_axis0(W: np.ndarray, nbits=4, group_size: int = 64, round_zero=None, optimize: bool = True,
                   iters: int = 20, beta: float = 10.0, lp_norm: float = 0.7):
    """Quantizer.quantize(axis=0): returns dict(Wq [gs, C] uint8, scale [1, C] f32 (= 1/scale), zero [1, C] f32, iters_run, err_hist), C = numel / gs."""
    W = _c(W, np.float32)
    if round_zero is None:
        round_zero = nbits == 4
    C = W.size // group_size
    Wq = np.empty((group_size, C), np.uint8)
    sc = np.empty((1, C), np.float32)
    ze = np.empty((1, C), np.float32)
    err = np.full((iters,), np.nan, np.float64)
    rc = lib().hqq_oracle_quantize_axis0(_p(W), W.size, group_size, max_v_of(nbits), int(bool(round_zero)), int(bool(optimize)),
                                         iters, beta, lp_norm, _p(Wq), _p(sc), _p(ze), _p(err))
    if rc < 0:
        raise ValueError(f"hqq_oracle_quantize_axis0 rc={rc}")
    return {"Wq": Wq, "scale": sc, "zero": ze, "iters_run": rc, "err_hist": err}


def quantize_tensorwise(W: np.ndarray, nbits=4, round_zero: bool = False):
    """Quantizer.quantize(channel_wise=False) (quantize.py:114-116,146): the whole tensor is one group, no solver; the levels keep the
    tensor's shape.  returns dict(Wq [rows, cols] uint8, scale 0-d f32 (= 1/scale), zero 0-d f32)."""
    W = _c(W, np.float32)
    r = quantize(W, nbits=nbits, group_size=W.size, round_zero=round_zero, optimize=False)
    return {"Wq": r["Wq"].reshape(W.shape), "scale": r["scale"].reshape(()), "zero": r["zero"].reshape(())}


def row_sum(x: np.ndarray) -> float:
    x = _c(x, np.float32)
    return float(lib().hqq_oracle_row_sum_f32(_p(x), x.size))


# ---------------------------------------------------------------------------------------------
# Quantizer.dequantize / HQQLinear forward
# ---------------------------------------------------------------------------------------------
def dequantize(nbits: int, packed: np.ndarray, scale_cd: np.ndarray, zero_cd: np.ndarray, N: int, K: int,
               group_size: int, code: int) -> np.ndarray:
    """scale_cd/zero_cd already in the compute dtype (np.float16 / raw-uint16 bf16 / float32). Returns [N,K] in cd."""
    packed = _c(packed, np.int32 if nbits == 3 else np.uint8)
    scale_cd, zero_cd = _c(scale_cd), _c(zero_cd)
    out = np.empty((N, K), np.float32 if code == F32 else np.uint16)
    rc = lib().hqq_oracle_dequantize(nbits, _p(packed), _p(scale_cd), _p(zero_cd), _p(out), N, K, group_size, code)
    assert rc == 0, rc
    return out.view(np.float16) if code == F16 else out


def matmul(x_cd: np.ndarray, Wd_cd: np.ndarray, bias_cd, code: int):
    """y = x @ Wd^T (+bias): double accumulation, one rounding to cd.  returns (y_cd, y_f32_unrounded)."""
    x_cd, Wd_cd = _c(x_cd), _c(Wd_cd)
    M, K = x_cd.shape
    N = Wd_cd.shape[0]
    y = np.empty((M, N), np.float32 if code == F32 else np.uint16)
    y32 = np.empty((M, N), np.float32)
    b = None if bias_cd is None else _c(bias_cd)
    rc = lib().hqq_oracle_matmul(_p(x_cd), _p(Wd_cd), _p(b), _p(y), _p(y32), M, N, K, code)
    assert rc == 0, rc
    return (y.view(np.float16) if code == F16 else y), y32


def forward(nbits, packed, scale_cd, zero_cd, bias_cd, x_cd, N, K, group_size, code):
    """dequantize + matmul, the per-call work of HQQBackend.PYTORCH (quantize.py:894-898)."""
    packed = _c(packed, np.int32 if nbits == 3 else np.uint8)
    x_cd = _c(x_cd)
    M = x_cd.shape[0]
    y = np.empty((M, N), np.float32 if code == F32 else np.uint16)
    b = None if bias_cd is None else _c(bias_cd)
    rc = lib().hqq_oracle_forward(nbits, _p(packed), _p(_c(scale_cd)), _p(_c(zero_cd)), _p(b), _p(x_cd), _p(y),
                                  M, N, K, group_size, code)
    assert rc == 0, rc
    return y.view(np.float16) if code == F16 else y
