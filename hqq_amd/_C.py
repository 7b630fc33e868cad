"""ctypes binding of libhqq_hip.so (C ABI in include/hqq_hip.h).

There is deliberately no fallback: if the library is missing or a symbol is absent the import of
the native layer fails loudly (RuntimeError) — the product path never silently degrades to eager
PyTorch or to the CPU oracle.  Build with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C hqq_amd/csrc`.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HQQ_AMD_LIB") or os.path.join(_HERE, "lib", "libhqq_hip.so")   # the override serves lab builds (tools/)
CSRC = os.path.join(_HERE, "csrc")

# every symbol include/hqq_hip.h declares, with its ctypes signature
_i64, _i32, _vp, _f32, _sz, _u32 = ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_float, ctypes.c_size_t, ctypes.c_uint32
SYMBOLS = {
    "hqq_hip_abi_version": (_i32, []),
    "hqq_hip_last_error": (ctypes.c_char_p, []),
    "hqq_hip_packed_rows": (_i64, [_i32, _i64]),
    "hqq_hip_pack": (_i32, [_i32, _vp, _i32, _i64, _i64, _vp, _vp]),
    "hqq_hip_unpack": (_i32, [_i32, _vp, _i64, _i64, _vp, _i32, _vp]),
    "hqq_hip_dequantize": (_i32, [_i32, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _vp]),
    "hqq_hip_meta_check": (_i32, [_i32, _vp, _vp, _i64, _i64, _i64, _i32, _vp, _vp]),
    "hqq_hip_w3s_pack": (_i32, [_vp, _vp, _i64, _i64, _vp]),
    "hqq_hip_w3s_unpack": (_i32, [_vp, _vp, _i64, _i64, _vp]),
    "hqq_hip_w3s_meta_check": (_i32, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "hqq_hip_add_rmsnorm": (_i32, [_vp, _vp, _vp, _f32, _vp, _i64, _i64, _i32, _vp]),
    "hqq_hip_rope_cache": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _vp]),
    "hqq_hip_silu_mul": (_i32, [_vp, _vp, _vp, _i64, _i32, _vp]),
    "hqq_hip_token_prologue": (_i32, [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _i32, _vp]),
    "hqq_hip_argmax_advance": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp, _vp]),
    "hqq_hip_attn_decode": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, ctypes.c_float, _i32, _i64, _vp, _sz, _vp]),
    "hqq_hip_rope_attn_decode": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, ctypes.c_float, _i32, _i64, _vp, _sz, _vp]),
    "hqq_hip_attn_decode_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "hqq_hip_gemv_workspace_bytes": (_sz, [_i32, _i32, _vp, _i64, _i64, _i64, _i32, _u32]),
    "hqq_hip_gemv": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _u32, _vp, _sz, _vp]),
    "hqq_hip_gemv_grouped": (_i32, [_i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _u32, _vp, _sz, _vp]),
    "hqq_hip_gemv_block": (_i32, [_i32, _i32, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _u32, _u32, _vp, _vp]),
    "hqq_hip_exchange": (_i32, [_i32, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _u32, _vp]),
    "hqq_hip_gemm": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _u32, _vp, _sz, _vp]),
    "hqq_hip_gemm_grouped_covers": (_i32, [_i32, _i32, _vp, _i64, _i64, _i64, _i32, _u32]),
    "hqq_hip_gemm_grouped_workspace_bytes": (_sz, [_i32, _i32, _vp, _i64, _i64, _i64, _i32, _u32]),
    "hqq_hip_gemm_grouped": (_i32, [_i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _u32, _vp, _sz, _vp]),
    "hqq_hip_gemm_dense": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _vp]),
    "hqq_hip_forward_workspace_bytes": (_sz, [_i32, _i64, _i64, _i64, _i64, _i32, _u32]),
    "hqq_hip_gemm_workspace_bytes": (_sz, [_i32, _i64, _i64, _i64, _i64, _i32, _u32]),
    "hqq_hip_gemm_plan": (_i32, [_i32, _i64, _i64, _i64, _i64, _i32, _u32, _vp]),
    "hqq_hip_forward_prefers_fused": (_i32, [_i32, _i64, _i64, _i64, _i64, _i32]),
    "hqq_hip_forward": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _u32, _vp, _sz, _vp]),
    "hqq_hip_quantize_workspace_bytes": (_sz, [_i64, _i64, _i32]),
    "hqq_hip_quantize": (_i32, [_vp, _i32, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _f32, _f32,
                                _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "hqq_hip_quantize_axis0": (_i32, [_vp, _i32, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _f32, _f32,
                                      _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "hqq_hip_optimize": (_i32, [_vp, _i32, _i64, _i64, _i32, _i32, _vp, _vp, _i32, _f32, _f32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "hqq_hip_quantize_tensor": (_i32, [_vp, _i32, _i64, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
}

ABI_VERSION = 8
_lib = None


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP sources for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j", str(min(8, os.cpu_count() or 1))]
    if force:
        cmd.append("-B")
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hqq_amd: building libhqq_hip.so failed:\n" + res.stdout[-4000:] + res.stderr[-4000:])
    if verbose:
        print(res.stdout[-2000:])
    return LIB_PATH


def lib() -> ctypes.CDLL:
    """Load libhqq_hip.so (after torch, so that torch's libamdhip64.so.7 is the one HIP runtime in the process)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"hqq_amd: native library {LIB_PATH} is missing. It is not optional: run "
            f"`make -C {CSRC}` (or __graft_entry__.build()). There is no PyTorch/CPU fallback.")
    import torch  # noqa: F401  (loads the HIP runtime this library must share)
    L = ctypes.CDLL(LIB_PATH)
    missing = [s for s in SYMBOLS if not hasattr(L, s)]
    if missing:
        raise RuntimeError(f"hqq_amd: {LIB_PATH} lacks symbols {missing}; rebuild it")
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    v = L.hqq_hip_abi_version()
    if v != ABI_VERSION:
        raise RuntimeError(f"hqq_amd: ABI version {v} != expected {ABI_VERSION}; rebuild libhqq_hip.so")
    _lib = L
    return L


def last_error() -> str:
    return lib().hqq_hip_last_error().decode("utf-8", "replace")


def check(rc: int, what: str) -> None:
    if rc == 0:
        return
    msg = last_error()
    if rc == -4:
        raise NotImplementedError(f"{what}: {msg}")
    if rc == -5:
        raise RuntimeError(f"{what}: {msg} (workspace)")
    raise RuntimeError(f"{what} failed (rc={rc}): {msg}")
