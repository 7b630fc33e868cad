"""CPU tests of the drop-in boundary: libhqq_hip.so loads and exports exactly what include/hqq_hip.h declares."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "hqq_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hqq_hip_\w+)\s*\(", src)))


def test_header_and_binding_agree():
    from hqq_amd import _C
    assert _declared() == sorted(_C.SYMBOLS)


def test_library_loads_and_exports_every_symbol():
    from hqq_amd import _C
    if not os.path.exists(_C.LIB_PATH):
        _C.build()
    L = _C.lib()
    for name in _declared():
        assert hasattr(L, name), name
    assert L.hqq_hip_abi_version() == _C.ABI_VERSION


def test_argument_errors_do_not_need_a_gpu():
    from hqq_amd import _C
    L = _C.lib()
    assert L.hqq_hip_packed_rows(4, 7) < 0 and L.hqq_hip_packed_rows(3, 7) == 1 and L.hqq_hip_packed_rows(5, 8) < 0
    assert L.hqq_hip_quantize_workspace_bytes(1024, 64, 20) > 0
    assert L.hqq_hip_quantize_workspace_bytes(1000, 64, 20) == 0
    # nbits=5 has no container of its own (the reference stores it in 8 bits): reported, never silently computed elsewhere
    rc = L.hqq_hip_gemv(5, 16, 16, 16, 16, None, 16, 1, 64, 64, 64, 1, 0, None, 0, None)
    assert rc == -4 and b"not covered" in L.hqq_hip_last_error()
    # unknown option bits are an argument error, not ignored
    assert L.hqq_hip_gemv(4, 16, 16, 16, 16, None, 16, 1, 64, 64, 64, 1, 1 << 15, None, 0, None) == -2
    # the decode kernel addresses a layer with 32-bit byte offsets: 2^22 packed rows x 1024 bytes is one byte too many, and says so
    assert L.hqq_hip_gemv(4, 16, 16, 16, 16, None, 16, 1, 1 << 23, 1024, 64, 1, 0, None, 0, None) == -2
    assert b"size overflow" in L.hqq_hip_last_error()
    # workspace sizes are pure host arithmetic: a bs=1 launch needs none; a small layer at 32 rows splits K and does
    import ctypes
    N1 = (ctypes.c_int64 * 1)(4096)
    assert L.hqq_hip_gemv_workspace_bytes(4, 1, N1, 1, 4096, 64, 1, 0) == 0
    assert L.hqq_hip_gemv_workspace_bytes(4, 1, N1, 32, 4096, 64, 1, 0) > 256 * 1024
    # the fused GEMM's plan is host arithmetic too: 128 rows of a 4096 x 4096 layer split K (partial tiles in the workspace), 8192 rows do not;
    # the routing hint: fused up to 640 rows (1024 when the plan fills the chip in one round) where the pipelined kernel applies (group_size 64)
    assert L.hqq_hip_gemm_workspace_bytes(4, 128, 4096, 4096, 64, 1, 0) > 256 * 1024
    assert L.hqq_hip_gemm_workspace_bytes(4, 8192, 4096, 4096, 64, 1, 0) == 0
    assert L.hqq_hip_forward_workspace_bytes(4, 1, 4096, 4096, 64, 1, 0) == 0
    assert L.hqq_hip_forward_prefers_fused(4, 512, 4096, 4096, 64, 1) == 1 and L.hqq_hip_forward_prefers_fused(4, 512, 4096, 4096, 64, 2) == 1
    assert L.hqq_hip_forward_prefers_fused(4, 4096, 4096, 4096, 64, 1) == 0 and L.hqq_hip_forward_prefers_fused(4, 128, 4096, 4096, 32, 1) == 0
    assert L.hqq_hip_forward_prefers_fused(4, 32, 4096, 4096, 64, 1) == 1
    # the peer-memory exchange validates its arguments before it launches anything
    VP1, VP2 = (ctypes.c_void_p * 1)(16), (ctypes.c_void_p * 2)(16, 16)
    st = ctypes.c_void_p(16)
    assert L.hqq_hip_exchange(1, VP1, (ctypes.c_int64 * 1)(512), 1, 4, 1, 17, 0, VP2, VP2, st, 0, None) == -2     # more ranks than HQQ_EXCHANGE_MAX_RANKS
    assert L.hqq_hip_exchange(1, VP1, (ctypes.c_int64 * 1)(511), 1, 4, 1, 2, 0, VP2, VP2, st, 0, None) == -2      # 511 columns do not split into two slab runs
    assert L.hqq_hip_exchange(1, VP1, (ctypes.c_int64 * 1)(512), 1, 4, 0, 2, 0, VP2, VP2, st, 0, None) == -3      # fp32 activations
    assert L.hqq_hip_exchange(1, VP1, (ctypes.c_int64 * 1)(512), 1, 4, 1, 2, 2, VP2, VP2, st, 0, None) == -2      # rank 2 of 2
    assert L.hqq_hip_exchange(1, VP1, (ctypes.c_int64 * 1)(512), 65, 4, 1, 2, 0, VP2, VP2, st, 0, None) == -2     # more rows than HQQ_EXCHANGE_MAX_ROWS
    # the per-token ends of the decode step (ABI 7) validate before they launch
    p16 = ctypes.c_void_p(16)
    assert L.hqq_hip_token_prologue(p16, p16, p16, 100, 4100, None, None, 1, 0, p16, None, None, None, 1, None) == -2     # H not a multiple of 8
    assert L.hqq_hip_token_prologue(p16, p16, p16, 100, 4096, p16, None, 64, 128, p16, p16, p16, None, 1, None) == -2     # a cos table without its sin table
    assert L.hqq_hip_token_prologue(p16, p16, p16, 100, 4096, None, None, 1, 0, p16, None, None, None, 0, None) == -4     # fp32
    assert L.hqq_hip_token_prologue(p16, p16, ctypes.c_void_p(24), 100, 4096, None, None, 1, 0, p16, None, None, None, 1, None) == -6   # misaligned embedding
    assert L.hqq_hip_argmax_advance(None, 100, 1, p16, None, None, None) == -2 and L.hqq_hip_argmax_advance(p16, 0, 1, p16, None, None, None) == -2
    assert L.hqq_hip_argmax_advance(p16, 100, 0, p16, None, None, None) == -4


def test_the_library_owns_no_device_memory_and_reads_no_environment():
    """boundary rule of include/hqq_hip.h: the caller owns every buffer, nothing is configured through the environment"""
    import subprocess
    from hqq_amd import _C
    if not os.path.exists(_C.LIB_PATH):
        _C.build()
    syms = subprocess.run(["nm", "-D", "--undefined-only", _C.LIB_PATH], capture_output=True, text=True).stdout
    for banned in (r"hipMalloc", r"hipFree", r"hipDeviceSynchronize", r"hipStreamSynchronize", r"getenv", r"hipMemset@", r"hipMemcpy"):
        assert not re.search(r"\b" + banned, syms), banned   # (hipMemsetAsync — stream-ordered — clears the meta check's counter)


def test_ops_refuse_cpu_tensors():
    import torch
    from hqq_amd import ops
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.pack(4, torch.zeros(8, 8, dtype=torch.uint8))


def test_no_single_rounding_fp16_products_in_the_norm_kernels():
    """`T(float(h) * rinv)` must round twice (fp32 product, then fp16), as HF's LlamaRMSNorm does.  hipcc folds the pair into v_fma_mixlo_f16 — ONE rounding
    (tools/r6/mixlo_probe.hip: 996 of 16.7 M products differ) — unless the product is made opaque (block_math.h El::r_prod).  The shipped code objects of the
    kernels that normalise (block.hip, gemv_block.hip) must not contain the fused form."""
    import glob
    import os
    import shutil
    import subprocess
    import tempfile
    from hqq_amd import _C
    bundler, objdump = "/opt/rocm/lib/llvm/bin/clang-offload-bundler", "/opt/rocm/lib/llvm/bin/llvm-objdump"
    objs = [o for o in glob.glob(os.path.join(_C.CSRC, "build", "*.o")) if os.path.basename(o).startswith(("block.", "gemv_block_"))]
    if not (objs and os.path.exists(bundler) and os.path.exists(objdump) and shutil.which("objcopy")):
        import pytest
        pytest.skip("needs the built objects and the ROCm LLVM tools")
    for o in objs:
        with tempfile.TemporaryDirectory() as d:
            subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", o, f"{d}/fb.bin"])
            subprocess.check_call([bundler, "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={d}/fb.bin", f"--output={d}/co.co", "--unbundle"])
            dis = subprocess.run([objdump, "-d", f"{d}/co.co"], capture_output=True, text=True, check=True).stdout
        assert "v_fma_mixlo_f16" not in dis and "v_fma_mixhi_f16" not in dis, os.path.basename(o)
