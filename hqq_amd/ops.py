"""Tensor-level wrappers over the C ABI (include/hqq_hip.h).  PyTorch only supplies device memory and the
current HIP stream; every function below enqueues hand-written gfx950 kernels from libhqq_hip.so.

No fallbacks: tensors must live on a ROCm device ("cuda" in torch), and a missing library or an
unsupported configuration raises (RuntimeError / NotImplementedError) instead of silently running
eager PyTorch.
"""
from __future__ import annotations

import torch
from torch import Tensor

from . import _C

F32, F16, BF16, U8 = 0, 1, 2, 3
_DT = {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16, torch.uint8: U8}
PER = {8: 1, 4: 2, 2: 4, 1: 8, 3: 10}
# Quantizer.bit_to_packing (hqq/core/quantize.py:40-49): container width per nbits
PACK_BITS = {8: 8, 6: 8, 5: 8, 4: 4, 3: 3, 2: 2, 1.58: 2, 1: 1}
GEMV_MAX_M = 16
SKINNY_MAX_M = 64   # HQQ_GEMV_MAX_M_SKINNY: fp16 / bf16, 8-/4-/2-bit, group_size 64, K % 256 == 0, K >= 512
GEMV_EXACT, GEMV_FACTORED = 0, 1
GEMV_MAX_GROUP = 4
# per-call option bits of the C ABI (include/hqq_hip.h HQQ_OPT_*)
OPT_FACTORED, OPT_META_SCALABLE, OPT_GEMV3_ROWWISE, OPT_GEMV3_SLABS, OPT_GEMM_REGTILE, OPT_GEMM_CLASSIC, OPT_GEMM_NARROW, OPT_GEMM_WIDE, OPT_GEMM_NOHYBRID, OPT_SKINNY_WIDE = 1, 2, 4, 8, 16, 32, 64, 128, 256, 512
OPT_W3S = 1024   # nbits = 3: W_q is the 3-bit stream layout of w3s_pack(), not the reference container
OPT_BATCH_SPLITK = 2048   # LAB builds only (tools/lab_kwave/build.sh): force the split-K kernel where the quarantined no-split kernel would serve


def OPT_SKINNY_KS(n: int) -> int:
    return int(n) << 24


# Default arithmetic of the decode wrappers below when a call passes no `opts` — a convenience of THIS module (tools, tests,
# bench); the library itself holds no mode.
_default_opts = 0


def is_available() -> bool:
    """True when libhqq_hip.so loads and a ROCm GPU is visible."""
    try:
        _C.lib()
    except (RuntimeError, OSError):
        return False
    return torch.cuda.is_available()


def _dt(t: torch.dtype) -> int:
    try:
        return _DT[t]
    except KeyError:
        raise TypeError(f"hqq_amd: dtype {t} not supported by the HIP kernels") from None


def _dev(*ts: Tensor) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("hqq_amd: HIP kernels need tensors on the GPU (device='cuda'); there is no CPU path")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def set_gemv_mode(mode: int) -> None:
    """GEMV_EXACT (default): reference-identical weights on the MFMA path; GEMV_FACTORED: fp32-factored dot2 path.  Applies to raw
    ops.* calls that pass no `opts` and to every layer forward (HQQLinear with HQQBackend.HIP, HQQLinearHIP, grouped projections),
    which combine it with their own meta-dependent bits through layer_opts()."""
    global _default_opts
    if mode not in (GEMV_EXACT, GEMV_FACTORED):
        raise ValueError(f"hqq_amd: unknown gemv mode {mode}")
    _default_opts = OPT_FACTORED if mode == GEMV_FACTORED else 0


def get_gemv_mode() -> int:
    return GEMV_FACTORED if (_default_opts & OPT_FACTORED) else GEMV_EXACT


def _opts(opts) -> int:
    return _default_opts if opts is None else int(opts)


def layer_opts(meta_opts: int) -> int:
    """Option bits for a LAYER's forward: its own meta-dependent bits (0 / OPT_META_SCALABLE), unless set_gemv_mode(GEMV_FACTORED) is in
    force — then the factored arithmetic for every layer (the three-op bit belongs to the exact rebuild and is dropped)."""
    if _default_opts & OPT_FACTORED:
        return OPT_FACTORED | (int(meta_opts) & OPT_W3S)   # (the layout bit describes the tensor, not the arithmetic: it always travels)
    return int(meta_opts)


# ---- caller-owned workspace of the split-K / slab-sharing decode launches (include/hqq_hip.h "Workspace") --------------------
# One zero-initialised buffer per device, sized for the largest launch seen so far.  Growing allocates a NEW buffer and keeps
# the old ones alive: a hipGraph captured earlier has the old address baked in and must stay valid (never free what a graph
# may reference).  Growth cannot happen inside stream capture (the new buffer would belong to the graph's private pool).
_ws_cur: dict = {}
_ws_retired: list = []
_WS_MIN = 8 << 20


_ws_last_stream: dict = {}


def _ws_serialise(key: int) -> None:
    """The device's workspace is ONE buffer (arrival counters + parked partial sums) and the C ABI forbids sharing it between calls that
    may run concurrently: a call from another stream than the previous workspace user first waits for everything that stream has
    enqueued.  (Outside stream capture; inside a capture the launches of one capture are ordered by the capture itself unless the
    caller forks streams — then give each branch its own buffer through the C ABI.)"""
    if torch.cuda.is_current_stream_capturing():
        return   # nothing runs during a capture, and a capture stream must not become the "last user" an eager call later waits on
    cur = torch.cuda.current_stream(key)
    last = _ws_last_stream.get(key)
    if last is not None and last != cur:
        cur.wait_stream(last)
    _ws_last_stream[key] = cur


def release_retired_workspaces() -> int:
    """Free the workspace buffers that growth retired.  Only when no captured hipGraph still replays launches that were captured with
    them (the caller knows; the library cannot).  Returns the bytes released."""
    n = sum(t.numel() for t in _ws_retired)
    _ws_retired.clear()
    return n


def reserve_workspace(device, nbytes: int) -> Tensor:
    """make sure the device's decode workspace holds `nbytes`; call it before capturing a graph whose launches need one"""
    dev = torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    cur = _ws_cur.get(key)
    _ws_serialise(key)
    if cur is not None and cur.numel() >= nbytes:
        return cur
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError(f"hqq_amd: this launch needs {nbytes} bytes of decode workspace, more than was reserved before stream capture; "
                           "run the step once eagerly (or call hqq_amd.ops.reserve_workspace) before capturing")
    new = torch.zeros(max(int(nbytes), _WS_MIN), dtype=torch.uint8, device=torch.device("cuda", key))
    if cur is not None:
        _ws_retired.append(cur)
    _ws_cur[key] = new
    return new


def _workspace(x: Tensor, nbits, Ns, M, K, group_size, opts):
    import ctypes
    n = len(Ns)
    need = int(_C.lib().hqq_hip_gemv_workspace_bytes(int(nbits), n, (ctypes.c_int64 * n)(*[int(v) for v in Ns]), int(M), int(K), int(group_size),
                                                     _dt(x.dtype), int(opts)))
    if not need:
        return None, 0
    ws = reserve_workspace(x.device, need)
    return ws.data_ptr(), ws.numel()


def meta_scalable(scale: Tensor, zero: Tensor, N: int, K: int, group_size: int, nbits: int) -> bool:
    """True when every (zero, scale) pair of the layer can take the three-op exact weight rebuild (hqq_hip_meta_check == 0 failing
    groups): pass OPT_META_SCALABLE for it then.  Synchronises (one 4-byte read-back): call it when a layer is prepared, not per forward."""
    _dev(scale, zero)
    if scale.dtype != torch.float16 or zero.dtype != torch.float16 or nbits not in (8, 4, 3, 2, 1) or (nbits != 3 and N % PER[nbits]):
        return False
    cnt = torch.empty(1, dtype=torch.int32, device=scale.device)
    with torch.cuda.device(scale.device):
        rc = _C.lib().hqq_hip_meta_check(int(nbits), _p(scale.contiguous()), _p(zero.contiguous()), int(N), int(K), int(group_size), F16, _p(cnt), _stream())
    _C.check(rc, "hqq_hip_meta_check")
    return int(cnt.item()) == 0


def w3s_covers(N: int, K: int, group_size) -> bool:
    """layers the 3-bit stream layout (csrc/w3s.h) can hold: group_size 64, an even number of output rows"""
    return group_size == 64 and N % 2 == 0 and K % 64 == 0 and N > 0 and K > 0


def w3s_pack(W_q: Tensor, N: int, K: int) -> Tensor:
    """The reference's 3-bit container ([ceil(N K / 640), 64] int32, BitPack.pack_3bit_32) -> the stream layout [N/2, K/16 * 3] int32 that the
    decode / GEMM kernels read with OPT_W3S (hqq_hip_w3s_pack).  What HQQLinearHIP does once when a 3-bit layer is patched — the
    re-layout step of the reference's optimised backends (hqq/backends/torchao.py:202-241, marlin.py:74-123)."""
    _dev(W_q)
    if W_q.dtype != torch.int32 or W_q.numel() != ((N * K // 64 + 9) // 10) * 64:
        raise ValueError(f"hqq_amd: w3s_pack takes the [ceil(N K / 640), 64] int32 container of a {N} x {K} layer")
    out = torch.empty((N // 2, K // 16 * 3), dtype=torch.int32, device=W_q.device)
    with torch.cuda.device(W_q.device):
        rc = _C.lib().hqq_hip_w3s_pack(_p(W_q.contiguous()), _p(out), int(N), int(K), _stream())
    _C.check(rc, "hqq_hip_w3s_pack")
    return out


def w3s_unpack(w3s: Tensor, N: int, K: int) -> Tensor:
    """stream layout -> the reference's container, bit for bit (zero padding rows included): state_dict() / dequantize() of a patched layer"""
    _dev(w3s)
    if w3s.dtype != torch.int32 or w3s.numel() != (N // 2) * (K // 16) * 3:
        raise ValueError(f"hqq_amd: w3s_unpack takes the [N/2, K/16 * 3] int32 stream layout of a {N} x {K} layer")
    out = torch.empty(((N * K // 64 + 9) // 10, 64), dtype=torch.int32, device=w3s.device)
    with torch.cuda.device(w3s.device):
        rc = _C.lib().hqq_hip_w3s_unpack(_p(w3s.contiguous()), _p(out), int(N), int(K), _stream())
    _C.check(rc, "hqq_hip_w3s_unpack")
    return out


def w3s_meta_scalable(scale: Tensor, zero: Tensor, N: int, K: int) -> bool:
    """meta_scalable() for a layer in the 3-bit stream layout (hqq_hip_w3s_meta_check): OPT_META_SCALABLE may accompany OPT_W3S then.  Synchronises."""
    _dev(scale, zero)
    if scale.dtype != torch.float16 or zero.dtype != torch.float16:
        return False
    cnt = torch.empty(1, dtype=torch.int32, device=scale.device)
    with torch.cuda.device(scale.device):
        rc = _C.lib().hqq_hip_w3s_meta_check(_p(scale.contiguous()), _p(zero.contiguous()), int(N), int(K), _p(cnt), _stream())
    _C.check(rc, "hqq_hip_w3s_meta_check")
    return int(cnt.item()) == 0


def packed_rows(nbits: int, rows: int) -> int:
    r = _C.lib().hqq_hip_packed_rows(int(nbits), int(rows))
    if r < 0:
        raise ValueError(f"hqq_amd: {rows} rows cannot be packed at {nbits} bits (rows must divide by {PER.get(nbits)})")
    return int(r)


def pack(nbits: int, W_q: Tensor) -> Tensor:
    """BitPack.pack_{8,4,2,1}bit_u8 / pack_3bit_32 (hqq/core/bitpack.py).  W_q: [rows, cols] integer levels
    (uint8, or float32 as produced by the reference solver)."""
    _dev(W_q)
    if W_q.dtype not in (torch.uint8, torch.float32):
        W_q = W_q.to(torch.uint8)
    W_q = W_q.contiguous()
    rows, cols = W_q.shape
    out = torch.empty((packed_rows(nbits, rows), cols), dtype=torch.int32 if nbits == 3 else torch.uint8, device=W_q.device)
    with torch.cuda.device(W_q.device):
        rc = _C.lib().hqq_hip_pack(nbits, _p(W_q), _dt(W_q.dtype), rows, cols, _p(out), _stream())
    _C.check(rc, "hqq_hip_pack")
    return out


def unpack(nbits: int, W_q: Tensor, dtype: torch.dtype = torch.uint8) -> Tensor:
    """BitPack.unpack_*(W_q, dtype) — returns [per*packed_rows, cols] (3-bit: including the padding rows)."""
    _dev(W_q)
    W_q = W_q.contiguous()
    prow, cols = W_q.shape
    out = torch.empty((PER[nbits] * prow, cols), dtype=dtype, device=W_q.device)
    with torch.cuda.device(W_q.device):
        rc = _C.lib().hqq_hip_unpack(nbits, _p(W_q), prow, cols, _p(out), _dt(dtype), _stream())
    _C.check(rc, "hqq_hip_unpack")
    return out


def dequantize(W_q: Tensor, scale: Tensor, zero: Tensor, N: int, K: int, group_size: int, nbits: int, axis: int = 1) -> Tensor:
    """Quantizer.dequantize / hqq_aten.dequantize: [N,K] in scale.dtype, bit-identical to the reference."""
    _dev(W_q, scale, zero)
    if scale.dtype != zero.dtype:
        raise TypeError("hqq_amd: scale and zero must share the compute dtype")
    groups = (N * K) // int(group_size)
    if scale.numel() != groups or zero.numel() != groups:   # the kernel reads N * K / group_size constants through raw pointers
        raise ValueError(f"hqq_amd: dequantize needs {groups} scale / zero values (N * K / group_size), got {scale.numel()} / {zero.numel()}")
    out = torch.empty((N, K), dtype=scale.dtype, device=W_q.device)
    with torch.cuda.device(W_q.device):
        rc = _C.lib().hqq_hip_dequantize(nbits, _p(W_q.contiguous()), _p(scale.contiguous()), _p(zero.contiguous()), _p(out),
                                         N, K, group_size, axis, _dt(scale.dtype), _stream())
    _C.check(rc, "hqq_hip_dequantize")
    return out


def _fwd(fn_name: str, x: Tensor, W_q: Tensor, scale: Tensor, zero: Tensor, bias, N: int, K: int, group_size: int, nbits: int,
         out: Tensor | None = None, opts=None) -> Tensor:
    _dev(x, W_q, scale, zero, bias)
    if x.dtype != scale.dtype or zero.dtype != scale.dtype or (bias is not None and bias.dtype != scale.dtype):
        raise TypeError("hqq_amd: x / scale / zero / bias must share the compute dtype")
    if x.shape[-1] != K:
        raise ValueError(f"hqq_amd: x has {x.shape[-1]} features, layer expects {K}")
    x2 = x.reshape(-1, K)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    M = x2.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    if M > 0:
        o = _opts(opts)
        with torch.cuda.device(x.device):
            if fn_name == "hqq_hip_gemv":
                ws, ws_bytes = _workspace(x, nbits, [N], M, K, group_size, o)
            else:   # hqq_hip_gemm / hqq_hip_forward
                need = int(getattr(_C.lib(), fn_name + "_workspace_bytes")(int(nbits), M, N, K, group_size, _dt(x.dtype), o))
                ws = reserve_workspace(x.device, need) if need else None
                ws, ws_bytes = (ws.data_ptr(), ws.numel()) if need else (None, 0)
            rc = getattr(_C.lib(), fn_name)(nbits, _p(x2), _p(W_q), _p(scale), _p(zero), _p(bias), _p(out), M, N, K, group_size,
                                            _dt(x.dtype), o, ws, ws_bytes, _stream())
        _C.check(rc, fn_name)
    return out.reshape(*x.shape[:-1], N)


def gemv(x, W_q, scale, zero, bias, N, K, group_size, nbits, out=None, opts=None) -> Tensor:
    """fused unpack->dequant->GEMV for decode-sized batches: 1 <= M <= 16 (3-bit, and bf16 outside the skinny-GEMM kernel: <= 4; FACTORED mode: <= 8); up to SKINNY_MAX_M where skinny_covers()."""
    return _fwd("hqq_hip_gemv", x, W_q, scale, zero, bias, N, K, group_size, nbits, out, opts)


def gemv_grouped(x: Tensor, layers, K: int, group_size: int, nbits: int, outs=None, opts=None):
    """Horizontal fusion: one launch for up to GEMV_MAX_GROUP layers that consume the same x (q/k/v, gate/up, ...).
    layers: sequence of (W_q, scale, zero, bias_or_None, N).  Returns the list of outputs [*, N_i]."""
    import ctypes
    n = len(layers)
    if not 1 <= n <= GEMV_MAX_GROUP:
        raise ValueError(f"hqq_amd: a GEMV group holds 1..{GEMV_MAX_GROUP} layers, got {n}")
    if x.shape[-1] != K:
        raise ValueError(f"hqq_amd: x has {x.shape[-1]} features, layers expect {K}")
    x2 = x.reshape(-1, K)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    M = x2.shape[0]
    for (W_q, s, z, b, N) in layers:
        _dev(x, W_q, s, z, b)
        if x.dtype != s.dtype or z.dtype != s.dtype or (b is not None and b.dtype != s.dtype):
            raise TypeError("hqq_amd: x / scale / zero / bias must share the compute dtype")
    if outs is None:
        outs = [torch.empty((M, L[4]), dtype=x.dtype, device=x.device) for L in layers]
    if M > 0:
        VP = ctypes.c_void_p * n
        has_bias = any(L[3] is not None for L in layers)
        o = _opts(opts)
        with torch.cuda.device(x.device):
            ws, ws_bytes = _workspace(x, nbits, [L[4] for L in layers], M, K, group_size, o)
            rc = _C.lib().hqq_hip_gemv_grouped(
                nbits, n, _p(x2), VP(*[_p(L[0]) for L in layers]), VP(*[_p(L[1]) for L in layers]), VP(*[_p(L[2]) for L in layers]),
                VP(*[_p(L[3]) for L in layers]) if has_bias else None, VP(*[_p(o_) for o_ in outs]),
                (ctypes.c_int64 * n)(*[int(L[4]) for L in layers]), M, K, group_size, _dt(x.dtype), o, ws, ws_bytes, _stream())
        _C.check(rc, "hqq_hip_gemv_grouped")
    return [o.reshape(*x.shape[:-1], L[4]) for o, L in zip(outs, layers)]


FUSED_GEMM_MAX_M = 2560   # gemm_pipe_wins (csrc/gemm_pipe.hip): beyond, dequantise + the dense GEMM is the faster route


def gemm_grouped_covers(dtype, layers_N, M: int, K: int, group_size, nbits: int, opts=None) -> bool:
    """True when hqq_hip_gemm_grouped serves this group: every layer on the pipelined fused GEMM (fp16 / bf16, 8 / 4 / 2 bit or the 3-bit stream layout, group_size 64, K % 128 == 0)"""
    import ctypes
    n = len(layers_N)
    if dtype not in _DT or not 1 <= n <= GEMV_MAX_GROUP or M < 1 or not group_size:
        return False
    return bool(_C.lib().hqq_hip_gemm_grouped_covers(int(nbits), n, (ctypes.c_int64 * n)(*[int(v) for v in layers_N]), int(M), int(K), int(group_size), _dt(dtype), _opts(opts)))


def gemm_grouped(x: Tensor, layers, K: int, group_size: int, nbits: int, outs=None, opts=None):
    """Horizontal fusion beyond the decode rows: ONE launch of the pipelined fused GEMM (+ one split-K reduce) for up to GEMV_MAX_GROUP layers that consume
    the same x (q / k / v, gate / up) — a decoder block at 65..2560 rows is 4 launches instead of 7.  layers: sequence of (W_q, scale, zero, bias_or_None, N).
    Returns the list of outputs [*, N_i].  The K split is chosen for the group's total width: a row can differ in the last bit from the layer launched alone."""
    import ctypes
    n = len(layers)
    if not 1 <= n <= GEMV_MAX_GROUP:
        raise ValueError(f"hqq_amd: a GEMM group holds 1..{GEMV_MAX_GROUP} layers, got {n}")
    if x.shape[-1] != K:
        raise ValueError(f"hqq_amd: x has {x.shape[-1]} features, layers expect {K}")
    x2 = x.reshape(-1, K)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    M = x2.shape[0]
    for (W_q, s, z, b, N) in layers:
        _dev(x, W_q, s, z, b)
        if x.dtype != s.dtype or z.dtype != s.dtype or (b is not None and b.dtype != s.dtype):
            raise TypeError("hqq_amd: x / scale / zero / bias must share the compute dtype")
    if outs is None:
        outs = [torch.empty((M, L[4]), dtype=x.dtype, device=x.device) for L in layers]
    if M > 0:
        VP = ctypes.c_void_p * n
        has_bias = any(L[3] is not None for L in layers)
        o = _opts(opts)
        Ns = (ctypes.c_int64 * n)(*[int(L[4]) for L in layers])
        with torch.cuda.device(x.device):
            need = int(_C.lib().hqq_hip_gemm_grouped_workspace_bytes(int(nbits), n, Ns, int(M), int(K), int(group_size), _dt(x.dtype), o))
            ws, ws_bytes = (None, 0)
            if need:
                w = reserve_workspace(x.device, need)
                ws, ws_bytes = w.data_ptr(), w.numel()
            rc = _C.lib().hqq_hip_gemm_grouped(
                nbits, n, _p(x2), VP(*[_p(L[0]) for L in layers]), VP(*[_p(L[1]) for L in layers]), VP(*[_p(L[2]) for L in layers]),
                VP(*[_p(L[3]) for L in layers]) if has_bias else None, VP(*[_p(o_) for o_ in outs]), Ns, M, K, group_size, _dt(x.dtype), o, ws, ws_bytes, _stream())
        _C.check(rc, "hqq_hip_gemm_grouped")
    return [o_.reshape(*x.shape[:-1], L[4]) for o_, L in zip(outs, layers)]


BLOCK_NORM, BLOCK_RESID, BLOCK_SILU = 1, 2, 4   # HQQ_BLOCK_* (include/hqq_hip.h)


def block_covers(dtype, K: int, group_size, nbits: int, w3s: bool, norm: bool) -> bool:
    """what hqq_hip_gemv_block serves: one activation row, fp16 / bf16, 4- / 2-bit or the 3-bit stream layout, group_size 64; K <= 8192 with the RMSNorm prologue"""
    return (dtype in (torch.float16, torch.bfloat16) and group_size == 64 and K % 64 == 0 and (nbits in (4, 2) or (nbits == 3 and w3s))
            and (not norm or K <= 8192) and K * 2 <= 144 * 1024 - 4096)


BLOCK_ROPE = 8


def gemv_block(x: Tensor, norm_weight, eps: float, layers, K: int, group_size: int, nbits: int, outs, flags: int, opts=None, rope=None):
    """The decoder block's launches with the glue folded in (csrc/gemv_block.hip; include/hqq_hip.h hqq_hip_gemv_block), ONE activation row:
      BLOCK_NORM               x = the residual stream; layers (W_q, scale, zero, N) like gemv_grouped's; outs[i] [1, N_i]
      BLOCK_NORM | BLOCK_SILU  ONE layer from pair_layers(gate, up): outs[0] [1, N / 2] = silu(gate) * up
      BLOCK_RESID              ONE layer; outs[0] is the residual stream, updated in place: h += layer(x)
      BLOCK_NORM | BLOCK_ROPE  q | k | v with q and k from rotary_pair_layout(): outs = [q_out [n_heads, hd], k_cache, v_cache [n_kv, L, hd]];
                               rope = (cos [hd], sin [hd], pos [1] int64 on the device, head_dim, cache_len): rope_cache() in the launch's epilogue"""
    import ctypes
    n = len(layers)
    rp = None
    if flags & BLOCK_ROPE:
        class _Rope(ctypes.Structure):
            _fields_ = [("cos", ctypes.c_void_p), ("sin", ctypes.c_void_p), ("pos", ctypes.c_void_p), ("head_dim", ctypes.c_int64), ("cache_len", ctypes.c_int64)]
        cos, sin, pos, hd, L = rope
        _dev(cos, sin, pos)
        if cos.numel() != hd or sin.numel() != hd or pos.dtype != torch.int64 or cos.dtype != x.dtype or sin.dtype != x.dtype:
            raise ValueError("hqq_amd: rope = (cos [head_dim], sin [head_dim] in the compute dtype, pos int64 on the device, head_dim, cache_len)")
        if outs[1].shape[-1] != hd or outs[1].shape[-2] != L or outs[1].shape != outs[2].shape or not (outs[1].is_contiguous() and outs[2].is_contiguous()):
            raise ValueError("hqq_amd: the caches must be contiguous [n_kv_heads, cache_len, head_dim] tensors")
        rp = ctypes.byref(_Rope(_p(cos), _p(sin), _p(pos), int(hd), int(L)))
    _dev(x, norm_weight, *[t for L in layers for t in L[:3]], *outs)
    if x.numel() != K:
        raise ValueError(f"hqq_amd: gemv_block serves one activation row of {K} features, got {tuple(x.shape)}")
    VP = ctypes.c_void_p * n
    o = _opts(opts)
    with torch.cuda.device(x.device):
        rc = _C.lib().hqq_hip_gemv_block(int(nbits), n, _p(x), _p(norm_weight), float(eps), VP(*[_p(L[0]) for L in layers]), VP(*[_p(L[1]) for L in layers]),
                                         VP(*[_p(L[2]) for L in layers]), VP(*[_p(t) for t in outs]), (ctypes.c_int64 * n)(*[int(L[-1]) for L in layers]),
                                         int(K), int(group_size), _dt(x.dtype), o, int(flags), rp, _stream())
    _C.check(rc, "hqq_hip_gemv_block")
    return outs


def pair_layers(gate, up, K: int, group_size: int, nbits: int, w3s: bool = False):
    """The PAIRED layout of two layers of equal shape that read the same input (LlamaMLP's gate_proj / up_proj): the level matrix of `gate` on top of
    the level matrix of `up`, packed as ONE layer of 2 N rows — BitPack's row slabs then put gate row n and up row n into the same packed row
    (4-bit: byte (n, k) = gate level << 4 | up level), which is what lets hqq_hip_gemv_block's epilogue form silu(gate[n]) * up[n] inside one wave.
    gate / up: (W_q, scale, zero, N) as the layers hold them (3-bit: the stream layout when w3s).  Returns (W_q, scale, zero, 2 N); the originals are
    untouched (state_dict() and the prefill path keep using them).  Levels, scale and zero are the layers' own: the same weights bit for bit."""
    (Wg, sg, zg, N), (Wu, su, zu, Nu) = gate, up
    if N != Nu or sg.dtype != su.dtype:
        raise ValueError("hqq_amd: pair_layers needs two layers of the same shape and compute dtype")
    if w3s:
        Wg, Wu = w3s_unpack(Wg, N, K), w3s_unpack(Wu, N, K)
    R = N * K // group_size
    Ug = unpack(nbits, Wg)[:R]          # level matrix [N K / gs, gs], row (n, g)
    Uu = unpack(nbits, Wu)[:R]
    W = pack(nbits, torch.cat([Ug, Uu], dim=0))
    if w3s:
        W = w3s_pack(W, 2 * N, K)
    return W, torch.cat([sg.reshape(-1), su.reshape(-1)]).contiguous(), torch.cat([zg.reshape(-1), zu.reshape(-1)]).contiguous(), 2 * N


def merge_layers(layers, K: int, group_size: int, nbits: int, w3s: bool = False):
    """ONE layer whose rows are the rows of `layers` in order (q_proj | k_proj | v_proj, gate_proj | up_proj: layers that read the same input): the
    level matrices stacked and packed again, scale / zero concatenated — the same levels and constants bit for bit, so y of the merged layer is the
    concatenation of the layers' outputs.  What it buys: one launch over sum(N) rows instead of one per layer (at 65..2560 activation rows, where
    hqq_hip_gemv_grouped does not reach: a 12288 x 4096 launch at 128 rows takes 29-31 us against 3 x 19-20).
    layers: (W_q, scale, zero, N) as the layers hold them (3-bit: the stream layout when w3s).  Returns (W_q, scale, zero, sum N); the originals are
    untouched.  (The reference has no such helper; vLLM's merged q|k|v / gate|up modules are the same idea: hqq/utils/vllm.py.)"""
    if not layers:
        raise ValueError("hqq_amd: merge_layers needs at least one layer")
    dt = layers[0][1].dtype
    U = []
    for W, s, z, N in layers:
        if s.dtype != dt or z.dtype != dt:
            raise ValueError("hqq_amd: merge_layers needs one compute dtype")
        if (N * K) % group_size or s.numel() != N * K // group_size or z.numel() != s.numel():
            raise ValueError("hqq_amd: merge_layers takes channel-wise layers quantised along axis 1 (one scale / zero per group of a row)")
        if w3s:
            W = w3s_unpack(W, N, K)
        U.append(unpack(nbits, W)[:N * K // group_size])   # level matrix [N K / gs, gs], row (n, g)
    Nt = sum(int(l[3]) for l in layers)
    Wm = pack(nbits, torch.cat(U, dim=0))
    if w3s:
        Wm = w3s_pack(Wm, Nt, K)
    return (Wm, torch.cat([l[1].reshape(-1) for l in layers]).contiguous(), torch.cat([l[2].reshape(-1) for l in layers]).contiguous(), Nt)


def rotary_pair_layout(layer, K: int, group_size: int, nbits: int, head_dim: int, w3s: bool = False):
    """The ROTARY-PAIRED row order of a q_proj / k_proj layer: element i < head_dim / 2 of head h becomes row h head_dim / 2 + i, its rotary partner
    i + head_dim / 2 row N / 2 + h head_dim / 2 + i — so that BitPack's row slabs hold both in ONE packed row and hqq_hip_gemv_block's epilogue can apply
    apply_rotary_pos_emb inside the wave that finishes the row (its outputs are written back in the natural order).  layer: (W_q, scale, zero, N) as the
    layer holds it; returns a permuted copy (the original is untouched: state_dict() and the prefill path keep using it).  Same levels, scale, zero per row."""
    W, s, z, N = layer
    if N % head_dim or head_dim % 2:
        raise ValueError("hqq_amd: rotary_pair_layout needs whole heads of an even size")
    G = K // group_size
    if w3s:
        W = w3s_unpack(W, N, K)
    rows = torch.arange(N, device=s.device).view(N // head_dim, head_dim)
    perm = torch.cat([rows[:, :head_dim // 2].reshape(-1), rows[:, head_dim // 2:].reshape(-1)])
    U = unpack(nbits, W)[:N * G].reshape(N, G * group_size).index_select(0, perm).reshape(N * G, group_size).contiguous()
    Wp = pack(nbits, U)
    if w3s:
        Wp = w3s_pack(Wp, N, K)
    return Wp, s.reshape(N, G).index_select(0, perm).reshape(-1).contiguous(), z.reshape(N, G).index_select(0, perm).reshape(-1).contiguous(), N


EXCHANGE_MAX_RANKS = 16
EXCHANGE_MAX_ROWS = 64


def exchange(y_loc, N_loc, nbits: int, world: int, rank: int, full_ptrs, flag_ptrs, status_ptr: int, spin_limit: int = 0) -> None:
    """One exchange point of a column-sharded decode step (csrc/exchange.hip, hqq_hip_exchange): this rank's [1, N_loc[j]] slices go
    straight into every rank's full row of layer j, in the reference's column order; returns when enqueued (the kernel finishes once all
    `world` ranks have delivered).  full_ptrs[p][j] / flag_ptrs[p]: raw device addresses (see hqq_amd.shard.PeerExchange, which owns them)."""
    import ctypes
    n = len(y_loc)
    if not 1 <= n <= GEMV_MAX_GROUP:
        raise ValueError(f"hqq_amd: an exchange point holds 1..{GEMV_MAX_GROUP} layers, got {n}")
    if not 1 <= world <= EXCHANGE_MAX_RANKS or len(full_ptrs) != world or len(flag_ptrs) != world:
        raise ValueError(f"hqq_amd: 1..{EXCHANGE_MAX_RANKS} ranks, one row set and one flag block per rank")
    M = y_loc[0].numel() // int(N_loc[0])
    if not 1 <= M <= EXCHANGE_MAX_ROWS:
        raise ValueError(f"hqq_amd: exchange takes 1..{EXCHANGE_MAX_ROWS} activation rows")
    for t, nl in zip(y_loc, N_loc):
        _dev(t)
        if t.numel() != M * nl or not t.is_contiguous() or t.element_size() != 2:
            raise ValueError("hqq_amd: exchange takes dense 2-byte activations, the same number of rows for every layer: y_loc[j] is [M, N_loc[j]]")
    dt = _dt(y_loc[0].dtype)
    VPn = ctypes.c_void_p * n
    VPf = ctypes.c_void_p * (world * n)
    VPw = ctypes.c_void_p * world
    with torch.cuda.device(y_loc[0].device):
        rc = _C.lib().hqq_hip_exchange(n, VPn(*[_p(t) for t in y_loc]), (ctypes.c_int64 * n)(*[int(v) for v in N_loc]), int(M), int(nbits), dt, int(world), int(rank),
                                       VPf(*[int(full_ptrs[p][j]) for p in range(world) for j in range(n)]), VPw(*[int(v) for v in flag_ptrs]),
                                       ctypes.c_void_p(int(status_ptr)), int(spin_limit), _stream())
    _C.check(rc, "hqq_hip_exchange")


def gemm(x, W_q, scale, zero, bias, N, K, group_size, nbits, out=None, opts=None) -> Tensor:
    """fused unpack->dequant->MFMA GEMM (prefill)."""
    return _fwd("hqq_hip_gemm", x, W_q, scale, zero, bias, N, K, group_size, nbits, out, opts)


# Which path `forward` takes by the number of activation rows M (measured on MI355X, tools/prefill_routes.py -> profiles/r04_prefill_routes_int4.txt):
#   M <= 16 (<= 64 where skinny_covers): the weight-streaming decode kernels;
#   65 <= M <= 2560 (hqq_hip_forward_prefers_fused): the pipelined split-K fused MFMA dequant-GEMM (gemm_pipe.hip);
#   beyond: the HIP dequantise kernel + the in-tree dense MFMA GEMM (hqq_hip_gemm_dense, gemm_dense.hip) — one extra write + read of the
#       fp16 weights (11-36 us), then the weights are rebuilt once, not once per 256-token tile: 1.0-1.2 PFLOP/s at M = 8192 against
#       0.87-0.95 for the fused kernel.  No library GEMM on any product path (`library_gemm=True` is the bench's comparison leg).
# `fused=True` forces the fused kernels for every M.  LIBRARY_GEMM_MIN_M (a historical name: the composition's GEMM is the in-tree one) applies to
# the decode-sized cases the skinny kernel does not cover.
LIBRARY_GEMM_MIN_M = 17


def skinny_covers(dtype, M, N, K, group_size, nbits, w3s: bool = False) -> bool:
    """a batch of up to SKINNY_MAX_M rows that the weight-streaming skinny-GEMM kernel serves (csrc/skinny.hip: skinny_covers); 3-bit layers
    in the stream layout only (w3s=True)"""
    if nbits == 3:
        return bool(w3s) and dtype in (torch.float16, torch.bfloat16) and group_size == 64 and 5 <= M <= SKINNY_MAX_M and K % 256 == 0 and K >= 512 and N % 2 == 0
    return (dtype in (torch.float16, torch.bfloat16) and nbits in (8, 4, 2) and group_size == 64 and 5 <= M <= SKINNY_MAX_M and K % 256 == 0 and K >= 512
            and N % (8 // nbits) == 0)


def decode_covers(dtype, M, N, K, group_size, nbits) -> bool:
    """what hqq_hip_gemv serves for M <= GEMV_MAX_M rows (include/hqq_hip.h); everything else is composed in `forward`"""
    if M > GEMV_MAX_M or not group_size or K % group_size:
        return False
    if nbits == 3:   # x (+ a 16-group tail) is staged in LDS: 144 KiB bound M * K; a row's groups must fit inside one slab
        G = K // 64
        return (dtype == torch.float16 and group_size == 64 and M <= 4 and M * (K + 1024) * 2 + 256 <= 144 * 1024
                and (N * G + 9) // 10 >= G)
    if nbits not in (8, 4, 2, 1) or group_size % 16 or K % 16 or N % (8 // nbits):
        return False
    if dtype == torch.bfloat16:
        return nbits in (4, 2) and M <= 4
    return dtype == torch.float16 and (M <= 4 or K % 64 == 0)


def gemm_dense(x: Tensor, W: Tensor, bias=None, out: Tensor | None = None) -> Tensor:
    """HQQLinear.matmul on dequantised weights: x [*, K] @ W[N, K].T (+ bias) on the in-tree MFMA GEMM (csrc/gemm_dense.hip), fp16 / bf16"""
    _dev(x, W, bias)
    N, K = W.shape
    if x.shape[-1] != K or x.dtype != W.dtype or (bias is not None and bias.dtype != W.dtype):
        raise TypeError("hqq_amd: gemm_dense takes x [*, K], W [N, K] and bias of one compute dtype")
    x2 = x.reshape(-1, K)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    M = x2.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    if M > 0:
        with torch.cuda.device(x.device):
            rc = _C.lib().hqq_hip_gemm_dense(_p(x2), _p(W.contiguous()), _p(bias), _p(out), M, N, K, _dt(x.dtype), _stream())
        _C.check(rc, "hqq_hip_gemm_dense")
    return out.reshape(*x.shape[:-1], N)


def dense_covers(dtype, N, K) -> bool:
    return dtype in (torch.float16, torch.bfloat16) and K % 64 == 0 and K >= 64 and N % 4 == 0


# Every composed forward (what the fused kernels do not cover: group sizes other than 64 outside gemm.hip's 4- / 2-bit fp16 tiles, K % 128 != 0, 8-bit / bf16 with
# other group sizes) runs dequantise kernel + the in-tree MFMA GEMM from 17 rows on — no library GEMM on any fp16 / bf16 axis-1 path (round 6; until then
# 17..2560 rows of such layers went to torch.matmul).  The dense kernel's 256 x 256 tiles leave CUs idle below ~2000 rows (16 workgroups at 256 rows of a
# 4096-wide layer: ~90 us where a library's small-tile kernels take ~40): a known cost on shapes outside BASELINE.json's, `library_gemm=True` is the opt-out.
DENSE_MIN_M = 17


def _compose(x, W, bias, out, N, K, library: bool) -> Tensor:
    """the route after the dequantise kernel: the in-tree MFMA GEMM from DENSE_MIN_M rows on; torch.matmul — what the reference itself calls — below that
    (decode-sized residue), for shapes the in-tree kernel does not cover (K % 64 != 0, N % 4 != 0), or when asked for (library=True)"""
    if not library and dense_covers(x.dtype, N, K) and x.numel() // K >= DENSE_MIN_M:
        return gemm_dense(x, W, bias, out=None if out is None else out.reshape(-1, N))
    y = torch.matmul(x.reshape(-1, K), W.t(), out=out)
    if bias is not None:
        y += bias
    return y.reshape(*x.shape[:-1], N)


def forward(x, W_q, scale, zero, bias, N, K, group_size, nbits, out=None, fused=None, opts=None, library_gemm: bool = False) -> Tensor:
    """y = x @ dequantize(W_q)^T (+ bias).  M <= 16 (<= 64 where the skinny-GEMM kernel applies): weight-streaming decode kernels;
    larger M: fused MFMA dequant-GEMM to 2560 rows, beyond — and for what the fused kernels do not cover, unless fused=True — the dequantise
    kernel + the in-tree dense MFMA GEMM (library_gemm=True: a library GEMM instead, the bench's comparison; also the residual route for
    K % 64 != 0 or N % 4 != 0).  Same dequantised weights either way.  fused=None also composes the few decode-sized cases the kernels do not cover (3-bit beyond 4 rows,
    bf16 beyond 4 rows outside the skinny-GEMM kernel, 5..16 rows with K % 64 != 0); fused=True never composes: an uncovered configuration raises."""
    M = x.numel() // K if K else 0
    if x.dtype != scale.dtype or zero.dtype != scale.dtype or (bias is not None and bias.dtype != scale.dtype):
        raise TypeError("hqq_amd: x / scale / zero / bias must share the compute dtype")
    if nbits == 3 and (_opts(opts) & OPT_W3S):
        # the 3-bit stream layout: the 4-bit container's kernels (1..4 rows: row-per-wave GEMV; 5..64: the skinny GEMM); beyond, the reference
        # container is restored on the fly for the dequantise kernel + dense GEMM (long prompts of a patched 3-bit layer)
        if M <= 4 or (M <= SKINNY_MAX_M and group_size == 64 and K % 256 == 0 and K >= 512 and x.dtype in (torch.float16, torch.bfloat16)) or fused or \
                (fused is None and M > SKINNY_MAX_M and x.dtype in _DT and bool(_C.lib().hqq_hip_forward_prefers_fused(4, M, int(N), int(K), int(group_size or 0), _dt(x.dtype)))):   # (asked as a 4-bit layer: same kernels, same plan; 5..64 rows outside the batched-decode kernels' shapes compose below)
            return _fwd("hqq_hip_forward", x, W_q, scale, zero, bias, N, K, group_size, nbits, out, opts)
        W = dequantize(w3s_unpack(W_q, N, K), scale.reshape(-1), zero.reshape(-1), N, K, group_size, 3, 1)
        return _compose(x, W, bias, out, N, K, library_gemm)
    if fused is None:
        fused = skinny_covers(x.dtype, M, N, K, group_size, nbits) or \
            (decode_covers(x.dtype, M, N, K, group_size, nbits) and not (LIBRARY_GEMM_MIN_M and M >= LIBRARY_GEMM_MIN_M)) or \
            (M > GEMV_MAX_M and x.is_cuda and nbits in (8, 4, 2) and x.dtype in _DT and
             bool(_C.lib().hqq_hip_forward_prefers_fused(int(nbits), M, int(N), int(K), int(group_size or 0), _dt(x.dtype))))
    if fused:
        return _fwd("hqq_hip_forward", x, W_q, scale, zero, bias, N, K, group_size, nbits, out, opts)
    W = dequantize(W_q, scale.reshape(-1), zero.reshape(-1), N, K, group_size, nbits, 1)
    return _compose(x, W, bias, out, N, K, library_gemm)


def quantize(W: Tensor, nbits=4, group_size: int = 64, round_zero: bool = False, optimize: bool = True,
             iters: int = 20, beta: float = 10.0, lp_norm: float = 0.7, return_info: bool = False, axis: int = 1):
    """Quantizer.quantize(channel_wise=True, bitpack=True) with optimize_weights_proximal_legacy, fused with packing.
    axis=1: groups are runs of `group_size` consecutive elements — returns (W_q packed [packed_rows(R), gs], scale [R,1] f32 (already
    inverted), zero [R,1] f32), R = numel / gs.  axis=0: W is viewed as [gs, C], C = numel / gs, every column a group — returns
    (W_q packed [packed_rows(gs), C], scale [1,C], zero [1,C]).  [+ info int32[2] on device with return_info]"""
    _dev(W)
    if axis not in (0, 1):
        raise ValueError("axis should be either 0 or 1")
    if W.dtype not in (torch.float32, torch.float16, torch.bfloat16):
        W = W.float()
    W = W.contiguous()
    numel = W.numel()
    if group_size is None or numel % group_size:
        raise ValueError("group_size should be divisble by the total tensor dimensions. shape: "
                         f"{tuple(W.shape)}, group_size: {group_size}")   # quantize.py:94-100
    pack_bits = PACK_BITS[nbits]
    max_v = int(round(2 ** nbits - 1))
    R = numel // group_size
    dev = W.device
    if axis == 1:
        prow = packed_rows(pack_bits, R)
        W_q = torch.empty((prow, group_size), dtype=torch.int32 if pack_bits == 3 else torch.uint8, device=dev)
        scale = torch.empty((R, 1), dtype=torch.float32, device=dev)
        zero = torch.empty((R, 1), dtype=torch.float32, device=dev)
    else:
        prow = packed_rows(pack_bits, group_size)
        W_q = torch.empty((prow, R), dtype=torch.int32 if pack_bits == 3 else torch.uint8, device=dev)
        scale = torch.empty((1, R), dtype=torch.float32, device=dev)
        zero = torch.empty((1, R), dtype=torch.float32, device=dev)
    info = torch.zeros((2,), dtype=torch.int32, device=dev)
    L = _C.lib()
    it = iters if optimize else 0
    ws_bytes = L.hqq_hip_quantize_workspace_bytes(numel, group_size, it)
    if ws_bytes == 0:
        raise ValueError(f"hqq_amd: bad quantize arguments (numel={numel}, group_size={group_size}, iters={it})")
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    fn = L.hqq_hip_quantize if axis == 1 else L.hqq_hip_quantize_axis0
    with torch.cuda.device(dev):
        rc = fn(_p(W), _dt(W.dtype), numel, group_size, max_v, pack_bits, int(bool(round_zero)), int(bool(optimize)),
                it, float(beta), float(lp_norm), _p(W_q), _p(scale), _p(zero), _p(info), _p(ws), ws_bytes, _stream())
    _C.check(rc, "hqq_hip_quantize" if axis == 1 else "hqq_hip_quantize_axis0")
    if return_info:
        return W_q, scale, zero, info
    return W_q, scale, zero


def optimize(W: Tensor, scale: Tensor, zero: Tensor, max_v: int, axis: int = 1, iters: int = 20, beta: float = 10.0, lp_norm: float = 0.7):
    """optimize_weights_proximal_legacy on its own (optimize.py:208-255): W is the grouped 2-D view ([groups, gs] for axis=1, [gs, groups] for
    axis=0), scale / zero float32 with one value per group.  Returns (levels uint8 in W's shape, zero float32 in zero's shape)."""
    _dev(W, scale, zero)
    if W.dtype not in (torch.float32, torch.float16, torch.bfloat16):
        W = W.float()
    W = W.contiguous()
    gs = W.shape[1] if axis == 1 else W.shape[0]
    R = W.numel() // gs
    sc, ze = scale.reshape(-1).float().contiguous(), zero.reshape(-1).float().contiguous()
    if sc.numel() != R or ze.numel() != R:
        raise ValueError(f"hqq_amd: scale / zero must hold one value per group ({R}), got {sc.numel()} / {ze.numel()}")
    dev = W.device
    levels = torch.empty(W.shape, dtype=torch.uint8, device=dev)
    zero_out = torch.empty(zero.shape, dtype=torch.float32, device=dev)
    info = torch.zeros((2,), dtype=torch.int32, device=dev)
    L = _C.lib()
    ws_bytes = int(L.hqq_hip_quantize_workspace_bytes(W.numel(), gs, int(iters))) + 4 * R
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = L.hqq_hip_optimize(_p(W), _dt(W.dtype), W.numel(), gs, int(axis), int(max_v), _p(sc), _p(ze), int(iters), float(beta), float(lp_norm),
                                _p(levels), _p(zero_out), _p(info), _p(ws), ws_bytes, _stream())
    _C.check(rc, "hqq_hip_optimize")
    return levels, zero_out


def quantize_tensorwise(W: Tensor, nbits=4, round_zero: bool = False):
    """Quantizer.quantize(channel_wise=False) (quantize.py:114-116): one scale / zero from the tensor's min and max, no solver; the
    levels packed in the tensor's own 2-D shape.  Returns (W_q packed [packed_rows(rows), cols], scale 0-d f32 (inverted), zero 0-d f32)."""
    _dev(W)
    if W.dim() != 2:
        raise ValueError("hqq_amd: quantize_tensorwise takes a 2-D tensor")
    if W.dtype not in (torch.float32, torch.float16, torch.bfloat16):
        W = W.float()
    W = W.contiguous()
    rows, cols = W.shape
    pack_bits = PACK_BITS[nbits]
    dev = W.device
    W_q = torch.empty((packed_rows(pack_bits, rows), cols), dtype=torch.int32 if pack_bits == 3 else torch.uint8, device=dev)
    meta = torch.empty((2,), dtype=torch.float32, device=dev)
    ws = torch.empty((16384,), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = _C.lib().hqq_hip_quantize_tensor(_p(W), _dt(W.dtype), rows, cols, int(round(2 ** nbits - 1)), pack_bits, int(bool(round_zero)),
                                              _p(W_q), _p(meta), meta.data_ptr() + 4, _p(ws), ws.numel(), _stream())
    _C.check(rc, "hqq_hip_quantize_tensor")
    return W_q, meta[0], meta[1]


# ---- the steps either side of the GEMVs in a decode step (csrc/block.hip; include/hqq_hip.h) --------------------------------------------
def add_rmsnorm(h: Tensor, delta, weight: Tensor, eps: float, out: Tensor | None = None) -> Tensor:
    """h += delta (in place, if delta is given), then LlamaRMSNorm(h) with `weight` — HF's arithmetic, one kernel.  h: [..., H] fp16, dense."""
    _dev(h, delta, weight)
    H = h.shape[-1]
    if out is None:
        out = torch.empty_like(h)
    with torch.cuda.device(h.device):
        rc = _C.lib().hqq_hip_add_rmsnorm(_p(h), _p(delta), _p(weight), float(eps), _p(out), h.numel() // H, H, _dt(h.dtype), _stream())
    _C.check(rc, "hqq_hip_add_rmsnorm")
    return out


def rope_cache(q: Tensor, k: Tensor, v: Tensor, cos: Tensor, sin: Tensor, pos: Tensor, k_cache: Tensor, v_cache: Tensor, q_out: Tensor) -> Tensor:
    """one token: q_out = rotary(q); rotary(k) and v go into the static caches [n_kv_heads, cache_len, head_dim] at position pos[0] (device int64)"""
    _dev(q, k, v, cos, sin, pos, k_cache, v_cache, q_out)
    hd = cos.shape[-1]
    if pos.dtype != torch.int64 or k_cache.shape[-1] != hd or not k_cache.is_contiguous() or not v_cache.is_contiguous():
        raise ValueError("hqq_amd: rope_cache takes an int64 position tensor and dense [n_kv_heads, cache_len, head_dim] caches")
    with torch.cuda.device(q.device):
        rc = _C.lib().hqq_hip_rope_cache(_p(q), _p(k), _p(v), _p(cos), _p(sin), _p(pos), _p(q_out), _p(k_cache), _p(v_cache), q.numel() // hd, k.numel() // hd, hd,
                                         k_cache.shape[-2], _dt(q.dtype), _stream())
    _C.check(rc, "hqq_hip_rope_cache")
    return q_out


def attn_decode(q: Tensor, k_cache: Tensor, v_cache: Tensor, pos: Tensor, out: Tensor, scaling: float, splits: int = 1, workspace: Tensor | None = None) -> Tensor:
    """one query per head against the static KV cache's first pos + 1 positions (fp16 / bf16; within rounding of SDPA, not bit-identical):
    q [n_heads, hd] (any view of n_heads * hd contiguous values), k_cache / v_cache [n_kv, cache_len, hd], pos int64[1] on the device, out [n_heads * hd];
    splits > 1: the keys of a head shared out over that many workgroups + a merging launch (long caches; attn_splits / attn_workspace)"""
    _dev(q, k_cache, v_cache, pos, out)
    n_kv, L, hd = k_cache.shape
    n_heads = q.numel() // hd
    if splits > 1 and workspace is None:
        workspace = attn_workspace(q.device, n_heads, hd, splits)
    with torch.cuda.device(q.device):
        rc = _C.lib().hqq_hip_attn_decode(_p(q), _p(k_cache), _p(v_cache), _p(pos), _p(out), n_heads, n_kv, hd, L, float(scaling), _dt(q.dtype),
                                          int(splits), _p(workspace), 0 if workspace is None else workspace.numel(), _stream())
    _C.check(rc, "hqq_hip_attn_decode")
    return out


def attn_splits(kv_len: int) -> int:
    """how many workgroups share a head's keys in the decode-attention kernel when up to kv_len of them are visible (1: no second launch)"""
    return 1 if kv_len <= 1024 else min(16, int(kv_len) // 512)


def attn_workspace(device, n_heads: int, head_dim: int, splits: int):
    """the (uninitialised) record buffer of a split launch, or None"""
    nb = int(_C.lib().hqq_hip_attn_decode_workspace_bytes(int(n_heads), int(head_dim), int(splits)))
    return torch.empty(nb, dtype=torch.uint8, device=device) if nb else None


def rope_attn_decode(q: Tensor, k: Tensor, v: Tensor, cos: Tensor, sin: Tensor, pos: Tensor, k_cache: Tensor, v_cache: Tensor, out: Tensor, scaling: float,
                     splits: int = 1, workspace: Tensor | None = None) -> Tensor:
    """rope_cache + attn_decode in one launch: raw q / k / v projections in, rotary applied in the kernel, the new key / value used from on-chip
    memory and written to the cache at `pos` for the following steps (the cache ends up bit-identical to rope_cache's)"""
    _dev(q, k, v, cos, sin, pos, k_cache, v_cache, out)
    n_kv, L, hd = k_cache.shape
    n_heads = q.numel() // hd
    if splits > 1 and workspace is None:
        workspace = attn_workspace(q.device, n_heads, hd, splits)
    with torch.cuda.device(q.device):
        rc = _C.lib().hqq_hip_rope_attn_decode(_p(q), _p(k), _p(v), _p(cos), _p(sin), _p(pos), _p(k_cache), _p(v_cache), _p(out), n_heads, n_kv, hd, L,
                                                float(scaling), _dt(q.dtype), int(splits), _p(workspace), 0 if workspace is None else workspace.numel(), _stream())
    _C.check(rc, "hqq_hip_rope_attn_decode")
    return out


def token_prologue(tok: Tensor, pos: Tensor, embed: Tensor, h: Tensor, cos_tab=None, sin_tab=None, cos=None, sin=None, mask=None) -> None:
    """The front of a decode step in one launch (csrc/block.hip, hqq_hip_token_prologue): h[H] = embed[tok]; cos / sin = row pos of the rotary tables
    [L, head_dim] (skipped when the tables are None); mask[L] = 0 up to pos, -inf beyond (skipped when None).  tok [1, 1] / pos [1] int64 on the device:
    graph-replay safe.  Copies and compares only: the same bits as embed_tokens(tok), index_select and torch.where produce."""
    _dev(tok, pos, embed, h)
    if tok.dtype != torch.int64 or pos.dtype != torch.int64 or embed.dim() != 2 or not embed.is_contiguous() or h.numel() != embed.shape[1] or h.dtype != embed.dtype:
        raise ValueError("hqq_amd: token_prologue takes int64 tok / pos, a dense [vocab, H] embedding and h [H] of its dtype")
    L, hd = 1, 0
    if cos_tab is not None:
        _dev(cos_tab, sin_tab, cos, sin)
        if cos_tab.shape != sin_tab.shape or cos_tab.dim() != 2 or not (cos_tab.is_contiguous() and sin_tab.is_contiguous()) or cos.numel() != cos_tab.shape[1] or \
                sin.numel() != cos_tab.shape[1] or any(t.dtype != embed.dtype for t in (cos_tab, sin_tab, cos, sin)):
            raise ValueError("hqq_amd: token_prologue takes dense [L, head_dim] rotary tables and [head_dim] outputs of the compute dtype")
        L, hd = int(cos_tab.shape[0]), int(cos_tab.shape[1])
    if mask is not None:
        _dev(mask)
        if mask.dtype != embed.dtype or not mask.is_contiguous() or (cos_tab is not None and mask.numel() != L):
            raise ValueError("hqq_amd: token_prologue's mask is a dense [L] tensor of the compute dtype, L the rotary tables' rows")
        L = int(mask.numel())
    with torch.cuda.device(h.device):
        rc = _C.lib().hqq_hip_token_prologue(_p(tok), _p(pos), _p(embed), int(embed.shape[0]), int(embed.shape[1]), _p(cos_tab), _p(sin_tab), L, hd, _p(h), _p(cos), _p(sin),
                                             _p(mask), _dt(embed.dtype), _stream())
    _C.check(rc, "hqq_hip_token_prologue")


def argmax_advance(logits: Tensor, next_tok: Tensor, tok: Tensor | None = None, pos: Tensor | None = None) -> None:
    """The back of a greedy decode step in one launch (hqq_hip_argmax_advance): next_tok[0] = logits.argmax() (the first index of the largest value, as torch.argmax),
    tok[0] = the same, pos[0] += 1 (each skipped when None).  int64 tensors on the device: graph-replay safe."""
    _dev(logits, next_tok)
    if not logits.is_contiguous() or next_tok.dtype != torch.int64 or (tok is not None and tok.dtype != torch.int64) or (pos is not None and pos.dtype != torch.int64):
        raise ValueError("hqq_amd: argmax_advance takes dense logits and int64 token / position tensors")
    with torch.cuda.device(logits.device):
        rc = _C.lib().hqq_hip_argmax_advance(_p(logits), logits.numel(), _dt(logits.dtype), _p(next_tok), _p(tok), _p(pos), _stream())
    _C.check(rc, "hqq_hip_argmax_advance")


def silu_mul(gate: Tensor, up: Tensor, out: Tensor | None = None) -> Tensor:
    """LlamaMLP's act_fn(gate) * up in one kernel (fp16)"""
    _dev(gate, up)
    if out is None:
        out = torch.empty_like(gate)
    with torch.cuda.device(gate.device):
        rc = _C.lib().hqq_hip_silu_mul(_p(gate), _p(up), _p(out), gate.numel(), _dt(gate.dtype), _stream())
    _C.check(rc, "hqq_hip_silu_mul")
    return out
