"""(lab) Python wrappers of the chained launches (gemv_chain.hip) and the persistent decode engine (engine.hip) — cut out of hqq_amd/ops.py when the two kernels left the library (round 4).  They bind hqq_hip_gemv_chained / hqq_hip_decode_* of a lab build."""
import torch
from hqq_amd import _C
from hqq_amd.ops import *   # noqa
from hqq_amd.ops import _dev, _p, _dt, _stream

# ---- chained decode launches (csrc/gemv_chain.hip; include/hqq_hip.h "Chained decode launches") --------------------------------
CHAIN_COUNTER_BYTES = 4096   # HQQ_CHAIN_COUNTER_BYTES


class _ChainLink(__import__("ctypes").Structure):
    """hqq_hip_chain_link (include/hqq_hip.h)"""
    import ctypes as _ct
    _fields_ = [("wait", _ct.c_void_p), ("signal", _ct.c_void_p), ("status", _ct.c_void_p), ("wait_arrivals", _ct.c_uint32), ("spin_limit", _ct.c_uint32)]


def gemv_chained(x: Tensor, layers, K: int, group_size: int, nbits: int, outs, opts: int = 0, wait: int | None = None, wait_arrivals: int = 0,
                 signal: int | None = None, status: int | None = None, spin_limit: int = 0) -> int:
    """One link of a chain of overlapped decode launches: gemv_grouped's arithmetic (same bits), but the launch does its x-independent
    work first and then waits in the kernel for the arrival counters at device address `wait` to reach `wait_arrivals`.
    layers: sequence of (W_q, scale, zero, bias_or_None, N); outs: the [M, N_i] output buffers.  wait / signal / status are raw device
    addresses (see LaunchChain, which owns them).  Returns the arrivals this launch adds to `signal`."""
    import ctypes
    n = len(layers)
    if not 1 <= n <= GEMV_MAX_GROUP:
        raise ValueError(f"hqq_amd: a GEMV group holds 1..{GEMV_MAX_GROUP} layers, got {n}")
    if x.shape[-1] != K or not x.is_contiguous():
        raise ValueError(f"hqq_amd: x must be dense with {K} features")
    M = x.numel() // K
    for (W_q, s, z, b, N) in layers:
        _dev(x, W_q, s, z, b)
        if x.dtype != s.dtype or z.dtype != s.dtype or (b is not None and b.dtype != s.dtype):
            raise TypeError("hqq_amd: x / scale / zero / bias must share the compute dtype")
    VP = ctypes.c_void_p * n
    has_bias = any(L[3] is not None for L in layers)
    link = _ChainLink(wait or None, signal or None, status or None, int(wait_arrivals), int(spin_limit))
    arrivals = ctypes.c_uint32(0)
    with torch.cuda.device(x.device):
        rc = _C.lib().hqq_hip_gemv_chained(
            int(nbits), n, _p(x), VP(*[_p(L[0]) for L in layers]), VP(*[_p(L[1]) for L in layers]), VP(*[_p(L[2]) for L in layers]),
            VP(*[_p(L[3]) for L in layers]) if has_bias else None, VP(*[_p(o_) for o_ in outs]),
            (ctypes.c_int64 * n)(*[int(L[4]) for L in layers]), M, K, group_size, _dt(x.dtype), int(opts), ctypes.byref(link),
            ctypes.byref(arrivals), _stream())
    _C.check(rc, "hqq_hip_gemv_chained")
    return int(arrivals.value)


class LaunchChain:
    """A list of dependent decode stages run as overlapped launches on two streams (csrc/gemv_chain.hip).

    stages: sequence of (x [M, K], layers) with layers = sequence of (W_q, scale, zero, bias_or_None, N, y [M, N]); stage s + 1 may read
    an output buffer of stage s (that is the point).  opts: one int or one per stage (0 / OPT_META_SCALABLE).  run() enqueues on the
    current stream and a side stream owned by this object, and joins them again: capturable in one hipGraph."""

    def __init__(self, stages, nbits: int, group_size: int = 64, opts=0, spin_limit: int = 0):
        n = len(stages)
        if n < 1:
            raise ValueError("hqq_amd: a launch chain needs at least one stage")
        self.stages = [(x, list(layers)) for x, layers in stages]
        self.nbits, self.group_size = int(nbits), int(group_size)
        self.opts = [int(opts)] * n if isinstance(opts, int) else [int(o) for o in opts]
        self.device = stages[0][0].device
        words = CHAIN_COUNTER_BYTES // 4
        self._sync = torch.zeros(n * words + 32, dtype=torch.int32, device=self.device)   # counters of every link | status word
        assert self._sync.data_ptr() % 128 == 0
        self._words = words
        self.spin_limit = int(spin_limit)
        with torch.cuda.device(self.device):
            self._side = torch.cuda.Stream()

    def _ctr(self, s: int) -> int:
        return self._sync.data_ptr() + 4 * self._words * s

    def run(self) -> None:
        n = len(self.stages)
        main = torch.cuda.current_stream(self.device)
        self._sync.zero_()                 # every counter block and the status word (a node of its own under capture)
        self._side.wait_stream(main)
        status = self._sync.data_ptr() + 4 * self._words * n
        arrivals = 0
        for s, (x, layers) in enumerate(self.stages):
            with torch.cuda.stream(main if s % 2 == 0 else self._side):
                arrivals = gemv_chained(x, [L[:5] for L in layers], x.shape[-1], self.group_size, self.nbits, [L[5] for L in layers], self.opts[s],
                                        wait=self._ctr(s - 1) if s > 0 else None, wait_arrivals=arrivals,
                                        signal=self._ctr(s) if s + 1 < n else None, status=status, spin_limit=self.spin_limit)
        main.wait_stream(self._side)

    def status(self) -> int:
        """0 after a run in which every in-kernel wait completed (synchronises)"""
        return int(self._sync[self._words * len(self.stages)].item())


class _StageDesc(__import__("ctypes").Structure):
    """hqq_hip_decode_stage (include/hqq_hip.h)"""
    import ctypes as _ct
    _fields_ = [("x", _ct.c_void_p), ("K", _ct.c_int64), ("n_layers", _ct.c_int32), ("reserved", _ct.c_int32),
                ("Wq", _ct.c_void_p * GEMV_MAX_GROUP), ("scale", _ct.c_void_p * GEMV_MAX_GROUP), ("zero", _ct.c_void_p * GEMV_MAX_GROUP),
                ("bias", _ct.c_void_p * GEMV_MAX_GROUP), ("y", _ct.c_void_p * GEMV_MAX_GROUP), ("N", _ct.c_int64 * GEMV_MAX_GROUP)]


class DecodePlan:
    """One launch for a whole list of dependent decode stages (csrc/engine.hip; include/hqq_hip.h "persistent decode engine").

    stages: sequence of (x [1, K], layers) with layers = sequence of (W_q, scale, zero, bias_or_None, N, y [1, N]); stage s + 1 may
    read an output buffer of stage s.  All tensors stay owned by the caller (this object keeps references)."""

    def __init__(self, stages, nbits: int, group_size: int = 64, opts: int = 0, grid: int = 0):
        import ctypes
        L = _C.lib()
        n = len(stages)
        if n < 1:
            raise ValueError("hqq_amd: a decode plan needs at least one stage")
        descs = (_StageDesc * n)()
        self._keep = []
        dev = None
        for i, (x, layers) in enumerate(stages):
            if not 1 <= len(layers) <= GEMV_MAX_GROUP:
                raise ValueError(f"hqq_amd: a stage holds 1..{GEMV_MAX_GROUP} layers, got {len(layers)}")
            K = x.shape[-1]
            if x.numel() != K or not x.is_contiguous():
                raise ValueError("hqq_amd: the decode engine takes one dense activation row per stage")
            _dev(x)
            dev = x.device if dev is None else dev
            d = descs[i]
            d.x, d.K, d.n_layers = x.data_ptr(), K, len(layers)
            for j, (W_q, s, z, b, N, y) in enumerate(layers):
                _dev(W_q, s, z, b, y)
                if x.dtype != s.dtype or z.dtype != s.dtype or y.dtype != s.dtype or (b is not None and b.dtype != s.dtype):
                    raise TypeError("hqq_amd: x / scale / zero / bias / y must share the compute dtype")
                if y.numel() != N or not y.is_contiguous():
                    raise ValueError("hqq_amd: y must be a dense [1, N] buffer")
                d.Wq[j], d.scale[j], d.zero[j], d.bias[j], d.y[j], d.N[j] = _p(W_q), _p(s), _p(z), _p(b), _p(y), int(N)
                self._keep += [W_q, s, z, b, y]
            self._keep.append(x)
        self.device = dev
        self.nbytes = int(L.hqq_hip_decode_plan_bytes(n))
        self._host = ctypes.create_string_buffer(self.nbytes)
        rc = L.hqq_hip_decode_plan_init(ctypes.addressof(self._host), self.nbytes, int(nbits), int(group_size), _dt(stages[0][0].dtype), 1, int(opts),
                                        ctypes.addressof(descs), n, int(grid))
        _C.check(rc, "hqq_hip_decode_plan_init")
        self._dev = torch.frombuffer(bytearray(self._host.raw), dtype=torch.uint8).to(dev)
        assert self._dev.data_ptr() % 256 == 0
        self._status_off = int(L.hqq_hip_decode_plan_status_offset(ctypes.addressof(self._host)))
        self.n_stages = n

    def run(self) -> None:
        import ctypes
        with torch.cuda.device(self.device):
            rc = _C.lib().hqq_hip_decode_run(ctypes.addressof(self._host), self._dev.data_ptr(), self.nbytes, _stream())
        _C.check(rc, "hqq_hip_decode_run")

    def status(self) -> int:
        """0 after a run in which every inter-workgroup hand-off completed (synchronises)"""
        return int(self._dev[self._status_off:self._status_off + 4].view(torch.int32).item())


