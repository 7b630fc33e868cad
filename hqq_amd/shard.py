"""Output-column shard of a quantised layer across the GPUs of one node + the RCCL all-gather that completes it.

y[:, n] = sum_k x[:, k] W[n, k]: with axis=1 every quantisation group lies inside one row of W, so the scale / zero / W_q
values of a row subset are those of the full layer.  Rows n and n + slot*N/per share a packed byte, so
a rank does not own a contiguous range of output rows but the packed-row block
    packed rows [r*Np/P, (r+1)*Np/P),  Np = N/per          (a zero-copy slice of the reference W_q)
i.e. the `per` output slabs  slot*N/per + [r*n', (r+1)*n'),  n' = N/(per*P).  Seen on its own that slice is a complete
packed layer with N/P rows; the all-gather returns [P, M, N/P] in (rank, slab, n') order, and `unpermute` maps it back.
3-bit containers mix rows of unrelated slabs (step = ceil(R/10)), so the shard is re-packed from the unpacked rows — and, on a GPU, kept in the 3-bit
stream layout (csrc/w3s.h) so that a sharded int3 layer runs the same one-launch kernels as a patched unsharded one (ShardedHQQForward; round 5).
"""
from __future__ import annotations

import torch
from torch import Tensor

from . import ops


def shard_rows(N: int, nbits: int, rank: int, world: int) -> Tensor:
    """global output rows owned by `rank`, in the local row order of its packed slice"""
    per = 1 if nbits == 3 else ops.PER[nbits]
    if N % (per * world):
        raise ValueError(f"out_features={N} cannot be split over {world} ranks at {nbits} bits (needs N % {per * world} == 0)")
    n1 = N // (per * world)
    base = torch.arange(n1) + rank * n1
    return torch.cat([base + slot * (N // per) for slot in range(per)])


def shard_packed(W_q: Tensor, scale: Tensor, zero: Tensor, bias, N: int, K: int, group_size: int, nbits: int, rank: int, world: int):
    """-> (W_q_local, scale_local, zero_local, bias_local, N_local): the rank's self-contained packed layer."""
    G = K // group_size
    rows = shard_rows(N, nbits, rank, world).to(scale.device)
    n_loc = N // world
    sc = scale.reshape(N, G)[rows].reshape(-1, 1).contiguous()
    ze = zero.reshape(N, G)[rows].reshape(-1, 1).contiguous()
    b = None if bias is None else bias[rows].contiguous()
    if nbits == 3:
        U = ops.unpack(3, W_q)[: N * G].reshape(N, G * group_size)[rows]
        Wl = ops.pack(3, U.reshape(n_loc * G, group_size).contiguous())
    else:
        per = ops.PER[nbits]
        prow = (N // per) * G                   # packed rows of the full layer ([prow, group_size] bytes == [N/per, K])
        p1 = prow // world
        Wl = W_q.reshape(prow, group_size)[rank * p1:(rank + 1) * p1]   # zero-copy view
    return Wl, sc, ze, b, n_loc


def unpermute(y_gathered: Tensor, N: int, nbits: int, world: int) -> Tensor:
    """[P, M, N/P] as all-gathered (rank-major, each rank in its local row order) -> [M, N] in global row order"""
    P, M, n_loc = y_gathered.shape
    per = 1 if nbits == 3 else ops.PER[nbits]
    n1 = N // (per * world)
    return y_gathered.reshape(P, M, per, n1).permute(1, 2, 0, 3).reshape(M, N)


def gather_columns(y_loc: Tensor, out_full: Tensor, N: int, nbits: int, group=None, coalesce: bool = True) -> Tensor:
    """All-gather ONE activation row straight into the reference's column order — no un-permute afterwards.

    y_loc [1, N/P] is this rank's output in its local order (slab-major: `per` runs of n' = N/(per*P) columns); out_full [1, N].  Slab s of
    the full row, columns [s*N/per, (s+1)*N/per), is the rank-major concatenation of every rank's run s — exactly what an all-gather of
    that run into that slice produces.  So one all-gather per slab, issued as ONE coalesced collective where the backend can (RCCL: a
    single grouped launch), replaces the gather of the whole shard + a permuting copy kernel.  One row only: with M > 1 rows a slab's
    columns are strided in memory (use the shard-wide gather + `unpermute` there)."""
    import torch.distributed as dist
    if y_loc.numel() * dist.get_world_size(group) != N or out_full.numel() != N:
        raise ValueError("hqq_amd: gather_columns takes one activation row: y_loc [1, N/P], out_full [1, N]")
    per = 1 if nbits == 3 else ops.PER[nbits]
    world = dist.get_world_size(group)
    n1 = N // (per * world)
    src, dst = y_loc.reshape(-1), out_full.reshape(-1)

    def issue():
        for s_ in range(per):
            dist.all_gather_into_tensor(dst[s_ * (N // per):(s_ + 1) * (N // per)], src[s_ * n1:(s_ + 1) * n1], group=group)
    cm = getattr(dist, "_coalescing_manager", None) if (coalesce and per > 1) else None
    if cm is not None and dist.get_backend(group) == "nccl":
        with cm(group=group, device=y_loc.device):
            issue()
    else:   # (gloo and friends: the same gathers one by one)
        issue()
    return out_full


# ---- which exchange points are worth sharding (round 6) ------------------------------------------------------------------------------------
# One rank's launch costs t ~ T0 + bytes / B with T0 ~ 3.8 us and B ~ 7.7 TB/s on an MI355X (profiles/r06_per_launch.txt): sharding a layer P ways
# takes bytes (1 - 1 / P) / B off the launch and adds one exchange point (a latency-bound all-gather of a few KiB, or the peer-store kernel: >= ~4 us).
# Below  bytes (1 - 1 / P) / B  <  exchange cost  the shard LOSES before anything else is counted — `bench.py`'s `shard_of_8` leg measures one rank
# of the 8-way 70B shard at 0.29 of its GPU's roofline against 0.60-0.63 for the unsharded stack.  So an exchange group (layers consumed together:
# q|k|v, o, gate|up, down) whose layers are ALL small is REPLICATED: every rank holds and computes it whole, and it has no exchange point.
# The threshold is on the group's packed bytes; nothing here has been measured over xGMI — the constant comes from the single-GPU launch model and
# is the first thing `tools/node_first_run.sh` should calibrate.
EXCHANGE_COST_US = 4.0          # assumed cost of one bs = 1 exchange point (unmeasured over xGMI)
STREAM_TB_S = 7.7               # marginal streaming rate of a decode launch on one MI355X


def replicate_below_bytes(world: int) -> int:
    """packed bytes of an exchange group below which replicating it beats sharding it `world` ways (launch model above)"""
    if world <= 1:
        return 0
    return int(EXCHANGE_COST_US * 1e-6 * STREAM_TB_S * 1e12 / (1.0 - 1.0 / world))


def plan_exchange_groups(group_bytes, world: int, threshold: int | None = None):
    """group_bytes: packed weight bytes of each exchange group of a block (whole layers, summed over the group) -> one of
    "sharded" / "replicated-small" per group.  Pure arithmetic: every rank derives the same plan."""
    thr = replicate_below_bytes(world) if threshold is None else threshold
    return ["replicated-small" if (world > 1 and b < thr) else "sharded" for b in group_bytes]


class ShardedHQQForward:
    """One rank's share of a column-sharded layer.  forward(x) = local fused forward + one all-gather over the process
    group (RCCL on GPUs; any torch.distributed backend works, the CPU tests use gloo with a stand-in local op)."""

    def __init__(self, W_q, scale, zero, bias, N, K, group_size, nbits, group=None, local_forward=None, peer=None, replicate: bool = False):
        """peer: (PeerExchange, point) — at one activation row the outputs are then exchanged by that object's kernel (peer-memory stores,
        csrc/exchange.hip) instead of a collective; the returned row is the exchange's buffer, valid until the point is used again.
        replicate: the plan (plan_exchange_groups) found this layer's exchange group too small to shard: the rank keeps the WHOLE layer and
        forward() is the local forward — no slice, no collective, the same result on every rank."""
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.peer = peer
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.N, self.K, self.gs, self.nbits = N, K, group_size, nbits
        self.replicated = bool(replicate)
        if self.replicated:
            self.Wq, self.scale, self.zero, self.bias, self.n_loc = W_q, scale, zero, bias, N
        else:
            self.Wq, self.scale, self.zero, self.bias, self.n_loc = shard_packed(W_q, scale, zero, bias, N, K, group_size, nbits, self.rank, self.world)
        self.opts = 0
        if local_forward is None and nbits == 3 and not self.replicated and self.Wq.is_cuda and ops.w3s_covers(self.n_loc, K, group_size):
            # a 3-bit shard is re-packed anyway (the reference container mixes unrelated rows): keep it in the 3-bit STREAM layout (csrc/w3s.h), which
            # runs through the 4-bit container's kernels — one launch per stage instead of two (0.38 instead of 0.16 of the HBM roofline on the 7B stack)
            self.Wq = ops.w3s_pack(self.Wq, self.n_loc, K)
            self.opts = ops.OPT_W3S | (ops.OPT_META_SCALABLE if (self.scale.dtype == torch.float16 and ops.w3s_meta_scalable(self.scale, self.zero, self.n_loc, K)) else 0)
        elif local_forward is None and nbits in (8, 4, 2) and self.Wq.is_cuda and self.scale.dtype == torch.float16 and group_size and K % group_size == 0 and \
                ops.meta_scalable(self.scale, self.zero, self.n_loc, K, group_size, nbits):
            self.opts = ops.OPT_META_SCALABLE   # (the shard is a complete packed layer: its own check decides the three-op rebuild)
        self._local = local_forward or (lambda x: ops.forward(x, self.Wq, self.scale, self.zero, self.bias, self.n_loc, K, group_size, nbits, opts=ops.layer_opts(self.opts)))

    # rows per chunk of a long prompt: the shard's GEMM of chunk c + 1 runs while chunk c's outputs are gathered (at
    # M = 65,536 a rank receives 896 MiB per layer — the gather is as long as the GEMM unless it overlaps it).  0 disables chunking
    OVERLAP_ROWS = 4096

    def _forward_chunked(self, x2: Tensor, rows: int) -> Tensor:
        """long prompts: per chunk of `rows` activation rows, the local GEMM on the current stream, then an ASYNC all-gather of its [rows, N / P]
        outputs (the collective library runs it on its own stream behind an event: the next chunk's GEMM overlaps it), un-permuted into the
        reference's column order once it has arrived"""
        M = x2.shape[0]
        out = torch.empty((M, self.N), dtype=x2.dtype, device=x2.device)
        pending = []
        for c0 in range(0, M, rows):
            y_c = self._local(x2[c0:c0 + rows]).reshape(-1, self.n_loc).contiguous()
            buf = torch.empty((self.world * y_c.shape[0], self.n_loc), dtype=y_c.dtype, device=y_c.device)
            work = self.dist.all_gather_into_tensor(buf, y_c, group=self.group, async_op=True)
            pending.append((work, buf, y_c, c0))            # (y_c is kept alive until its gather has completed)
            if len(pending) > 1:                            # at most two chunks in flight: drain the older one
                self._drain(pending.pop(0), out)
        while pending:
            self._drain(pending.pop(0), out)
        return out

    def _drain(self, item, out: Tensor) -> None:
        work, buf, y_c, c0 = item
        work.wait()
        mc = y_c.shape[0]
        out[c0:c0 + mc] = unpermute(buf.view(self.world, mc, self.n_loc), self.N, self.nbits, self.world)

    def forward(self, x: Tensor) -> Tensor:
        if self.replicated:   # the whole layer on every rank: nothing to exchange
            return self._local(x).reshape(*x.shape[:-1], self.N)
        rows_in = x.numel() // self.K
        if self.OVERLAP_ROWS and rows_in >= 2 * self.OVERLAP_ROWS and self.world > 1:
            return self._forward_chunked(x.reshape(-1, self.K), self.OVERLAP_ROWS).reshape(*x.shape[:-1], self.N)
        y_loc = self._local(x).reshape(-1, self.n_loc).contiguous()
        M = y_loc.shape[0]
        if self.peer is not None and M <= self.peer[0].rows:   # one row, or a decode batch the arena holds: peer-memory stores, nothing to un-permute
            px, e = self.peer
            px.run(e, [y_loc])
            return px.full(e, 0, rows=M).reshape(*x.shape[:-1], self.N)
        if M == 1:   # decode: straight into the reference's column order, no un-permute
            full = torch.empty((1, self.N), dtype=y_loc.dtype, device=y_loc.device)
            return gather_columns(y_loc, full, self.N, self.nbits, self.group).reshape(*x.shape[:-1], self.N)
        out = torch.empty((self.world * M, self.n_loc), dtype=y_loc.dtype, device=y_loc.device)   # rank-major concatenation
        self.dist.all_gather_into_tensor(out, y_loc, group=self.group)
        return unpermute(out.view(self.world, M, self.n_loc), self.N, self.nbits, self.world).reshape(*x.shape[:-1], self.N)

    __call__ = forward


# ---- fine-grained device memory for the peer-memory exchange ---------------------------------------------------------------------------
# Peers store into an arena over xGMI while the owning GPU polls flags inside a running kernel and the next kernel reads rows whose
# addresses repeat every token.  Ordinary (coarse-grained) device memory — what torch's caching allocator hands out — does not promise that
# such writes become visible during a running kernel, or that the local L2 holds no stale copy; collective libraries allocate their
# signal words and buffers uncached / fine-grained for this reason (advisor, round 3).  The arena therefore comes from
# hipExtMallocWithFlags(hipDeviceMallocUncached, else hipDeviceMallocFinegrained) on the process's own HIP runtime (ctypes; the library
# libhqq_hip.so itself never allocates), is exported with hipIpcGetMemHandle and wrapped as a tensor through __cuda_array_interface__.
# Where that is not available the torch allocator serves it as before and `memory_kind` says so.
_HIP_RT = None


def _hip_runtime():
    """the libamdhip64 this process already runs on (torch's), as a ctypes handle; None if it cannot be found"""
    global _HIP_RT
    if _HIP_RT is None:
        import ctypes
        path = None
        try:
            with open("/proc/self/maps") as f:
                for line in f:
                    if "libamdhip64" in line:
                        path = line.split()[-1]
                        break
        except OSError:
            pass
        try:
            _HIP_RT = ctypes.CDLL(path) if path else False
        except OSError:
            _HIP_RT = False
    return _HIP_RT or None


class _RawBuffer:
    """device memory this module allocated itself, seen by torch through the CUDA array interface (zero-copy)"""

    def __init__(self, ptr: int, nbytes: int, owner=None):
        self.ptr, self.nbytes, self.owner = int(ptr), int(nbytes), owner
        self.__cuda_array_interface__ = {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.ptr, False), "version": 2}


def _alloc_fine_grained(nbytes: int, device: torch.device):
    """-> (tensor uint8 [nbytes], kind, raw pointer) from hipExtMallocWithFlags, or None"""
    import ctypes
    rt = _hip_runtime()
    if rt is None or not hasattr(rt, "hipExtMallocWithFlags"):
        return None
    with torch.cuda.device(device):
        for flag, kind in ((0x3, "uncached (hipDeviceMallocUncached)"), (0x1, "fine-grained (hipDeviceMallocFinegrained)")):
            p = ctypes.c_void_p()
            if rt.hipExtMallocWithFlags(ctypes.byref(p), ctypes.c_size_t(nbytes), ctypes.c_uint(flag)) == 0 and p.value:
                try:
                    t = torch.as_tensor(_RawBuffer(p.value, nbytes), device=device)
                    t.zero_()
                    torch.cuda.synchronize(device)
                    return t, kind, int(p.value)
                except Exception:   # noqa: BLE001  (no array-interface support in this build: give the memory back, use the allocator)
                    rt.hipFree(p)
                    return None
    return None


def _ipc_export(ptr: int) -> bytes | None:
    import ctypes
    rt = _hip_runtime()
    h = ctypes.create_string_buffer(64)   # hipIpcMemHandle_t
    if rt is None or rt.hipIpcGetMemHandle(h, ctypes.c_void_p(ptr)) != 0:
        return None
    return bytes(h.raw)


def _ipc_open(handle: bytes, nbytes: int, device: torch.device):
    import ctypes
    rt = _hip_runtime()
    p = ctypes.c_void_p()
    hb = ctypes.create_string_buffer(handle, 64)

    class _H(ctypes.Structure):
        _fields_ = [("reserved", ctypes.c_char * 64)]
    hv = _H.from_buffer_copy(hb.raw)
    rt.hipIpcOpenMemHandle.argtypes = [ctypes.POINTER(ctypes.c_void_p), _H, ctypes.c_uint]
    with torch.cuda.device(device):
        rc = rt.hipIpcOpenMemHandle(ctypes.byref(p), hv, ctypes.c_uint(1))   # hipIpcMemLazyEnablePeerAccess
    if rc != 0 or not p.value:
        raise RuntimeError(f"hipIpcOpenMemHandle failed (rc={rc})")
    return torch.as_tensor(_RawBuffer(p.value, nbytes), device=device)


class PeerExchange:
    """The exchange points of a column-sharded decode step over PEER MEMORY (csrc/exchange.hip, hqq_hip_exchange) instead of a
    collective library: one small kernel per point stores this rank's slices straight into every rank's full rows — in the reference's
    column order, so nothing is permuted afterwards — raises a flag per peer and waits for the others' flags.  One activation row.

    points: one list per exchange point with the FULL widths N of the layers exchanged together (a decoder block:
    [[Nq, Nk, Nv], [No], [Ngate, Nup], [Ndown]]).  Every rank builds the object with the same arguments; construction is collective
    (the arenas' IPC handles are all-gathered over `group`).  Consecutive run() calls must alternate between at least two points
    (the kernel's re-use rule: it protects the ROWS — a captured graph holding an odd sequence of points must not be replayed back to
    back; the flags carry generations and need no such care); run() enforces it for eager sequences.  The arena (flag lines, status, rows)
    is fine-grained / uncached device memory where the runtime offers it (`memory_kind`).  status() / check() report a wait that gave up;
    reset() (collective) clears it.

        px = PeerExchange(points, nbits, torch.float16, device)
        px.run(e, [y_q, y_k, y_v])      # y_*: this rank's [1, N/P] outputs of point e, local order
        x_next = px.full(e, 0)          # [1, Nq], complete once the kernel has finished (stream order)
    """

    ALIGN = 256

    @staticmethod
    def _release(raw_ptr, opened, device) -> None:
        import ctypes
        rt = _hip_runtime()
        if rt is None:
            return
        try:
            with torch.cuda.device(device):
                torch.cuda.synchronize(device)
                for p in opened:
                    rt.hipIpcCloseMemHandle(ctypes.c_void_p(p))
                if raw_ptr:
                    rt.hipFree(ctypes.c_void_p(raw_ptr))
        except Exception:   # noqa: BLE001  (interpreter shutdown: the driver reclaims the process's memory anyway)
            pass

    def close(self, group=None) -> None:
        """Give the arena and the peers' IPC mappings back to the runtime — the ONLY way they are released: garbage collection leaks them on purpose
        (a peer may still have this arena mapped and be storing into it, and full() hands out views into it; a rank that drops its object early, on
        an exception path say, must not turn the others' stores into writes to freed memory — advisor, round 5).  COLLECTIVE: every rank calls it; with a
        process group it first synchronises the device and barriers, so that no rank's kernels can still store into an arena when it goes.  The object —
        and every tensor full() / rows handed out — is unusable afterwards."""
        if self._closed:
            return
        self._closed = True
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        if self._collective:   # (built over a process group; local_group()'s in-process ranks share one caller and need no barrier)
            import torch.distributed as dist
            dist.barrier(group=self._group if group is None else group)
        PeerExchange._release(self._raw_ptr, list(self._opened), self.device)

    def __init__(self, points, nbits: int, dtype, device, group=None, spin_limit: int = 0, _arenas=None, _rank=None, _world=None, rows: int = 1):
        import torch.distributed as dist
        self.nbits, self.dtype, self.device = int(nbits), dtype, torch.device(device)
        self.rows = int(rows)   # activation rows the arena holds per layer (a decode batch of up to ops.EXCHANGE_MAX_ROWS rows; round 5)
        if not 1 <= self.rows <= ops.EXCHANGE_MAX_ROWS:
            raise ValueError(f"hqq_amd: PeerExchange serves 1..{ops.EXCHANGE_MAX_ROWS} activation rows")
        if _arenas is None:
            self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        else:
            self.world, self.rank = int(_world), int(_rank)
        if not 1 <= self.world <= ops.EXCHANGE_MAX_RANKS:
            raise ValueError(f"hqq_amd: PeerExchange serves 1..{ops.EXCHANGE_MAX_RANKS} ranks")
        per = 1 if nbits == 3 else ops.PER[nbits]
        self.points = [[int(n) for n in pt] for pt in points]
        for pt in self.points:
            if not 1 <= len(pt) <= ops.GEMV_MAX_GROUP:
                raise ValueError(f"hqq_amd: an exchange point holds 1..{ops.GEMV_MAX_GROUP} layers")
            for n in pt:
                if n % (per * self.world):
                    raise ValueError(f"hqq_amd: out_features={n} cannot be split over {self.world} ranks at {nbits} bits")
        if len(self.points) < 2:
            raise ValueError("hqq_amd: PeerExchange needs at least two exchange points (consecutive exchanges must alternate)")
        # arena (identical layout on every rank): [flag blocks: one 128-byte line per point | status line | rows, 256-byte aligned]
        self._flag_off = [128 * e for e in range(len(self.points))]
        self._status_off = 128 * len(self.points)
        off = self._status_off + 128
        self._row_off = []
        for pt in self.points:
            offs = []
            for n in pt:
                off = (off + self.ALIGN - 1) // self.ALIGN * self.ALIGN
                offs.append(off)
                off += 2 * n * self.rows
            self._row_off.append(offs)
        self.arena_bytes = (off + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        assert self.arena_bytes == self._layout_bytes(self.points, self.world, self.rows)
        self.spin_limit = int(spin_limit)
        self._last = None
        self._group = group
        self._collective = _arenas is None and self.world > 1
        self.memory_kind = "caller's tensors (single-process group)"
        self._raw_ptr = None
        if _arenas is not None:      # single-process group (tests): the ranks' arenas are plain tensors of this process
            self._arenas = list(_arenas)
        else:
            raw = _alloc_fine_grained(self.arena_bytes, self.device)
            use_raw = raw is not None
            if self.world > 1:   # every rank must take the same route (the handles are of different kinds)
                flags_ = [None] * self.world
                dist.all_gather_object(flags_, use_raw, group=group)
                use_raw = all(flags_)
            if use_raw:
                mine, self.memory_kind, self._raw_ptr = raw
            else:
                if raw is not None:
                    _hip_runtime().hipFree(__import__("ctypes").c_void_p(raw[2]))
                mine = torch.zeros(self.arena_bytes, dtype=torch.uint8, device=self.device)
                self.memory_kind = "coarse-grained (torch caching allocator: hipExtMallocWithFlags unavailable on some rank)"
            torch.cuda.synchronize(self.device)
            if self.world == 1:
                self._arenas = [mine]
            else:
                handles = [None] * self.world
                if use_raw:
                    dist.all_gather_object(handles, _ipc_export(self._raw_ptr), group=group)
                else:
                    from torch.multiprocessing.reductions import reduce_tensor
                    dist.all_gather_object(handles, reduce_tensor(mine), group=group)
                # (a process cannot open its own handle; peers' arenas are mapped into this process: stores to them travel over xGMI)
                err = None
                try:
                    if use_raw:
                        if any(h is None for h in handles):
                            raise RuntimeError("hipIpcGetMemHandle failed on some rank")
                        self._arenas = [mine if p == self.rank else _ipc_open(handles[p], self.arena_bytes, self.device) for p in range(self.world)]
                    else:
                        self._arenas = [mine if p == self.rank else handles[p][0](*handles[p][1]) for p in range(self.world)]
                except Exception as e:   # noqa: BLE001  (IPC not available between these devices / processes)
                    err = e
                # every rank reports; this is also the barrier: nobody stores into an arena before every rank has mapped all of them, and
                # a rank that could not map them does not leave the others waiting
                oks = [None] * self.world
                dist.all_gather_object(oks, err is None, group=group)
                if not all(oks):
                    bad = [p for p, o in enumerate(oks) if not o]
                    raise RuntimeError(f"hqq_amd: PeerExchange: rank(s) {bad} could not map their peers' arenas" + (f" ({type(err).__name__}: {err})" if err else ""))
        # what this object took from the runtime itself (the peers' IPC mappings, its own arena) goes back in close() only — never on garbage collection
        self._opened = [int(a.data_ptr()) for p, a in enumerate(self._arenas) if self._raw_ptr is not None and self.world > 1 and p != self.rank]
        self._closed = False
        if any(a.numel() < self.arena_bytes for a in self._arenas):
            raise ValueError("hqq_amd: PeerExchange arenas are smaller than the layout (ranks disagree about the points)")
        self._base = [a.data_ptr() for a in self._arenas]
        mine = self._arenas[self.rank]
        elems = {torch.float16: torch.float16, torch.bfloat16: torch.bfloat16}[dtype]
        self._rows = [[mine[o:o + 2 * n * self.rows].view(elems).view(self.rows, n) for o, n in zip(offs, pt)] for offs, pt in zip(self._row_off, self.points)]
        # per point: every rank's row addresses and flag-block address (what hqq_hip_exchange takes), computed once
        self._full_ptrs = [[[b + o for o in offs] for b in self._base] for offs in self._row_off]
        self._flag_ptrs = [[b + f for b in self._base] for f in self._flag_off]
        self._n_loc = [[n // self.world for n in pt] for pt in self.points]

    @classmethod
    def local_group(cls, points, nbits: int, dtype, device, world: int, spin_limit: int = 0, rows: int = 1):
        """`world` ranks inside ONE process (tests): same kernel and layout, the arenas are ordinary tensors.  One process cannot count on
        its ranks' kernels being resident together (streams share hardware queues, and a kernel would wait for one queued behind it):
        pass spin_limit=1 — the waits give up at once, the stores still happen — and clear flags and status between rounds, as
        tests/test_exchange_gpu.py does.  Real waits need one process per rank."""
        arenas = [torch.zeros(cls._layout_bytes(points, world, rows), dtype=torch.uint8, device=device) for _ in range(world)]
        return [cls(points, nbits, dtype, device, spin_limit=spin_limit, _arenas=arenas, _rank=r, _world=world, rows=rows) for r in range(world)]

    @classmethod
    def _layout_bytes(cls, points, world: int, rows: int = 1) -> int:
        off = 128 * len(points) + 128
        for pt in points:
            for n in pt:
                off = (off + cls.ALIGN - 1) // cls.ALIGN * cls.ALIGN + 2 * int(n) * int(rows)
        return (off + cls.ALIGN - 1) // cls.ALIGN * cls.ALIGN

    def full(self, e: int, j: int = 0, rows: int | None = None) -> Tensor:
        """this rank's full rows of layer j of point e — [rows, N] (default: all the arena holds), reference column order; complete after run(e, ...) in stream order"""
        return self._rows[e][j] if rows is None else self._rows[e][j][:rows]

    def status(self) -> int:
        """0, or 1 + the rank whose flag a wait gave up on (then the rows of that exchange are undefined)"""
        a = self._arenas[self.rank]
        return int(a[self._status_off:self._status_off + 4].view(torch.int32).item())

    def check(self) -> None:
        """raise if a wait of any earlier exchange gave up (synchronises); the rows of that exchange were undefined"""
        st = self.status()
        if st:
            raise RuntimeError(f"hqq_amd: PeerExchange: a wait for rank {st - 1} gave up (status {st}); outputs of that exchange were undefined — reset() before going on")

    def reset(self) -> None:
        """collective: every rank clears its flag words, launch tickets and status between two barriers (after a reported time-out, or to
        restart the generation count); no exchange may be in flight"""
        torch.cuda.synchronize(self.device)
        if self._collective:
            import torch.distributed as dist
            dist.barrier(group=self._group)
        self._arenas[self.rank][:self._status_off + 128].zero_()
        torch.cuda.synchronize(self.device)
        if self._collective:
            import torch.distributed as dist
            dist.barrier(group=self._group)
        self._last = None

    def run(self, e: int, y_loc) -> None:
        if self._last == e:
            raise RuntimeError("hqq_amd: consecutive exchanges must alternate between at least two points (csrc/exchange.hip re-use rule)")
        pt = self.points[e]
        if len(y_loc) != len(pt):
            raise ValueError(f"hqq_amd: exchange point {e} holds {len(pt)} layers, got {len(y_loc)}")
        if y_loc[0].numel() // self._n_loc[e][0] > self.rows:
            raise ValueError(f"hqq_amd: this PeerExchange holds {self.rows} activation rows per layer")
        ops.exchange(list(y_loc), self._n_loc[e], self.nbits, self.world, self.rank, self._full_ptrs[e], self._flag_ptrs[e],
                     self._base[self.rank] + self._status_off, self.spin_limit)
        self._last = e   # (only an exchange that was enqueued counts)
