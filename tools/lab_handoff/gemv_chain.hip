// gemv_chain.hip — the decode launch of gemv.hip as a link of a CHAIN of dependent launches that overlap (gfx950, bs <= 4).
//
// Replaces, like gemv.hip, the reference's per-call chain of an axis=1 HQQLinear at a few activation rows
//   BitPack.unpack_* -> (W_r - zero) * scale -> torch.matmul(x, W.t()) (+ bias)
//   (hqq/core/bitpack.py:31-64, hqq/core/quantize.py:183-199, :880-898)
// for the case the decode loop is made of: launch s + 1 reads the activation row launch s writes (q|k|v -> o -> gate|up -> down -> ...,
// hqq/utils/generation_hf.py:117-540 issues them one after the other).  Same packed layout, same exact weights
// round16(round16(q - z) * s), same MFMA contraction in the same order as gemv.hip: bit-identical outputs.
//
// Why.  A dependent launch of the row-per-wave kernel costs ~3.4 us on top of its bytes (measured with in-kernel time stamps, DESIGN.md
// section 3.1: hardware boundary 1.25 + arguments / first request 0.7 + pipeline fill 1.0 + tail 0.4) and its ~4 us of VALU work — the exact
// weight rebuild — can only start once x is there.  Neither depends on x.  So consecutive launches go to TWO streams (even / odd) and
// launch s + 1 starts while launch s is still streaming: it requests its first units, REBUILDS them to fp16 MFMA operands in registers
// (the x-independent 52 of the kernel's ~90 VALU instructions per KiB), requests the next units, and only then waits — in the kernel — for
// launch s to publish its outputs.  When x arrives the waiting launch has its first 4 units per wave (16 MB chip-wide at 8 waves per CU:
// half of an average 7B launch, all of a 4096 x 4096 layer) either rebuilt or in flight, and the predecessor's boundary, prologue,
// fill and tail have been spent under somebody else's stream.
//
// Measured (round 3): correct and bit-identical, the overlap works as described — and the 7B stack takes 1.28-1.35 ms
// per token against 1.00-1.06 for the stream-ordered launches, because the in-kernel hand-off below costs 4-6 us where a kernel boundary
// costs 3.4 (store acknowledgement, arrival, poll, x: each a trip across the fabric).  An opt-in entry point, not the default path.
//
// Protocol (MI355X_MICROARCH.md "inter-workgroup visibility", recipe R1: write-through payload, drained, then the flag)
//   producer  every output is stored write-through (sc1); a wave that has stored its last row drains (s_waitcnt vmcnt(0)) and takes
//             a ticket in LDS; the workgroup's last wave adds 1 to one of GC_SHARDS device-scope arrival counters (one 128-byte line each)
//   consumer  wave 0 of the first 16 workgroups polls the counters relaxed (sc1 loads, s_sleep between polls) until their sum reaches the
//             producer's workgroup count and raises 16 copies of a flag; wave 0 of every other workgroup polls one copy (all 512
//             workgroups polling the eight counter lines saturated those lines' memory channels: 7 us per hand-off); the other waves
//             sleep at an LDS-only barrier (their weight loads stay in flight); then all waves read x with sc1 loads (they bypass
//             this CU's L1) into LDS
//   bounded   a poll loop gives up after `spin_limit` rounds, writes the status word and runs on without waiting (wrong results, reported;
//             never a hang)
//   residency at most two links are alive at any time (link s + 2 is stream-ordered behind link s), and a link places at most HALF of what
//             the occupancy query admits per CU, so every workgroup of both is resident whatever the dispatch order: a waiting
//             workgroup can never keep a producing one off the chip
//
// HBM-bandwidth bound; algorithmic bytes are those of gemv.hip (0.5625 B/param at 4-bit).
#include "gemv_shared.h"

namespace hqq {

constexpr int GC_SHARDS = 8;              // arrival counters of one link, one 128-byte line each
constexpr int GC_FLAGS = 16;              // copies of the link's "complete" flag behind the counters, one 128-byte line each
constexpr int GC_RELAYS = 16;             // workgroups of the waiting launch that watch the counters and raise the flags
constexpr int GC_WG_PER_CU = 2;           // workgroups (of GV_WAVES waves) a link places per CU
static_assert((GC_SHARDS + GC_FLAGS) * 128 <= HQQ_CHAIN_COUNTER_BYTES, "counter block");

struct GvChain {
  const uint32_t* wait;    // the predecessor's arrival counters, or null: x is complete when the launch starts
  uint32_t* signal;        // this launch's arrival counters, or null
  uint32_t* status;        // time-out word
  uint32_t wait_total;     // arrivals that complete the predecessor (its workgroups)
  uint32_t spin_limit;
#ifdef GC_LAB_TS
  unsigned long long* ts;  // lab: [wave][8] time stamps
#endif
};

#define GC_GLOBAL __attribute__((address_space(1)))

__device__ __forceinline__ void gc_lds_barrier() {
  // workgroup barrier that orders LDS only: __syncthreads() would also wait for every weight load in flight (vmcnt(0))
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// NPRE: units per wave rebuilt to MFMA operands before x is there (16 VGPRs per unit and slab)
template <int NBITS, int M, bool SUB, int NPRE>
__global__ __launch_bounds__(GV_WAVES * 64, 4) void gemv_chain_kernel(GV_IN_PARAMS, const GvOut o, const GvChain ch) {
  const GvIn a = GV_IN_PACK;
  constexpr int PER = 8 / NBITS;
  static_assert(NPRE >= 0 && NPRE <= 2, "units rebuilt ahead");
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  u32x4* xs = reinterpret_cast<u32x4*>(smem);   // [M][K/1024 (padded)][2 planes][64 lanes] x 16 B

#ifdef GC_LAB_TS
  unsigned long long t_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  t_[0] = __builtin_amdgcn_s_memrealtime();
#define GC_TS(i) t_[i] = __builtin_amdgcn_s_memrealtime();
#else
#define GC_TS(i)
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = a.K, G = a.G;
  const int nsteps = (K + GV_KSTEP - 1) / GV_KSTEP;
  const int nunits = (nsteps + GV_U - 1) / GV_U;
  const int planes_per_m = nsteps * 2 * 64;
  const int stride = gridDim.x * GV_WAVES;
  const int total = a.total_prow;
  uint32_t* tickets = reinterpret_cast<uint32_t*>(smem + a.red_off);   // waves of this workgroup that have stored and drained their last row
  if (tid == 0) *tickets = 0;   // (ordered before any ticket by the barriers in front of the x staging)

  // identical to gemv.hip's issue(), group_size 64: GV_U weight loads + 2 PER meta loads per call, valid or not
  auto issue = [&](Unit<PER, true>& un, const LayerCtx& c, int prow, int unit, bool live) {
    const int p = live ? prow - c.row0 : 0;
    const int rows_per_slab = live ? c.N / PER : 0;
    const int Glive = live ? G : 0, Klive = live ? K : 0;
    const __amdgpu_buffer_rsrc_t rw = buffer_rsrc(c.Wq), rz = buffer_rsrc(c.zero), rs = buffer_rsrc(c.scale);
    int g = unit * (GV_UNIT / 64) + lane;
    g = g < Glive ? g : 0;
#pragma unroll
    for (int s = 0; s < PER; ++s) {
      const uint32_t row_off = static_cast<uint32_t>(p + s * rows_per_slab) * static_cast<uint32_t>(G) * 2u;
      un.z[s] = __builtin_amdgcn_raw_buffer_load_b16(rz, g * 2, row_off, 0);
      un.sc[s] = __builtin_amdgcn_raw_buffer_load_b16(rs, g * 2, row_off, 0);
    }
    const uint32_t wrow_off = static_cast<uint32_t>(p) * static_cast<uint32_t>(K);
#pragma unroll
    for (int u = 0; u < GV_U; ++u) {
      int k0 = unit * GV_UNIT + u * GV_KSTEP + lane * 16;
      k0 = k0 < Klive ? k0 : 0;
      un.w[u] = __builtin_amdgcn_raw_buffer_load_b128(rw, k0, wrow_off, 2 /* nt: streamed once */);
    }
  };
  auto advance = [&](int& p, int& u, LayerCtx& c) {
    u += 1;
    if (u >= nunits) {
      u = 0;
      p += stride;
      if (p >= c.end && p < total) c = select_layer(a, p);
    }
  };

  uint32_t magic;
  asm volatile("v_mov_b32 %0, 0x64006400" : "=v"(magic));
  f32x4 acc[M][PER];
#pragma unroll
  for (int m = 0; m < M; ++m)
#pragma unroll
    for (int s = 0; s < PER; ++s) acc[m][s] = f32x4{0.f, 0.f, 0.f, 0.f};

  struct Rebuilt { h8_t a[GV_U][PER][2]; };
  // a unit's packed weights -> the MFMA A operands of every (load, slab): what SlabExact does in front of its MFMAs, nothing of x in it
  auto rebuild = [&](const Unit<PER, true>& cur, Rebuilt& r) {
#pragma unroll
    for (int u = 0; u < GV_U; ++u) {
      uint32_t zs[PER];
#pragma unroll
      for (int s = 0; s < PER; ++s) {
        uint32_t mine = cur.z[s] | (cur.sc[s] << 16);
        if constexpr (SUB) mine = scale_meta_sub<NBITS>(mine, s);
        zs[s] = __builtin_amdgcn_ds_bpermute((u * 16 + (lane >> 2)) << 2, mine);
      }
      SlabRebuild<NBITS, 0, PER, SUB>::run(cur.w[u], zs, r.a[u], magic);
    }
  };
  // the rebuilt operands exist HERE: left alone the compiler sinks the whole rebuild to its use behind the wait for x (pure arithmetic,
  // fewer live registers) and nothing is done ahead
  auto pin = [&](Rebuilt& r) {
#pragma unroll
    for (int u = 0; u < GV_U; ++u)
#pragma unroll
      for (int s = 0; s < PER; ++s) {
        asm volatile("" : "+v"(r.a[u][s][0]));
        asm volatile("" : "+v"(r.a[u][s][1]));
      }
  };
  // a finished row: one reduction per output row, lane (m * PER + s) stores its value WRITE-THROUGH
  auto finish_row = [&](int prow) {
    const OutCtx oc = select_out(a, o, prow);
    const int p = prow - oc.row0;
    const int rows_per_slab = oc.N / PER;
    float mine = 0.f;
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
      for (int s = 0; s < PER; ++s) {
        const float v = diag_sum(acc[m][s]);
        acc[m][s] = f32x4{0.f, 0.f, 0.f, 0.f};
        mine = (lane == m * PER + s) ? v : mine;
      }
    if (lane < M * PER) {
      const int m = lane / PER, s = lane - m * PER;
      const int n = p + s * rows_per_slab;
      half_t ov = static_cast<half_t>(mine);
      if (oc.bias) ov = ov + oc.bias[n];   // `out += bias` on the rounded matmul result (quantize.py:896-897)
      __hip_atomic_store((uint16_t GC_GLOBAL*)(oc.y) + (static_cast<int64_t>(m) * oc.N + n), __builtin_bit_cast(uint16_t, ov),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  // rebuilt operands against x from LDS — the MFMAs of SlabExact in SlabExact's order (per load: slab by slab, k 0..7 then k 8..15)
  auto contract = [&](const Rebuilt& r, int prow, int unit) {
#pragma unroll
    for (int u = 0; u < GV_U; ++u) {
      const int step = unit * GV_U + u;
      if (step < nsteps) {
        h8_t b0[M], b1[M];
#pragma unroll
        for (int m = 0; m < M; ++m) {
          b0[m] = __builtin_bit_cast(h8_t, xs[m * planes_per_m + (step * 2 + 0) * 64 + lane]);
          b1[m] = __builtin_bit_cast(h8_t, xs[m * planes_per_m + (step * 2 + 1) * 64 + lane]);
        }
#pragma unroll
        for (int s = 0; s < PER; ++s)
#pragma unroll
          for (int m = 0; m < M; ++m) {
            acc[m][s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(r.a[u][s][0], b0[m], acc[m][s], 0, 0, 0);
            acc[m][s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(r.a[u][s][1], b1[m], acc[m][s], 0, 0, 0);
          }
      }
    }
    if (unit + 1 >= nunits) finish_row(prow);
  };
  // in the streaming loop: gemv.hip's consume() — rebuild and contract slab by slab (SlabExact: 8 operand registers live at a time)
  auto consume = [&](const Unit<PER, true>& cur, int prow, int unit) {
#pragma unroll
    for (int u = 0; u < GV_U; ++u) {
      const int step = unit * GV_U + u;
      uint32_t zs[PER];
#pragma unroll
      for (int s = 0; s < PER; ++s) {
        uint32_t mine = cur.z[s] | (cur.sc[s] << 16);
        if constexpr (SUB) mine = scale_meta_sub<NBITS>(mine, s);
        zs[s] = __builtin_amdgcn_ds_bpermute((u * 16 + (lane >> 2)) << 2, mine);
      }
      if (step < nsteps) {
        h8_t b0[M], b1[M];
#pragma unroll
        for (int m = 0; m < M; ++m) {
          b0[m] = __builtin_bit_cast(h8_t, xs[m * planes_per_m + (step * 2 + 0) * 64 + lane]);
          b1[m] = __builtin_bit_cast(h8_t, xs[m * planes_per_m + (step * 2 + 1) * 64 + lane]);
        }
        SlabExact<NBITS, M, 0, PER, SUB>::run(cur.w[u], zs, b0, b1, acc, magic);
      }
    }
    if (unit + 1 >= nunits) finish_row(prow);
  };

  // ---- before x: request, rebuild, request ----
  int p0 = blockIdx.x * GV_WAVES + wave, u0 = 0;
  const bool live0 = p0 < total;
  p0 = live0 ? p0 : total - 1;   // waves with no row at all still request (load counts stay uniform)
  LayerCtx lc = select_layer(a, p0);
  Unit<PER, true> ua, ub;
  issue(ua, lc, p0, u0, live0);
  int p1 = p0, u1 = u0;
  advance(p1, u1, lc);
  const bool live1 = live0 && p1 < total;
  issue(ub, lc, live1 ? p1 : p0, u1, live1);
  GC_TS(1)
  // cursor: the position behind the youngest requested unit
  int pc = p1, uc = u1;
  Rebuilt ra, rb;
  bool la = live0, lb = live1;      // is the unit held in ua / ub live (to be consumed)?
  int pa = p0, qa = u0, pb = p1, qb = u1;   // and which (row, unit) it is
  if constexpr (NPRE >= 1) {
    rebuild(ua, ra);
    pin(ra);
    advance(pc, uc, lc);
    la = live1 && pc < total;
    pa = pc; qa = uc;
    issue(ua, lc, la ? pc : p0, uc, la);
  }
  if constexpr (NPRE >= 2) {
    rebuild(ub, rb);
    pin(rb);
    advance(pc, uc, lc);
    lb = la && pc < total;
    pb = pc; qb = uc;
    issue(ub, lc, lb ? pc : p0, uc, lb);
  }
  GC_TS(2)

  // ---- the predecessor's outputs.  Two levels, so that nobody's polls share a memory channel with the arrivals they wait for: the
  //      first GC_RELAYS workgroups poll the arrival counters (few readers on the lines the producers' atomics go to) and raise
  //      GC_FLAGS copies of a flag, one 128-byte line each; every other workgroup polls one copy (512 workgroups polling the eight
  //      counter lines themselves saturate those lines' channels: measured 7 us from the last arrival to the first reader, and the
  //      producer's own weight rows on those channels queue behind the polls) ----
  if (ch.wait != nullptr) {
    if (wave == 0) {
      const uint32_t GC_GLOBAL* w = (const uint32_t GC_GLOBAL*)ch.wait;
      const bool relay = blockIdx.x < GC_RELAYS;
      bool gave_up = false;
      if (relay) {
        for (uint32_t spins = 0;; ++spins) {
          uint32_t v = 0;
          if (lane < GC_SHARDS) v = __hip_atomic_load(w + lane * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          uint32_t sum = 0;
#pragma unroll
          for (int i = 0; i < GC_SHARDS; ++i) sum += static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), i));
          if (sum >= ch.wait_total) break;
          if (spins >= ch.spin_limit) { gave_up = true; break; }
          __builtin_amdgcn_s_sleep(1);
        }
        // (every relay raises every copy: the first to see the counters complete is the one the others profit from)
        if (lane < GC_FLAGS) __hip_atomic_store((uint32_t GC_GLOBAL*)w + (GC_SHARDS + lane) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        const uint32_t GC_GLOBAL* f = w + (GC_SHARDS + (blockIdx.x % GC_FLAGS)) * 32;
        for (uint32_t spins = 0;; ++spins) {
          uint32_t v = 0;
          if (lane == 0) v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (__builtin_amdgcn_readfirstlane(static_cast<int>(v)) != 0) break;
          if (spins >= ch.spin_limit) { gave_up = true; break; }
          __builtin_amdgcn_s_sleep(4);
        }
      }
      if (gave_up && lane == 0) __hip_atomic_store((uint32_t GC_GLOBAL*)ch.status, 1u + blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  gc_lds_barrier();
  GC_TS(3)

  // ---- x -> LDS in the order the weight rebuild produces values; sc1 loads: they bypass this CU's L1, which may hold the row a
  //      previous token left at the same address; reads past K return 0 (buffer bounds) ----
  {
    const int chunks_per_m = nsteps * 64;
    constexpr int XB = 3;   // chunks per thread and batch: all of a batch's loads are out before the first is written to LDS
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(a.x + static_cast<int64_t>(m) * K), 0, K * 2, 0x00020000);
      for (int j0 = tid; j0 < chunks_per_m; j0 += XB * GV_WAVES * 64) {
        u32x4 v[XB][2];
#pragma unroll
        for (int b = 0; b < XB; ++b) {
          const int j = j0 + b * GV_WAVES * 64;
          v[b][0] = __builtin_amdgcn_raw_buffer_load_b128(rx, j * 32, 0, 16 /* sc1 */);
          v[b][1] = __builtin_amdgcn_raw_buffer_load_b128(rx, j * 32 + 16, 0, 16);
        }
#pragma unroll
        for (int b = 0; b < XB; ++b) {
          const int j = j0 + b * GV_WAVES * 64;
          if (j < chunks_per_m) {
            const int it = j >> 6, ln = j & 63;
            xs[m * planes_per_m + (it * 2 + 0) * 64 + ln] = permute_x8(v[b][0]);
            xs[m * planes_per_m + (it * 2 + 1) * 64 + ln] = permute_x8(v[b][1]);
          }
        }
      }
    }
  }
  gc_lds_barrier();
  GC_TS(4)

  // ---- the units rebuilt ahead meet x ----
  if constexpr (NPRE >= 1) { if (live0) contract(ra, p0, u0); }
  if constexpr (NPRE >= 2) { if (live1) contract(rb, p1, u1); }
  GC_TS(5)

  // ---- two units in flight: consume the older, request the one behind the younger into its registers.  One loop shape, one exit at
  //      the bottom; a unit past the wave's last one is still requested (one cache line) and not consumed, so that every wait is an
  //      exact count.  With NPRE == 1 the older unit sits in ub. ----
  auto stream = [&](Unit<PER, true>& first, Unit<PER, true>& second, bool lf, int pf, int qf, bool ls, int ps, int qs) {
    bool more;
    do {
      if (lf) consume(first, pf, qf);
      advance(pc, uc, lc);
      lf = ls && pc < total;
      pf = pc; qf = uc;
      issue(first, lc, lf ? pc : p0, uc, lf);
      if (ls) consume(second, ps, qs);
      advance(pc, uc, lc);
      ls = lf && pc < total;
      ps = pc; qs = uc;
      issue(second, lc, ls ? pc : p0, uc, ls);
      more = lf;   // (ls implies lf)
    } while (more);
  };
  if constexpr (NPRE == 1) stream(ub, ua, lb, pb, qb, la, pa, qa);
  else stream(ua, ub, la, pa, qa, lb, pb, qb);
  GC_TS(6)

  // ---- publish: this wave's rows are stored and acknowledged; the workgroup's last wave arrives ----
  if (ch.signal != nullptr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint32_t old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(tickets, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    old = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(old)));
    if (old == GV_WAVES - 1 && lane == 0)
      __hip_atomic_fetch_add((uint32_t GC_GLOBAL*)ch.signal + (blockIdx.x % GC_SHARDS) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#ifdef GC_LAB_TS
  GC_TS(7)
  if (lane == 0 && ch.ts) { const int wg = blockIdx.x * GV_WAVES + wave; for (int i = 0; i < 8; ++i) ch.ts[wg * 8 + i] = t_[i]; }
#endif
}

static int gc_num_cus() {
  static int n_cus = 0;
  if (n_cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) n_cus = n;
    else { (void)hipGetLastError(); n_cus = 256; }
  }
  return n_cus;
}

#ifdef GC_LAB_TS
unsigned long long* g_gc_lab_ts = nullptr;   // lab: [launch][wave][8] time stamps, 4096 waves per launch slot
int g_gc_lab_idx = 0;
#endif

template <int NBITS, int M, bool SUB>
static int launch_chain(const GvArgs& args, GvChain ch, uint32_t* arrivals, hipStream_t st) {
  constexpr int PER = 8 / NBITS;
#ifdef GC_NPRE
  constexpr int NPRE = GC_NPRE;
#else
  constexpr int NPRE = (PER * M <= 2) ? 2 : ((PER * M <= 4) ? 1 : 0);   // 32 VGPRs per unit at 4 bits / one row; budget: 128 VGPRs per wave
#endif
  GvArgs a = args;
  const int nsteps = (a.K + GV_KSTEP - 1) / GV_KSTEP;
  const size_t xs_bytes = static_cast<size_t>(M) * nsteps * GV_KSTEP * 2;
  a.red_off = static_cast<int>((xs_bytes + 15) & ~static_cast<size_t>(15));
  const size_t lds = a.red_off + 16;
  if (lds > static_cast<size_t>(GV_LDS_MAX)) { set_error("hqq_hip_gemv_chained: K=%d too long to stage %d rows of x in LDS", a.K, M); return HQQ_ERR_UNSUPPORTED; }
  auto kern = gemv_chain_kernel<NBITS, M, SUB, NPRE>;
  if (lds > 64 * 1024) {
    static LdsRaised raised;   // per instantiation (and device)
    if (const int rc = raise_lds_limit(raised, reinterpret_cast<const void*>(kern), GV_LDS_MAX, "hqq_hip_gemv_chained")) return rc;
  }
  // Two links are co-resident: place at most half of what a CU admits of this kernel (registers, LDS).  Cached per instantiation for the
  // LDS size last asked about (a host query, no stream operation).
  static size_t occ_lds = ~static_cast<size_t>(0);
  static int occ = 0;
  if (occ_lds != lds) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(kern), GV_WAVES * 64, lds) != hipSuccess) { (void)hipGetLastError(); n = 0; }
    occ = n;
    occ_lds = lds;
  }
  int per_cu = occ / 2;
  per_cu = per_cu > GC_WG_PER_CU ? GC_WG_PER_CU : per_cu;
  if (per_cu < 1) { set_error("hqq_hip_gemv_chained: a CU admits %d workgroup(s) of this launch; two co-resident links need 2", occ); return HQQ_ERR_UNSUPPORTED; }
  const int tiles = (a.total_prow + GV_WAVES - 1) / GV_WAVES;
  const int cap = gc_num_cus() * per_cu;
  const int grid = tiles < cap ? tiles : cap;
  GvIn in;
  GvOut out;
  for (int i = 0; i < GV_MAXL; ++i) {
    in.Wq[i] = a.Wq[i]; in.scale[i] = a.scale[i]; in.zero[i] = a.zero[i]; in.N[i] = a.N[i]; in.prow_end[i] = a.prow_end[i];
    out.bias[i] = a.bias[i]; out.y[i] = a.y[i];
  }
  in.x = a.x; in.K = a.K; in.gs = a.gs; in.G = a.G; in.total_prow = a.total_prow; in.red_off = a.red_off; in.ksplit = 0;
#ifdef GV_LAB_TS
  in.ts = nullptr;
#endif
#ifdef GC_LAB_TS
  ch.ts = g_gc_lab_ts ? g_gc_lab_ts + static_cast<size_t>(g_gc_lab_idx++) * 4096 * 8 : nullptr;
#endif
  hipLaunchKernelGGL(kern, dim3(grid), dim3(GV_WAVES * 64), lds, st, GV_IN_ARGS(in), out, ch);
  if (arrivals) *arrivals = static_cast<uint32_t>(grid);
  return check_launch("hqq_hip_gemv_chained");
}

template <int NBITS, bool SUB>
static int dispatch_chain_m(int M, const GvArgs& a, const GvChain& ch, uint32_t* arrivals, hipStream_t st) {
  switch (M) {
    case 1: return launch_chain<NBITS, 1, SUB>(a, ch, arrivals, st);
    case 2: return launch_chain<NBITS, 2, SUB>(a, ch, arrivals, st);
    case 3: return launch_chain<NBITS, 3, SUB>(a, ch, arrivals, st);
    case 4: return launch_chain<NBITS, 4, SUB>(a, ch, arrivals, st);
  }
  return HQQ_ERR_SHAPE;
}

}  // namespace hqq

using namespace hqq;

#ifdef GC_LAB_TS
extern "C" void hqq_hip_lab_set_chain_ts(unsigned long long* p) { g_gc_lab_ts = p; g_gc_lab_idx = 0; }
#endif

extern "C" int hqq_hip_gemv_chained(int nbits, int n_layers, const void* x, const void* const* Wq, const void* const* scale,
                                    const void* const* zero, const void* const* bias, void* const* y, const int64_t* N,
                                    int64_t M, int64_t K, int64_t group_size, int dtype, uint32_t opts, const hqq_hip_chain_link* link,
                                    uint32_t* arrivals, void* stream) {
  clear_stale_error();
  if (opts & ~static_cast<uint32_t>(HQQ_OPT_META_SCALABLE)) { set_error("hqq_hip_gemv_chained: option bits 0x%x (exact weights only: 0 or HQQ_OPT_META_SCALABLE)", opts); return HQQ_ERR_UNSUPPORTED; }
  if (n_layers < 1 || n_layers > HQQ_GEMV_MAX_GROUP) { set_error("hqq_hip_gemv_chained: n_layers=%d outside [1,%d]", n_layers, HQQ_GEMV_MAX_GROUP); return HQQ_ERR_SHAPE; }
  if (!link || !x || !Wq || !scale || !zero || !y || !N) { set_error("hqq_hip_gemv_chained: null argument"); return HQQ_ERR_SHAPE; }
  if (nbits != 8 && nbits != 4 && nbits != 2) { set_error("hqq_hip_gemv_chained: nbits=%d not covered (8 / 4 / 2)", nbits); return HQQ_ERR_UNSUPPORTED; }
  if (dtype != HQQ_F16 || group_size != 64) { set_error("hqq_hip_gemv_chained: covers fp16, group_size 64 (got dtype %d, gs %lld)", dtype, (long long)group_size); return HQQ_ERR_UNSUPPORTED; }
  if (M < 1 || M > GV_EXACT_ROWWISE_MAX_M) { set_error("hqq_hip_gemv_chained: M=%lld outside [1,%d]", (long long)M, GV_EXACT_ROWWISE_MAX_M); return HQQ_ERR_SHAPE; }
  if (K <= 0 || K % 64 || K > INT32_MAX / 2) { set_error("hqq_hip_gemv_chained: bad K"); return HQQ_ERR_SHAPE; }
  if (!aligned16(x)) { set_error("hqq_hip_gemv_chained: pointers must be 16-byte aligned"); return HQQ_ERR_ALIGN; }
  if ((link->wait && !link->status) || (link->wait && link->wait_arrivals == 0)) { set_error("hqq_hip_gemv_chained: a waiting link needs a status word and the predecessor's arrival count"); return HQQ_ERR_SHAPE; }
  if ((reinterpret_cast<uintptr_t>(link->wait) | reinterpret_cast<uintptr_t>(link->signal)) & 127u) { set_error("hqq_hip_gemv_chained: arrival counters must be 128-byte aligned"); return HQQ_ERR_ALIGN; }
  const int per = 8 / nbits;
  GvArgs a;
  int64_t total = 0;
  for (int i = 0; i < n_layers; ++i) {
    if (N[i] <= 0 || N[i] % per) { set_error("hqq_hip_gemv_chained: needs N %% %d == 0 (got N=%lld)", per, (long long)N[i]); return N[i] <= 0 ? HQQ_ERR_SHAPE : HQQ_ERR_UNSUPPORTED; }
    if (N[i] * (K / group_size) > INT32_MAX || (N[i] / per) * K > static_cast<int64_t>(UINT32_MAX)) { set_error("hqq_hip_gemv_chained: size overflow"); return HQQ_ERR_SHAPE; }
    if (!Wq[i] || !scale[i] || !zero[i] || !y[i]) { set_error("hqq_hip_gemv_chained: null layer pointer"); return HQQ_ERR_SHAPE; }
    if (!aligned16(Wq[i])) { set_error("hqq_hip_gemv_chained: pointers must be 16-byte aligned"); return HQQ_ERR_ALIGN; }
    total += N[i] / per;
    if (total > INT32_MAX) { set_error("hqq_hip_gemv_chained: size overflow"); return HQQ_ERR_SHAPE; }
    a.Wq[i] = static_cast<const uint8_t*>(Wq[i]);
    a.scale[i] = static_cast<const half_t*>(scale[i]);
    a.zero[i] = static_cast<const half_t*>(zero[i]);
    a.bias[i] = bias ? static_cast<const half_t*>(bias[i]) : nullptr;
    a.y[i] = static_cast<half_t*>(y[i]);
    a.N[i] = static_cast<int>(N[i]);
    a.prow_end[i] = static_cast<int>(total);
  }
  for (int i = n_layers; i < GV_MAXL; ++i) {
    a.Wq[i] = a.Wq[n_layers - 1]; a.scale[i] = a.scale[n_layers - 1]; a.zero[i] = a.zero[n_layers - 1]; a.bias[i] = a.bias[n_layers - 1];
    a.y[i] = a.y[n_layers - 1]; a.N[i] = a.N[n_layers - 1]; a.prow_end[i] = a.prow_end[n_layers - 1];
  }
  a.x = static_cast<const half_t*>(x);
  a.K = static_cast<int>(K);
  a.gs = 64;
  a.G = static_cast<int>(K / 64);
  a.total_prow = static_cast<int>(total);
  a.red_off = 0;
  a.ksplit = 0;
#ifdef GV_LAB_TS
  a.ts = nullptr;
#endif
  GvChain ch;
  ch.wait = static_cast<const uint32_t*>(link->wait);
  ch.signal = static_cast<uint32_t*>(link->signal);
  ch.status = static_cast<uint32_t*>(link->status);
  ch.wait_total = link->wait_arrivals;
  ch.spin_limit = link->spin_limit ? link->spin_limit : (1u << 16);
#ifdef GC_LAB_TS
  ch.ts = nullptr;
#endif
  hipStream_t st = as_stream(stream);
  const bool sub = (opts & HQQ_OPT_META_SCALABLE) != 0;
  const int mm = static_cast<int>(M);
  switch (nbits) {
    case 8: return sub ? dispatch_chain_m<8, true>(mm, a, ch, arrivals, st) : dispatch_chain_m<8, false>(mm, a, ch, arrivals, st);
    case 4: return sub ? dispatch_chain_m<4, true>(mm, a, ch, arrivals, st) : dispatch_chain_m<4, false>(mm, a, ch, arrivals, st);
    case 2: return sub ? dispatch_chain_m<2, true>(mm, a, ch, arrivals, st) : dispatch_chain_m<2, false>(mm, a, ch, arrivals, st);
  }
  return HQQ_ERR_NBITS;
}
