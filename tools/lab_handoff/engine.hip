// engine.hip — the persistent cross-layer decode engine (gfx950): ONE launch walks a whole token's list of dependent linear
// stages (q|k|v -> o -> gate|up -> down -> next block ...), bs = 1.
//
// Replaces, for a chain of axis=1 HQQLinear layers at one activation row, the per-layer call chain
//   BitPack.unpack_* -> (W_r - zero) * scale -> torch.matmul(x, W.t()) (+ bias)      [per layer, per token]
//   (hqq/core/bitpack.py:31-64, hqq/core/quantize.py:183-199, :880-898; the decode loop that issues them one after the other is
//   hqq/utils/generation_hf.py:117-540)
// i.e. what gemv.hip does per launch — same packed layout, same exact weights round16(round16(q - z) * s), same MFMA contraction —
// but without the ~4 us every dependent launch costs (boundary, kernarg fetch, prologue, first-byte latency, tail): measured in
// round 1, 128 launches x 4.2 us were 0.5 ms of a 1.2 ms token.
//
// Shape of the engine
//   grid     one workgroup of 16 waves per CU, all co-resident (persistent).  Waves 0..14 stream weights, wave 15 is the
//            control wave (it alone touches the inter-workgroup protocol, so no streaming wave ever drains its loads).
//   stage    a group of <= 4 layers that read the same x (exactly a gemv.hip launch).  Its packed rows — one concatenated row
//            space — are cut into contiguous chunks, one per workgroup; the chunk's `steps` (1 KiB of one packed row = 1024 k,
//            one global_load_dwordx4 per lane) are cut into 15 contiguous ranges, one per streaming wave: every wave of the chip
//            gets the same number of bytes (+-1 step) whatever the layer shape (the row-per-wave grid of gemv.hip hands a wave
//            1 or 2 rows of q|k|v: 75 % balance).
//   units    each streaming wave keeps two units of 4 steps (2 x 4 KiB) in flight in two register sets and requests the unit after
//            next as soon as it has consumed one (the ping-pong of gemv.hip; an LDS-DMA ring was tried first: the DMA issue
//            path tops out near 4.5 TB/s with one 256-byte and one 1-KiB piece per step, tools/engine_ts.py).  The
//            issue cursor runs ahead of the consume cursor ACROSS stage boundaries: weights and group constants do not depend
//            on x, so while a stage's results are published and the next x is awaited the HBM pipe keeps serving the next
//            stage's first 120 KiB per CU (30 MB chip-wide).
//   partials a wave leaves one 16-float partial per (row, slab) it touched in LDS; after the workgroup barrier 16 lanes per
//            output add the partials of a row in wave order (fixed order: reproducible), round once to fp16 (+ bias).
//   hand-off (MI355X_MICROARCH.md, inter-workgroup visibility) y is stored write-through (sc1) by the control wave, which then
//            drains (s_waitcnt vmcnt(0)) and adds 1 to one of 8 sharded arrival counters; the control waves of all workgroups
//            poll the 8 counters relaxed, then read the next x with sc1 loads into LDS.  Placement-independent; every spin is
//            bounded (a time-out word is set and the launch runs to its end without waiting).
//
// HBM-bandwidth bound: the algorithmic bytes are those of gemv.hip (0.5625 B/param at 4-bit).
#include <string.h>

#include "decode_common.h"

namespace hqq {

constexpr int EN_MAXL = HQQ_GEMV_MAX_GROUP;
constexpr int EN_WAVES = 16;
constexpr int EN_NSW = EN_WAVES - 1;   // streaming waves
constexpr int EN_LDS_MAX = 160 * 1024;
constexpr int EN_STEP = 1024;          // k per step
#ifndef EN_UNIT
#define EN_UNIT 4
#endif
constexpr int EN_U = EN_UNIT;          // steps per unit; two units (2 x EN_U KiB per wave, 120 KiB per CU at 4) are in flight
constexpr int EN_SHARDS = 8;           // arrival counters (one 128-byte line each)
constexpr int EN_SYNC_WORDS = (EN_SHARDS + 1) * 32;   // + the time-out word
constexpr uint32_t EN_SPIN_LIMIT = 1u << 21;

struct alignas(256) EnStage {
  const uint8_t* Wq[EN_MAXL];
  const half_t* scale[EN_MAXL];
  const half_t* zero[EN_MAXL];
  const half_t* bias[EN_MAXL];
  half_t* y[EN_MAXL];
  const half_t* x;
  int N[EN_MAXL];
  int prow_end[EN_MAXL];   // end (exclusive) of layer i's packed rows in the stage's concatenated row space; unused entries repeat the last
  int K, G, spr /* steps per packed row */, total_prow;
  int rows_q, rows_rem;    // total_prow = nwg * rows_q + rows_rem: workgroup c owns rows_q + (c < rows_rem) rows
  uint32_t spr_inv;        // ceil(2^32 / spr) (spr > 1): t / spr = umulhi(t, spr_inv) for the step counts that occur (en_div)
  int pad_;
};
static_assert(sizeof(EnStage) == 256, "one descriptor per 256-byte record");

struct EnArgs {
  const EnStage* stages;
  uint32_t* sync;       // EN_SYNC_WORDS words, zero at launch
  int n_stages;
  int part_off, tab_off, ybuf_off;   // LDS byte offsets behind the x buffer
  int maxf;             // most rows a streaming wave touches in one stage
#ifdef EN_LAB_TS
  unsigned long long* ts;   // lab: [workgroup][stage][8] control-wave time stamps
#endif
};

__device__ __forceinline__ void lds_barrier() {
  // workgroup barrier that orders LDS only: __syncthreads() would also wait for every weight load in flight (vmcnt(0))
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ float row16_sum(float v) {   // sum over each row of 16 lanes, valid in every lane of the row
  auto dpp_add = [](float x, auto ctrl) {
    const int y = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xF, 0xF, true);
    return x + __builtin_bit_cast(float, y);
  };
  v = dpp_add(v, std::integral_constant<int, 0xB1>{});
  v = dpp_add(v, std::integral_constant<int, 0x4E>{});
  v = dpp_add(v, std::integral_constant<int, 0x141>{});
  v = dpp_add(v, std::integral_constant<int, 0x140>{});
  return v;
}

// Address spaces, spelled out: stage descriptors are read through the CONSTANT address space (scalar loads; a plain pointer
// would make them vector loads — the kernel also stores to global memory — and every descriptor read would then sit in the same
// counter as the weight stream); weights, group constants, x, y through GLOBAL (a pointer fetched from memory is otherwise
// generic: flat_load counts in both vmcnt and lgkmcnt).
#define EN_GLOBAL __attribute__((address_space(1)))
#define EN_CONST __attribute__((address_space(4)))
typedef const EnStage EN_CONST* en_stage_p;
template <typename T> __device__ __forceinline__ const T EN_GLOBAL* as_global(const T* p) { return (const T EN_GLOBAL*)p; }
template <typename T> __device__ __forceinline__ T EN_GLOBAL* as_global(T* p) { return (T EN_GLOBAL*)p; }

// the chunk of a stage's row space workgroup c owns, and the range of its steps streaming wave w owns
struct EnGeo { int r0, nrows, S; };
__device__ __forceinline__ EnGeo en_geo(en_stage_p st, int c) {
  const int rows_q = st->rows_q, rows_rem = st->rows_rem;
  EnGeo g;
  g.r0 = c * rows_q + (c < rows_rem ? c : rows_rem);
  g.nrows = rows_q + (c < rows_rem ? 1 : 0);
  g.S = g.nrows * st->spr;
  return g;
}
// t / spr for the step counts that occur (spr_inv = ceil(2^32 / spr) does not fit 32 bits for spr = 1)
__device__ __forceinline__ int en_div(int t, int spr, uint32_t spr_inv) {
  return spr == 1 ? t : static_cast<int>(__umulhi(static_cast<uint32_t>(t), spr_inv));
}
__device__ __forceinline__ void en_range(int S, int w, int& a, int& b) {
  const int qs = S / EN_NSW, rs = S - qs * EN_NSW;
  a = w * qs + (w < rs ? w : rs);
  b = a + qs + (w < rs ? 1 : 0);
}

template <int U>
struct EnUnit {        // one unit in flight: U steps of 16 bytes of packed weights per lane + the raw group constants this lane fetched
  u32x4 w[U];
  uint16_t z[U], sc[U];
};

template <int NBITS, bool SUB, int U>
__global__ __launch_bounds__(EN_WAVES * 64) void decode_engine_kernel(const EnArgs a) {
  static_assert(U >= 1 && U <= 8, "steps per unit");
  constexpr int PER = 8 / NBITS;
  static_assert(PER * 16 <= 64, "one dword per lane fetches the step's group constants (two arrays, PER slabs, 16 groups)");
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  u32x4* xs = reinterpret_cast<u32x4*>(smem);                            // [steps][2 planes][64 lanes] x 16 B
  float* part = reinterpret_cast<float*>(smem + a.part_off);             // [EN_NSW][maxf][PER][16]
  int* tab = reinterpret_cast<int*>(smem + a.tab_off);                   // first / last chunk row of each streaming wave
  uint16_t* ybuf = reinterpret_cast<uint16_t*>(smem + a.ybuf_off);       // the chunk's outputs, fp16 bits

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = blockIdx.x, nwg = gridDim.x;
  const int n_stages = a.n_stages;
  const int maxf = a.maxf;
  const en_stage_p stages = (en_stage_p)a.stages;

  // ---- after the workgroup's streaming waves have left their partials: 16 lanes per output, partials added in wave order ----
  auto reduce_stage = [&](const EnGeo& g) {
    const int nout = g.nrows * PER;
    const int j = tid & 15;                       // lane j of a row of 16 lanes adds what streaming wave j left for the output
    const int fr = tab[j], lr = tab[16 + j];      // (j = 15: the control wave's slots hold an empty range)
    for (int o = tid >> 4; o < nout; o += EN_WAVES * 4) {
      const int rl = o / PER, slab = o - rl * PER;
      float sum = 0.f;
      if (rl >= fr && rl <= lr) {
        const f32x4* p4 = reinterpret_cast<const f32x4*>(part + ((j * maxf + (rl - fr)) * PER + slab) * 16);
        const f32x4 v0 = p4[0], v1 = p4[1], v2 = p4[2], v3 = p4[3];
        sum = ((v0[0] + v0[1]) + (v0[2] + v0[3])) + ((v1[0] + v1[1]) + (v1[2] + v1[3]));
        sum += ((v2[0] + v2[1]) + (v2[2] + v2[3])) + ((v3[0] + v3[1]) + (v3[2] + v3[3]));
      }
      sum = row16_sum(sum);                       // fixed tree over the waves: reproducible
      if (j == 0) ybuf[o] = __builtin_bit_cast(uint16_t, static_cast<half_t>(sum));
    }
  };

  if (wave == EN_NSW) {
    // =============================== control wave ===============================
    auto stage_x = [&](en_stage_p st) {
      // x[K] -> LDS in the order the weight rebuild produces values; reads past K return 0 (buffer bounds)
      const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(st->x), 0, st->K * 2, 0x00020000);
      const int nsteps = st->spr;
      constexpr int XB = 6;   // steps per batch: 12 loads of 16 bytes in flight per lane
      for (int s0 = 0; s0 < nsteps; s0 += XB) {
        u32x4 v[XB][2];
#pragma unroll
        for (int u = 0; u < XB; ++u) {
          const int off = ((s0 + u) * 64 + lane) * 32;
          v[u][0] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 16 /* sc1 */);
          v[u][1] = __builtin_amdgcn_raw_buffer_load_b128(rx, off + 16, 0, 16);
        }
#pragma unroll
        for (int u = 0; u < XB; ++u)
          if (s0 + u < nsteps) {
            xs[((s0 + u) * 2 + 0) * 64 + lane] = permute_x8(v[u][0]);
            xs[((s0 + u) * 2 + 1) * 64 + lane] = permute_x8(v[u][1]);
          }
      }
    };
    bool dead = false;   // a wait timed out somewhere: run to the end without waiting
    const uint32_t my_cnt = lane < EN_SHARDS ? static_cast<uint32_t>((nwg - lane + EN_SHARDS - 1) / EN_SHARDS) : 0u;
    uint32_t EN_GLOBAL* sync = as_global(a.sync);
#ifdef EN_LAB_TS
    unsigned long long* ts = a.ts ? a.ts + static_cast<size_t>(c) * n_stages * 8 : nullptr;
#define EN_TS(s, i) if (ts && lane == 0) ts[(s) * 8 + (i)] = __builtin_amdgcn_s_memrealtime();
#else
#define EN_TS(s, i)
#endif
    if (lane == 0) { tab[EN_NSW] = 1; tab[16 + EN_NSW] = 0; }   // reduce_stage: "wave 15" contributes to no row
    stage_x(stages);
    lds_barrier();   // B0
    for (int s = 0; s < n_stages; ++s) {
      const en_stage_p st = stages + s;
      const EnGeo g = en_geo(st, c);
      EN_TS(s, 0)
      lds_barrier();   // B1: partials of stage s are in LDS
      EN_TS(s, 1)
#ifdef EN_LAB_DUMP
      if (a.ts && c == 0 && s == 0) {   // lab: LDS behind the x buffer, before the reduction
        const uint32_t* src = reinterpret_cast<const uint32_t*>(smem + a.part_off);
        uint32_t* dst = reinterpret_cast<uint32_t*>(a.ts);
        for (int i = lane; i < 4096; i += 64) dst[i] = src[i];
      }
#endif
      reduce_stage(g);
      lds_barrier();   // B1': ybuf complete
#ifdef EN_LAB_DUMP
      if (a.ts && c == 0 && s == 0) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(smem + a.tab_off);
        uint32_t* dst = reinterpret_cast<uint32_t*>(a.ts) + 4096;
        for (int i = lane; i < 256; i += 64) dst[i] = src[i];
      }
#endif
      // ---- publish this chunk of y (write-through), then arrive ----
      const int nout = g.nrows * PER;
      const int pe0 = st->prow_end[0], pe1 = st->prow_end[1], pe2 = st->prow_end[2];
      for (int o = lane; o < nout; o += 64) {
        const int rl = o / PER, slab = o - rl * PER;
        const int row = g.r0 + rl;
        const int l = (row >= pe0) + (row >= pe1) + (row >= pe2);   // entries past the last layer repeat it: never true
        const int row0 = l == 0 ? 0 : l == 1 ? pe0 : l == 2 ? pe1 : pe2;
        const int Nl = l == 0 ? st->N[0] : l == 1 ? st->N[1] : l == 2 ? st->N[2] : st->N[3];
        half_t* yl = l == 0 ? st->y[0] : l == 1 ? st->y[1] : l == 2 ? st->y[2] : st->y[3];
        const half_t* bl = l == 0 ? st->bias[0] : l == 1 ? st->bias[1] : l == 2 ? st->bias[2] : st->bias[3];
        const int n = (row - row0) + slab * (Nl / PER);
        half_t v = __builtin_bit_cast(half_t, ybuf[o]);
        if (bl) v = v + as_global(bl)[n];   // `out += bias` on the rounded matmul result (quantize.py:896-897)
        __hip_atomic_store(reinterpret_cast<uint16_t EN_GLOBAL*>(as_global(yl)) + n, __builtin_bit_cast(uint16_t, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the write-through stores have been acknowledged
      EN_TS(s, 2)
      if (lane == 0) __hip_atomic_fetch_add(sync + (c % EN_SHARDS) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (s + 1 == n_stages) break;
      // ---- every chunk of stage s published? ----
      if (!dead) {
        const uint32_t target = my_cnt * static_cast<uint32_t>(s + 1);
        for (uint32_t spins = 0;; ++spins) {
          uint32_t v = 0xFFFFFFFFu;
          if (lane <= EN_SHARDS) v = __hip_atomic_load(sync + lane * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const bool ok = lane >= EN_SHARDS || v >= target;
          const bool tmo = lane == EN_SHARDS && v != 0u;
          if (__builtin_amdgcn_ballot_w64(tmo) != 0 || spins >= EN_SPIN_LIMIT) {
            if (lane == 0) __hip_atomic_store(sync + EN_SHARDS * 32, 1u + static_cast<uint32_t>(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            dead = true;
            break;
          }
          if (__builtin_amdgcn_ballot_w64(ok) == ~0ull) break;
          __builtin_amdgcn_s_sleep(1);
        }
      }
      EN_TS(s, 3)
      stage_x(st + 1);
      EN_TS(s, 4)
      lds_barrier();   // B2: x of stage s + 1 is in LDS
    }
    return;
  }

  // =============================== streaming waves ===============================
  uint32_t magic;
  asm volatile("v_mov_b32 %0, 0x64006400" : "=v"(magic));
  // SUB: the lane that fetches a step's group constants scales them once, (z, s) -> (z 2^-J, s 2^J), J by the slab it fetched for
  half2_t f_lane = {static_cast<half_t>(1.0f), static_cast<half_t>(1.0f)};
  if constexpr (SUB) {
    const int slab = (lane >> 4) < PER ? (lane >> 4) : 0;
    const int J = 9 - NBITS * (PER - 1 - slab);
    f_lane = half2_t{static_cast<half_t>(1.0f / static_cast<float>(1 << J)), static_cast<half_t>(static_cast<float>(1 << J))};
  }

  // ---- issue cursor: the next unit to request (two units are in flight at any time, across stage boundaries) ----
  // A unit is the next min(U, steps left in this wave's range of the stage) steps; every issue() emits exactly 3 U loads
  // (weights, zero, scale per step; steps the unit does not have re-read a valid address), so the wait before a unit is consumed
  // is the exact s_waitcnt vmcnt(3 U) the compiler derives.
  const int lane_k = lane * 16;                               // byte (= k) offset of this lane inside a step
  const int lane_g = lane & 15;
  const int lane_slab = (lane >> 4) < PER ? (lane >> 4) : 0;  // lanes >= 16 * PER duplicate slab 0 (never consumed)
  int is = 0, it = 0, it_end = 0, ikstep = 0, irow = 0;
  bool ilive = true;
  int iK = 0, iG = 0, ispr = 1;
  en_stage_p ist = stages;
  const uint8_t EN_GLOBAL* iwrow = nullptr;      // first byte of the current packed row
  const uint8_t EN_GLOBAL* iWq = nullptr;
  const half_t EN_GLOBAL *izero = nullptr, *iscale = nullptr;
  int irow0 = 0, iend = 0, irps = 1;   // the layer the row belongs to: first / end row in the stage's row space, rows per slab
  int ibase_r = 0;                     // per lane: (p + lane_slab * rows_per_slab) * G

  auto iss_layer = [&]() {   // irow left the current layer (or a new stage began): pick the layer
    const int pe0 = ist->prow_end[0], pe1 = ist->prow_end[1], pe2 = ist->prow_end[2];
    const int l = (irow >= pe0) + (irow >= pe1) + (irow >= pe2);
    iWq = as_global(ist->Wq[l]);
    izero = as_global(ist->zero[l]);
    iscale = as_global(ist->scale[l]);
    irow0 = l == 0 ? 0 : ist->prow_end[l - 1];
    iend = ist->prow_end[l];
    irps = ist->N[l] / PER;
  };
  auto iss_row = [&]() {
    if (irow >= iend) iss_layer();
    const int p = irow - irow0;
    iwrow = iWq + static_cast<int64_t>(p) * iK;
    ibase_r = (p + lane_slab * irps) * iG;
  };
  auto iss_stage = [&]() {   // position the cursor on this wave's first step of stage `is` (skipping stages it has no step in)
    for (;;) {
      if (is >= n_stages) { ilive = false; return; }   // past the end: harmless re-reads of the last row's first bytes
      ist = stages + is;
      const EnGeo g = en_geo(ist, c);
      int ra, rb;
      en_range(g.S, wave, ra, rb);
      if (rb > ra) {
        iK = ist->K; iG = ist->G; ispr = ist->spr;
        const int rl = en_div(ra, ispr, ist->spr_inv);
        it = ra; it_end = rb;
        ikstep = ra - rl * ispr;
        irow = g.r0 + rl;
        iend = -1;   // force the layer pick
        iss_row();
        return;
      }
      ++is;
    }
  };
  auto issue = [&](EnUnit<U>& un) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool real = ilive && it < it_end;   // wave-uniform
      int koff = ikstep * EN_STEP + lane_k;
      int g = ikstep * 16 + lane_g;
      if (!real) { koff = lane_k; g = lane_g; }
      koff = koff < iK ? koff : 0;               // lanes past K re-read the row start: their x is zero in LDS
      g = g < iG ? g : 0;
      const int r = ibase_r + g;
      un.z[u] = __builtin_bit_cast(uint16_t, izero[r]);
      un.sc[u] = __builtin_bit_cast(uint16_t, iscale[r]);
      un.w[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 EN_GLOBAL*>(iwrow + koff));
      if (real) {
        ++it;
        if (it < it_end && ++ikstep == ispr) { ikstep = 0; ++irow; iss_row(); }
      }
    }
    if (ilive && it == it_end) { ++is; iss_stage(); }   // the next unit starts the wave's range of a later stage
  };

  // ---- consume cursor ----
  int cs = 0, ct = 0, ct_end = 0, ckstep = 0, crow = 0, frow = 0, cspr = 1;
  bool cany = false;   // this wave has steps in stage cs
  EnGeo cg{0, 0, 0};
  f32x4 acc[1][PER];
#pragma unroll
  for (int s = 0; s < PER; ++s) acc[0][s] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool diag_lane = ((lane & 15) >> 2) == (lane >> 4);   // the 16 lanes that hold the diagonal of a 16x16 MFMA tile

  auto cons_stage = [&]() {   // this wave's range of stage cs
    const en_stage_p st = stages + cs;
    cg = en_geo(st, c);
    int ra, rb;
    en_range(cg.S, wave, ra, rb);
    cspr = st->spr;
    const uint32_t inv = st->spr_inv;
    ct = ra; ct_end = rb;
    cany = rb > ra;
    crow = en_div(ra, cspr, inv);
    ckstep = ra - crow * cspr;
    frow = crow;
    if (lane == 0) {
      tab[wave] = rb > ra ? frow : 1;
      tab[16 + wave] = rb > ra ? en_div(rb - 1, cspr, inv) : 0;
    }
  };
  auto flush = [&]() {   // the row's partial sums of this wave: the tile diagonals, 16 floats per slab
    const int f = crow - frow;
#pragma unroll
    for (int s = 0; s < PER; ++s) {
      const int i = lane & 3;
      const float v = i == 0 ? acc[0][s][0] : i == 1 ? acc[0][s][1] : i == 2 ? acc[0][s][2] : acc[0][s][3];
      if (diag_lane) part[((wave * maxf + f) * PER + s) * 16 + (lane & 15)] = v;
      acc[0][s] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  // the next min(U, ct_end - ct) steps of this wave's range are in `un`
  auto consume = [&](const EnUnit<U>& un) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (ct < ct_end) {
        uint32_t mine = static_cast<uint32_t>(un.z[u]) | (static_cast<uint32_t>(un.sc[u]) << 16);
        if constexpr (SUB) mine = __builtin_bit_cast(uint32_t, as_h2(mine) * f_lane);
        uint32_t zs[PER];
#pragma unroll
        for (int s = 0; s < PER; ++s) zs[s] = __builtin_amdgcn_ds_bpermute((s * 16 + (lane >> 2)) << 2, mine);
        h8_t b0[1], b1[1];
        b0[0] = __builtin_bit_cast(h8_t, xs[(ckstep * 2 + 0) * 64 + lane]);
        b1[0] = __builtin_bit_cast(h8_t, xs[(ckstep * 2 + 1) * 64 + lane]);
        SlabExact<NBITS, 1, 0, PER, SUB>::run(un.w[u], zs, b0, b1, acc, magic);
        ++ct;
        if (++ckstep == cspr) { flush(); ckstep = 0; ++crow; }
      }
    }
  };
#ifdef EN_LAB_TS
  unsigned long long* wts = (a.ts && (wave == 0 || wave == EN_NSW - 1)) ? a.ts + static_cast<size_t>(c) * n_stages * 8 : nullptr;
#define EN_WTS(s, i) if (wts && lane == 0) wts[(s) * 8 + (i) + (wave == 0 ? 0 : 1)] = __builtin_amdgcn_s_memrealtime();
#else
#define EN_WTS(s, i)
#endif
  // this wave's part of stage cs is done: partials -> outputs -> (next stage's x); false after the last stage
  auto next_stage = [&]() -> bool {
    if (cany && ckstep != 0) flush();   // a row this wave shares with the next one
    EN_WTS(cs, 5)
    lds_barrier();   // B1
    reduce_stage(cg);
    lds_barrier();   // B1'
    if (++cs == n_stages) return false;
    lds_barrier();   // B2
    cons_stage();
    return true;
  };

  // ---- prologue: two units in flight before anything else, then wait for x of stage 0 ----
  // (a wave without a single step in the whole plan still requests something valid: row 0 of stage 0)
  iK = ist->K; iG = ist->G; irow = 0; iend = -1;
  iss_row();
  iss_stage();
  EnUnit<U> ua, ub;
  issue(ua);
  issue(ub);
  cons_stage();
  lds_barrier();   // B0

  // ---- ping-pong: consume the older unit, request the unit after the younger one into its registers ----
  for (;;) {
    while (ct == ct_end) if (!next_stage()) return;
    consume(ua);
    issue(ua);
    while (ct == ct_end) if (!next_stage()) return;
    consume(ub);
    issue(ub);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// host side: the plan (a caller-owned blob: header | stage descriptors | sync words, identical on host and device)
// ---------------------------------------------------------------------------------------------------------------------
constexpr uint32_t EN_MAGIC = 0x48515145u;   // "EQQH"
struct alignas(256) EnPlanHeader {
  uint32_t magic, version;
  int n_stages, nbits, sub, grid;
  int xs_bytes, part_off, tab_off, ybuf_off, maxf, lds_bytes;
  uint64_t stages_off, sync_off, total;
};
static_assert(sizeof(EnPlanHeader) == 256, "header record");
static inline size_t en_align256(size_t v) { return (v + 255) & ~static_cast<size_t>(255); }
static inline size_t en_plan_bytes(int n_stages) {
  return sizeof(EnPlanHeader) + sizeof(EnStage) * static_cast<size_t>(n_stages) + en_align256(sizeof(uint32_t) * EN_SYNC_WORDS);
}

template <int NBITS>
static int en_launch(const EnPlanHeader& h, const EnArgs& a, hipStream_t st) {
  auto launch = [&](auto kern) -> int {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, EN_LDS_MAX);
    if (e != hipSuccess) { set_error("hqq_hip_decode_run: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e)); return static_cast<int>(e); }
    hipLaunchKernelGGL(kern, dim3(h.grid), dim3(EN_WAVES * 64), h.lds_bytes, st, a);
    return check_launch("hqq_hip_decode_run");
  };
  return h.sub ? launch(decode_engine_kernel<NBITS, true, EN_U>) : launch(decode_engine_kernel<NBITS, false, EN_U>);
}

}  // namespace hqq

using namespace hqq;

#ifdef EN_LAB_TS
unsigned long long* g_en_lab_ts = nullptr;
extern "C" void hqq_hip_lab_set_engine_ts(unsigned long long* p) { g_en_lab_ts = p; }
#endif

extern "C" size_t hqq_hip_decode_plan_bytes(int n_stages) { return n_stages < 1 ? 0 : en_plan_bytes(n_stages); }

extern "C" int hqq_hip_decode_plan_init(void* plan_host, size_t plan_bytes, int nbits, int64_t group_size, int dtype, int64_t M, uint32_t opts,
                                        const hqq_hip_decode_stage* stages, int n_stages, int grid) {
  clear_stale_error();
  if (!plan_host || !stages || n_stages < 1) { set_error("hqq_hip_decode_plan_init: null / empty argument"); return HQQ_ERR_SHAPE; }
  if (plan_bytes < en_plan_bytes(n_stages)) { set_error("hqq_hip_decode_plan_init: plan buffer %zu < %zu bytes", plan_bytes, en_plan_bytes(n_stages)); return HQQ_ERR_WORKSPACE; }
  if (nbits != 8 && nbits != 4 && nbits != 2) { set_error("hqq_hip_decode_plan_init: the engine covers nbits 8/4/2 (got %d)", nbits); return HQQ_ERR_UNSUPPORTED; }
  if (group_size != 64 || dtype != HQQ_F16 || M != 1) {
    set_error("hqq_hip_decode_plan_init: the engine covers group_size 64, fp16, one activation row (got gs=%lld dtype=%d M=%lld)", (long long)group_size, dtype, (long long)M);
    return HQQ_ERR_UNSUPPORTED;
  }
  if (opts & ~(HQQ_OPT_META_SCALABLE)) { set_error("hqq_hip_decode_plan_init: unknown option bits 0x%x", opts); return HQQ_ERR_SHAPE; }
  if (grid <= 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) { (void)hipGetLastError(); n = 256; }
    grid = n;   // one workgroup per CU: the whole grid is co-resident
  }
  if (grid > 1024) { set_error("hqq_hip_decode_plan_init: grid %d too large", grid); return HQQ_ERR_SHAPE; }
  const int per = 8 / nbits;
  EnPlanHeader* h = static_cast<EnPlanHeader*>(plan_host);
  EnStage* out = reinterpret_cast<EnStage*>(static_cast<char*>(plan_host) + sizeof(EnPlanHeader));
  memset(plan_host, 0, en_plan_bytes(n_stages));
  int max_spr = 0, maxf = 1, max_rows = 0;
  for (int s = 0; s < n_stages; ++s) {
    const hqq_hip_decode_stage& in = stages[s];
    EnStage& d = out[s];
    const int64_t K = in.K;
    if (in.n_layers < 1 || in.n_layers > EN_MAXL) { set_error("hqq_hip_decode_plan_init: stage %d has %d layers (1..%d)", s, in.n_layers, EN_MAXL); return HQQ_ERR_SHAPE; }
    if (K <= 0 || K % 64 || K > (1 << 20)) { set_error("hqq_hip_decode_plan_init: stage %d: K=%lld must be a multiple of the group size (<= 2^20)", s, (long long)K); return HQQ_ERR_SHAPE; }
    if (!in.x || !aligned16(in.x)) { set_error("hqq_hip_decode_plan_init: stage %d: x null or not 16-byte aligned", s); return in.x ? HQQ_ERR_ALIGN : HQQ_ERR_SHAPE; }
    int64_t total = 0;
    for (int i = 0; i < in.n_layers; ++i) {
      if (in.N[i] <= 0 || in.N[i] % per) { set_error("hqq_hip_decode_plan_init: stage %d layer %d: N=%lld must divide by %d", s, i, (long long)in.N[i], per); return in.N[i] <= 0 ? HQQ_ERR_SHAPE : HQQ_ERR_UNSUPPORTED; }
      if (in.N[i] * (K / 64) > INT32_MAX / 2) { set_error("hqq_hip_decode_plan_init: size overflow"); return HQQ_ERR_SHAPE; }
      if (!in.Wq[i] || !in.scale[i] || !in.zero[i] || !in.y[i]) { set_error("hqq_hip_decode_plan_init: stage %d layer %d: null pointer", s, i); return HQQ_ERR_SHAPE; }
      if (!aligned16(in.Wq[i])) { set_error("hqq_hip_decode_plan_init: packed weights must be 16-byte aligned"); return HQQ_ERR_ALIGN; }
      total += in.N[i] / per;
      if (total > (1 << 24)) { set_error("hqq_hip_decode_plan_init: size overflow"); return HQQ_ERR_SHAPE; }
      d.Wq[i] = static_cast<const uint8_t*>(in.Wq[i]);
      d.scale[i] = static_cast<const half_t*>(in.scale[i]);
      d.zero[i] = static_cast<const half_t*>(in.zero[i]);
      d.bias[i] = static_cast<const half_t*>(in.bias[i]);
      d.y[i] = static_cast<half_t*>(in.y[i]);
      d.N[i] = static_cast<int>(in.N[i]);
      d.prow_end[i] = static_cast<int>(total);
    }
    for (int i = in.n_layers; i < EN_MAXL; ++i) {
      d.Wq[i] = d.Wq[in.n_layers - 1]; d.scale[i] = d.scale[in.n_layers - 1]; d.zero[i] = d.zero[in.n_layers - 1]; d.bias[i] = d.bias[in.n_layers - 1];
      d.y[i] = d.y[in.n_layers - 1]; d.N[i] = d.N[in.n_layers - 1]; d.prow_end[i] = d.prow_end[in.n_layers - 1];
    }
    d.x = static_cast<const half_t*>(in.x);
    d.K = static_cast<int>(K);
    d.G = static_cast<int>(K / 64);
    d.spr = static_cast<int>((K + EN_STEP - 1) / EN_STEP);
    d.total_prow = static_cast<int>(total);
    d.rows_q = d.total_prow / grid;
    d.rows_rem = d.total_prow % grid;
    d.spr_inv = static_cast<uint32_t>(((1ull << 32) + d.spr - 1) / d.spr);
    const int rows = d.rows_q + (d.rows_rem ? 1 : 0);
    const int64_t S = static_cast<int64_t>(rows) * d.spr;
    if (S > (1 << 20)) { set_error("hqq_hip_decode_plan_init: stage %d: %lld steps per workgroup", s, (long long)S); return HQQ_ERR_UNSUPPORTED; }
    const int L = static_cast<int>((S + EN_NSW - 1) / EN_NSW);        // most steps one streaming wave owns
    const int f = L > 0 ? (L - 1) / d.spr + 2 : 1;                     // most rows it touches
    maxf = f > maxf ? f : maxf;
    max_spr = d.spr > max_spr ? d.spr : max_spr;
    max_rows = rows > max_rows ? rows : max_rows;
  }
  h->magic = EN_MAGIC; h->version = HQQ_HIP_ABI_VERSION;
  h->n_stages = n_stages; h->nbits = nbits; h->sub = (opts & HQQ_OPT_META_SCALABLE) ? 1 : 0; h->grid = grid;
  h->xs_bytes = max_spr * 2 * 64 * 16;
  h->maxf = maxf;
  h->part_off = h->xs_bytes;
  h->tab_off = h->part_off + static_cast<int>(en_align256(sizeof(float) * EN_NSW * maxf * per * 16));
  h->ybuf_off = h->tab_off + 256;
  h->lds_bytes = h->ybuf_off + static_cast<int>(en_align256(sizeof(uint16_t) * static_cast<size_t>(max_rows) * per));
  if (h->lds_bytes > EN_LDS_MAX) { set_error("hqq_hip_decode_plan_init: %d bytes of LDS needed (K too long / chunks too tall for one workgroup)", h->lds_bytes); return HQQ_ERR_UNSUPPORTED; }
  h->stages_off = sizeof(EnPlanHeader);
  h->sync_off = sizeof(EnPlanHeader) + sizeof(EnStage) * static_cast<size_t>(n_stages);
  h->total = en_plan_bytes(n_stages);
  return 0;
}

extern "C" int hqq_hip_decode_run(const void* plan_host, void* plan_dev, size_t plan_bytes, void* stream) {
  clear_stale_error();
  if (!plan_host || !plan_dev) { set_error("hqq_hip_decode_run: null plan"); return HQQ_ERR_SHAPE; }
  const EnPlanHeader& h = *static_cast<const EnPlanHeader*>(plan_host);
  if (h.magic != EN_MAGIC || h.total > plan_bytes || h.n_stages < 1) { set_error("hqq_hip_decode_run: not an initialised plan"); return HQQ_ERR_SHAPE; }
  if (reinterpret_cast<uintptr_t>(plan_dev) & 255u) { set_error("hqq_hip_decode_run: the device copy of the plan must be 256-byte aligned"); return HQQ_ERR_ALIGN; }
  hipStream_t st = as_stream(stream);
  char* base = static_cast<char*>(plan_dev);
  // arrival counters and the time-out word start from zero on every call (a memset node under graph capture)
  hipError_t e = hipMemsetAsync(base + h.sync_off, 0, sizeof(uint32_t) * EN_SYNC_WORDS, st);
  if (e != hipSuccess) { set_error("hqq_hip_decode_run: hipMemsetAsync: %s", hipGetErrorString(e)); return static_cast<int>(e); }
  EnArgs a;
  a.stages = reinterpret_cast<const EnStage*>(base + h.stages_off);
  a.sync = reinterpret_cast<uint32_t*>(base + h.sync_off);
  a.n_stages = h.n_stages;
  a.part_off = h.part_off; a.tab_off = h.tab_off; a.ybuf_off = h.ybuf_off; a.maxf = h.maxf;
#ifdef EN_LAB_TS
  a.ts = g_en_lab_ts;
#endif
  switch (h.nbits) {
    case 8: return en_launch<8>(h, a, st);
    case 4: return en_launch<4>(h, a, st);
    case 2: return en_launch<2>(h, a, st);
  }
  return HQQ_ERR_NBITS;
}

/* the time-out word of the last run (0 = every hand-off completed); the caller copies it back after synchronising */
extern "C" size_t hqq_hip_decode_plan_status_offset(const void* plan_host) {
  const EnPlanHeader& h = *static_cast<const EnPlanHeader*>(plan_host);
  return h.magic == EN_MAGIC ? static_cast<size_t>(h.sync_off) + sizeof(uint32_t) * EN_SHARDS * 32 : 0;
}
