// kwave.hip — fused unpack -> dequantize -> GEMM for decode with a batch (5 <= M <= 64 activation rows; fp16 / bf16; 8-/4-/3(stream)-/2-bit,
// group_size 64) WITHOUT a K split across workgroups.  gfx950.  Round 5; supersedes skinny.hip's split-K launches where it covers.
//
// Reference chain replaced (axis = 1): BitPack.unpack_* -> (W_r - zero) * scale -> torch.matmul(x, W.t()) (+ bias)
//   hqq/core/bitpack.py:31-64, hqq/core/quantize.py:183-199, :880-898; patching.py:82-86.
// The weights it multiplies are bit-identical to hqq_hip_dequantize / Quantizer.dequantize (two roundings in the compute dtype); the fp32
// summation order is fixed by the shape alone (no atomics, no arrival order), so a row of y does not depend on the batch it sits in.
//
// Why another kernel.  skinny.hip fills the chip by cutting K across workgroups: partial tiles through HBM
// scratch, an arrival counter, three device-scope round trips in the finish — ~9 us per launch that no byte and no flop explains.
// tools/lab_batch/batch.hip removed the split but pushed all of x through LDS-DMA (25-45 GB/s per CU) and lost.  Here:
//   * a workgroup owns RT tiles of 16 OUTPUT rows (16 / per packed rows x per slabs: one MFMA A tile holds every slab of its packed
//     rows, lanes 0..RP-1 of a 16-lane row slab 0, the next RP slab 1, ...) and the WHOLE K: no partial tiles leave the CU;
//   * its eight waves split K (steps of 128 k, wave w takes steps [w S / 8, (w + 1) S / 8)) and meet ONCE, in LDS, after the loop:
//     sums in wave order, one rounding, store — bits reproducible by construction;
//   * a wave reads only ITS K slice of x, by plain buffer loads straight into MFMA B fragments (16 tokens x 64 contiguous bytes per
//     instruction, L2 hits): nothing goes through LDS or LDS-DMA; rows >= M are out of the buffer's range and read as zero;
//   * weights: one load instruction per tile and step covers whole 128-byte lines (lane (rp, h, c): packed row rp, piece h of the
//     step, 16 / 12 / 8 bytes), and the piece a lane's slab needs comes over by one DPP move (two slabs) or ds_swizzle (four slabs)
//     per dword — each lane rebuilds ONLY its own slab, with the slab's bit offset as a lane constant (mask, 2^-sh);
//   * group constants: lane (R, c) loads the dword holding (zero | zero') resp. (scale | scale') of its output row for step
//     4 q + c — 16 rows x 16 contiguous bytes per instruction, once per four steps — and ds_bpermute hands step j's dword round;
//   * everything a step needs (weights, x, every fourth step the constants) is one unit of a register ring, D steps ahead; dead
//     units (past the wave's range) read through a zero-length descriptor: zeros, no traffic, exact vmcnt counts everywhere.
// RT (tiles per workgroup) is chosen on the host so that the grid is about one workgroup per CU (x is re-read by every workgroup:
// 2 M / (RP RT) bytes of x per weight byte pass through each CU's L1); it changes which rows share a workgroup, never a sum.
#include <type_traits>

#include "hqq_common.h"   // (lab: built with -Ihqq_amd/csrc, tools/lab_kwave/build.sh)
#include "w3s.h"
#include <stdlib.h>

#ifndef KW_NBITS
#error "compile with -DKW_NBITS=8|4|3|2 (Makefile)"
#endif

namespace hqq {
namespace kw {

constexpr int KW_MAXL = HQQ_GEMV_MAX_GROUP;
constexpr int KW_WAVES = 8;
constexpr int KW_T = KW_WAVES * 64;
[[maybe_unused]] constexpr int KW_KSTEP = 128;             // k per step
constexpr int KW_MAX_RTMT = 12;           // RT * MT <= 12: the eight waves' partial tiles meet in 96 KiB of LDS

constexpr int kw_per(int nbits) { return nbits == 3 ? 2 : 8 / nbits; }               // slabs per packed row (3: the stream layout)
constexpr int kw_wv(int nbits) { return nbits == 8 ? 8 : nbits == 4 ? 4 : nbits == 3 ? 3 : 2; }   // dwords per lane, tile and step
constexpr int kw_row_bytes_num(int nbits) { return nbits == 3 ? 3 : 4; }             // packed row bytes = K * num / 4
// steps a wave keeps in flight (a divisor of the four-step round): four where the ring (16 MT + wv RT registers per step), the
// accumulators and the constants leave the rebuild its ~70 working registers inside 256, else two
constexpr int kw_depth(int nbits, int mt, int rt) {
  return 4 * (16 * mt + kw_wv(nbits) * rt) + 4 * rt * mt + 4 * rt + 70 <= 275 ? 4 : 2;
}
// (MT, RT) pairs that are built: the partial tiles fit the LDS and a ring of two steps fits the registers (measured: no scratch)
constexpr bool kw_combo(int nbits, int mt, int rt) {
  return rt * mt <= KW_MAX_RTMT && 2 * (16 * mt + kw_wv(nbits) * rt) + 4 * rt * mt + 4 * rt + 70 <= 265;
}

struct KwArgs {
  const uint8_t* Wq[KW_MAXL];
  const half_t* scale[KW_MAXL];
  const half_t* zero[KW_MAXL];
  const half_t* bias[KW_MAXL];
  half_t* y[KW_MAXL];
  int N[KW_MAXL];        // out_features
  int wg_end[KW_MAXL];   // end (exclusive) of layer i's workgroups (a workgroup never straddles two layers); unused entries repeat the last
  const half_t* x;
  int M, K, S;           // S = K / 128 steps
};

typedef _Float16 kw_h8_t __attribute__((ext_vector_type(8)));
typedef __bf16 kw_bf2_t __attribute__((ext_vector_type(2)));
typedef __bf16 kw_bf8_t __attribute__((ext_vector_type(8)));
typedef float kw_f2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t kw_u32(half2_t h) { return __builtin_bit_cast(uint32_t, h); }

// per-lane constants of the lane's slab (bit offset sh of its field inside a byte)
struct KwLane {
  uint32_t mask;      // fp16 paths: field mask in both 16-bit halves; bf16: in all four bytes
  half2_t k1, k2;     // four-op rebuild: 2^-sh, -1024 2^-sh
  half2_t fz, fs;     // three-op rebuild: 2^-J, 2^J (J = 9 - sh), applied to the fetched (zero, zero') / (scale, scale') pairs
  float inv;          // bf16: 2^-sh
  bool slab1;         // 3-bit stream layout: the lane rebuilds slab 1
};

// (k0..k7) -> (k0,k2,k1,k3,k4,k6,k5,k7): the order the byte-pair extraction produces (decode_common.h permute_x8)
__device__ __forceinline__ u32x4 kw_permute_x8(u32x4 v) {
  u32x4 r;
  r.x = __builtin_amdgcn_perm(v.y, v.x, 0x05040100u);   // (v.x lo, v.y lo)
  r.y = __builtin_amdgcn_perm(v.y, v.x, 0x07060302u);   // (v.x hi, v.y hi)
  r.z = __builtin_amdgcn_perm(v.w, v.z, 0x05040100u);
  r.w = __builtin_amdgcn_perm(v.w, v.z, 0x07060302u);
  return r;
}

// fp16: ND dwords (4 k of the lane's slab per dword) -> 2 ND weight pairs, rebuilt exactly as Quantizer.dequantize does
// (round16(round16(q - z) * s), quantize.py:198); zz / ss: the group's (z, z) / (s, s) — already scaled by 2^-J / 2^J when SUB
template <int ND, bool SUB>
__device__ __forceinline__ void kw_rebuild_f16(const uint32_t (&w)[ND], half2_t zz, half2_t ss, const KwLane& lc, uint32_t magic, uint32_t (&o)[2 * ND]) {
  half2_t q[2 * ND];
  if constexpr (SUB) {   // the masked field read as fp16 IS the subnormal q 2^(sh-24) (decode_common.h SlabExact<.., SUB>)
    const half2_t lift = {static_cast<half_t>(32768.0f), static_cast<half_t>(32768.0f)};
#pragma unroll
    for (int d = 0; d < ND; ++d) {
      q[2 * d] = as_h2(w[d] & lc.mask);              // bytes (4d+0, 4d+2)
      q[2 * d + 1] = as_h2((w[d] >> 8) & lc.mask);   // bytes (4d+1, 4d+3)
    }
#pragma unroll
    for (int i = 0; i < 2 * ND; ++i) q[i] = __builtin_elementwise_fma(q[i], lift, -zz);   // rounding 1
  } else {
#pragma unroll
    for (int d = 0; d < ND; ++d) {
      q[2 * d] = as_h2((w[d] & lc.mask) | magic);            // 1024 + q 2^sh
      q[2 * d + 1] = as_h2(((w[d] >> 8) & lc.mask) | magic);
    }
#pragma unroll
    for (int i = 0; i < 2 * ND; ++i) q[i] = __builtin_elementwise_fma(q[i], lc.k1, lc.k2);   // exact integer level
#pragma unroll
    for (int i = 0; i < 2 * ND; ++i) q[i] = q[i] - zz;                                          // rounding 1
  }
#pragma unroll
  for (int i = 0; i < 2 * ND; ++i) o[i] = kw_u32(q[i] * ss);                                    // rounding 2
}

// bf16: through fp32 (skinny.hip SkSlabBF16): v_cvt_f32_ubyteN lifts the masked byte F q, one fma forms q - z with a single fp32 rounding,
// v_cvt_pk_bf16_f32 rounds it, v_dot2_f32_bf16 against (s, 0) / (0, s) forms the exact product, a second v_cvt_pk rounds again
template <int B>
__device__ __forceinline__ float kw_ubyte(uint32_t v) { return static_cast<float>((v >> (8 * B)) & 0xFFu); }
template <int ND>
__device__ __forceinline__ void kw_rebuild_bf16(const uint32_t (&w)[ND], uint32_t z_bf, uint32_t s_bf, const KwLane& lc, uint32_t (&o)[2 * ND]) {
  const float zf = __uint_as_float(z_bf << 16);
  const kw_bf2_t s_lo = __builtin_bit_cast(kw_bf2_t, s_bf & 0xFFFFu);   // (s, 0)
  const kw_bf2_t s_hi = __builtin_bit_cast(kw_bf2_t, s_bf << 16);       // (0, s)
#pragma unroll
  for (int d = 0; d < ND; ++d) {
    const uint32_t fq = w[d] & lc.mask;
    const kw_f2_t dq[2] = {{__builtin_fmaf(kw_ubyte<0>(fq), lc.inv, -zf), __builtin_fmaf(kw_ubyte<2>(fq), lc.inv, -zf)},
                           {__builtin_fmaf(kw_ubyte<1>(fq), lc.inv, -zf), __builtin_fmaf(kw_ubyte<3>(fq), lc.inv, -zf)}};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const kw_bf2_t dr = __builtin_convertvector(dq[h], kw_bf2_t);                 // rounding 1
      const kw_f2_t pw = {__builtin_amdgcn_fdot2_f32_bf16(dr, s_lo, 0.f, false), __builtin_amdgcn_fdot2_f32_bf16(dr, s_hi, 0.f, false)};
      o[2 * d + h] = __builtin_bit_cast(uint32_t, __builtin_convertvector(pw, kw_bf2_t));   // rounding 2
    }
  }
}

// 3-bit stream layout: the lane's slab of a 12-byte chunk (w3s.h) -> 8 weight pairs in NATURAL k order.  The pair fields of both slabs
// are extracted (w3s_fields), the lane keeps its slab's; pair 7 sits at another field offset in slab 1 (class 0 instead of 2).
template <bool BF16, bool SUB>
__device__ __forceinline__ void kw_rebuild_w3s(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t z_raw, uint32_t s_raw, const KwLane& lc, uint32_t magic, uint32_t (&o)[8]) {
  uint32_t f2[2][8], f[8];
  w3s_fields(w0, w1, w2, f2);
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = lc.slab1 ? f2[1][j] : f2[0][j];
  if constexpr (BF16) {
    const float zf = __uint_as_float(z_raw << 16);
    const kw_bf2_t s_lo = __builtin_bit_cast(kw_bf2_t, s_raw & 0xFFFFu);
    const kw_bf2_t s_hi = __builtin_bit_cast(kw_bf2_t, s_raw << 16);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float up = j < 7 ? static_cast<float>(1 << (24 - 3 * W3S_CLS[0][j])) : (lc.slab1 ? static_cast<float>(1 << 24) : static_cast<float>(1 << 18));
      const half2_t h = as_h2(f[j]);
      const kw_f2_t dq = {__builtin_fmaf(static_cast<float>(h.x), up, -zf), __builtin_fmaf(static_cast<float>(h.y), up, -zf)};
      const kw_bf2_t dr = __builtin_convertvector(dq, kw_bf2_t);
      const kw_f2_t pw = {__builtin_amdgcn_fdot2_f32_bf16(dr, s_lo, 0.f, false), __builtin_amdgcn_fdot2_f32_bf16(dr, s_hi, 0.f, false)};
      o[j] = __builtin_bit_cast(uint32_t, __builtin_convertvector(pw, kw_bf2_t));
    }
  } else if constexpr (SUB) {
    const half2_t lift = {static_cast<half_t>(32768.0f), static_cast<half_t>(32768.0f)};
    const half2_t pr = as_h2(z_raw | (s_raw << 16));
    half2_t nz[3], ss[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int J = 9 - 3 * c;
      const half2_t fj = {static_cast<half_t>(1.0f / static_cast<float>(1 << J)), static_cast<half_t>(static_cast<float>(1 << J))};
      const half2_t p = pr * fj;   // (z 2^-J, s 2^J): exact for every group of a layer that passed hqq_hip_w3s_meta_check
      nz[c] = half2_t{-p.x, -p.x};
      ss[c] = half2_t{p.y, p.y};
    }
    const half2_t nz7 = lc.slab1 ? nz[0] : nz[2], ss7 = lc.slab1 ? ss[0] : ss[2];
    half2_t q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) q[j] = __builtin_elementwise_fma(as_h2(f[j]), lift, j < 7 ? nz[W3S_CLS[0][j]] : nz7);   // rounding 1
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = kw_u32(q[j] * (j < 7 ? ss[W3S_CLS[0][j]] : ss7));                                // rounding 2
  } else {
    const half2_t zz = {__builtin_bit_cast(half_t, static_cast<uint16_t>(z_raw)), __builtin_bit_cast(half_t, static_cast<uint16_t>(z_raw))};
    const half2_t ss = {__builtin_bit_cast(half_t, static_cast<uint16_t>(s_raw)), __builtin_bit_cast(half_t, static_cast<uint16_t>(s_raw))};
    half2_t q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int p = 3 * W3S_CLS[0][j];
      const float inv = 1.0f / static_cast<float>(1 << p);
      half2_t k1 = {static_cast<half_t>(inv), static_cast<half_t>(inv)};
      half2_t k2 = {static_cast<half_t>(-1024.0f * inv), static_cast<half_t>(-1024.0f * inv)};
      if (j == 7) {   // slab 1: class 0
        const half2_t k1b = {static_cast<half_t>(1.0f), static_cast<half_t>(1.0f)}, k2b = {static_cast<half_t>(-1024.0f), static_cast<half_t>(-1024.0f)};
        k1 = lc.slab1 ? k1b : k1;
        k2 = lc.slab1 ? k2b : k2;
      }
      q[j] = __builtin_elementwise_fma(as_h2(f[j] | magic), k1, k2);   // exact integer level
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) q[j] = q[j] - zz;          // rounding 1
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = kw_u32(q[j] * ss);   // rounding 2
  }
}

template <int NBITS, int MT, int RT>
struct KwUnit {   // what a wave has in flight for one step
  uint32_t w[RT][kw_wv(NBITS)];
  u32x4 x[MT][4];
};

template <int NBITS, int MT, int RT, bool BF16, bool SUB>
__global__ __launch_bounds__(KW_T) void kwave_kernel(const KwArgs a) {
  constexpr bool W3 = NBITS == 3;
  constexpr int PER = kw_per(NBITS);
  constexpr int RP = 16 / PER;              // packed rows per tile
  constexpr int D = kw_depth(NBITS, MT, RT);   // steps in flight per wave
  static_assert(kw_combo(NBITS, MT, RT), "partial tiles exceed the LDS budget / the ring exceeds the registers");
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  using frag_t = std::conditional_t<BF16, kw_bf8_t, kw_h8_t>;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int R = lane & 15, c = lane >> 4;
  const int sl = R / RP, rp = R % RP;       // the lane's slab — also the piece of a step its load covers — and packed row inside the tile
  const int K = a.K, S = a.S, M = a.M;

  // ---- the workgroup's layer (scalar selects) ----
  const int wg = blockIdx.x;
  const uint8_t* Wq = a.Wq[0];
  const half_t* scale = a.scale[0];
  const half_t* zero = a.zero[0];
  int N = a.N[0], wg0 = 0, li = 0;
#pragma unroll
  for (int i = 1; i < KW_MAXL; ++i) {
    const bool in = wg >= a.wg_end[i - 1];
    Wq = pick(in, a.Wq[i], Wq);
    scale = pick(in, a.scale[i], scale);
    zero = pick(in, a.zero[i], zero);
    N = pick(in, a.N[i], N);
    wg0 = pick(in, a.wg_end[i - 1], wg0);
    li += in ? 1 : 0;
  }
  const int rows_per_slab = N / PER;                     // packed rows of the layer
  const int tile0 = (wg - wg0) * RT;                     // first tile of this workgroup inside the layer
  const int row_bytes = K / 4 * kw_row_bytes_num(NBITS);  // bytes per packed row
  const int s0 = wave * S / KW_WAVES, s1 = (wave + 1) * S / KW_WAVES;   // this wave's steps

  // ---- descriptors.  Bounded: packed rows past the layer's last (ragged tile) and activation rows >= M read as zero. ----
  const uint32_t w_bytes = static_cast<uint32_t>(rows_per_slab) * static_cast<uint32_t>(row_bytes);
  const uint32_t x_bytes = static_cast<uint32_t>(M) * static_cast<uint32_t>(K) * 2u;
  const __amdgpu_buffer_rsrc_t rz = buffer_rsrc(zero), rs = buffer_rsrc(scale);

  uint32_t wvoff[RT], xvoff[MT], mbase[RT];
  {
    const int piece = W3 ? sl * 48 + c * 12 : NBITS == 4 ? sl * 64 + c * 16 : NBITS == 2 ? sl * 32 + c * 8 : c * 16;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int prow = (tile0 + rt) * RP + rp;
      // (row index capped so that the product stays below 2^32; anything >= rows_per_slab is out of range anyway)
      wvoff[rt] = static_cast<uint32_t>(prow < rows_per_slab ? prow : rows_per_slab) * static_cast<uint32_t>(row_bytes) + static_cast<uint32_t>(piece);
      const int pc = prow < rows_per_slab ? prow : rows_per_slab - 1;
      mbase[rt] = (static_cast<uint32_t>(sl * rows_per_slab + pc) * static_cast<uint32_t>(2 * S)) * 2u;   // byte offset of the output row's group constants
    }
#pragma unroll
    for (int t = 0; t < MT; ++t)
      xvoff[t] = static_cast<uint32_t>(16 * t + R) * static_cast<uint32_t>(K) * 2u + static_cast<uint32_t>(NBITS == 2 ? c * 16 : c * 32);
  }

  // ---- lane constants of the lane's slab ----
  KwLane lc;
  {
    const int sh = W3 ? 0 : NBITS * (PER - 1 - sl);
    const uint32_t m1 = ((NBITS == 8) ? 0xFFu : ((1u << NBITS) - 1u)) << sh;
    lc.mask = BF16 ? m1 * 0x01010101u : (m1 | (m1 << 16));
    const uint16_t e_dn = static_cast<uint16_t>((15 - sh) << 10);   // 2^-sh
    const half_t dn = __builtin_bit_cast(half_t, e_dn);
    lc.k1 = half2_t{dn, dn};
    const half_t m1024 = static_cast<half_t>(-1024.0f);
    lc.k2 = half2_t{m1024 * dn, m1024 * dn};
    const int J = 9 - sh;
    const half_t fzv = __builtin_bit_cast(half_t, static_cast<uint16_t>((15 - J) << 10)), fsv = __builtin_bit_cast(half_t, static_cast<uint16_t>((15 + J) << 10));
    lc.fz = half2_t{fzv, fzv};
    lc.fs = half2_t{fsv, fsv};
    lc.inv = __uint_as_float(static_cast<uint32_t>(127 - sh) << 23);
    lc.slab1 = sl != 0;
  }
  uint32_t magic;
  asm volatile("v_mov_b32 %0, 0x64006400" : "=v"(magic));   // opaque to the optimiser: stays in a VGPR

  f32x4 acc[RT][MT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- a unit's requests: RT (8-bit: 2 RT) weight loads, then 4 MT loads of x.  Always the same number of instructions; a dead unit
  //      (step >= s1) reads through zero-length descriptors ----
  using Unit = KwUnit<NBITS, MT, RT>;
  auto issue = [&](Unit& un, int step) {
    const bool live = step < s1;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(Wq), 0, live ? w_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(a.x), 0, live ? x_bytes : 0u, 0x00020000);
    const int so_w = live ? step * (W3 ? 96 : 128) : 0, so_x = live ? step * 256 : 0;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      if constexpr (NBITS == 4) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rw, wvoff[rt], so_w, 2 /* nt */);
        un.w[rt][0] = v.x; un.w[rt][1] = v.y; un.w[rt][2] = v.z; un.w[rt][3] = v.w;
      } else if constexpr (W3) {
        const u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(rw, wvoff[rt], so_w, 2);
        un.w[rt][0] = v.x; un.w[rt][1] = v.y; un.w[rt][2] = v.z;
      } else if constexpr (NBITS == 2) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rw, wvoff[rt], so_w, 2);
        un.w[rt][0] = v.x; un.w[rt][1] = v.y;
      } else {
        const u32x4 v0 = __builtin_amdgcn_raw_buffer_load_b128(rw, wvoff[rt], so_w, 2);
        const u32x4 v1 = __builtin_amdgcn_raw_buffer_load_b128(rw, wvoff[rt] + 64u, so_w, 2);
        un.w[rt][0] = v0.x; un.w[rt][1] = v0.y; un.w[rt][2] = v0.z; un.w[rt][3] = v0.w;
        un.w[rt][4] = v1.x; un.w[rt][5] = v1.y; un.w[rt][6] = v1.z; un.w[rt][7] = v1.w;
      }
    }
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // fragment q of the step: 2-bit: k = 32 q + 8 c .. +7; else k = 64 (q >> 1) + 16 c + 8 (q & 1) .. +7
        const uint32_t qoff = NBITS == 2 ? q * 64u : (q >> 1) * 128u + (q & 1) * 16u;
        un.x[t][q] = __builtin_amdgcn_raw_buffer_load_b128(rx, xvoff[t] + qoff, so_x, 0);
      }
  };

  // ---- group constants: a round = four steps; lane (R, c) fetches the dword pair of step (round base + c) of its output row ----
  struct Meta { uint32_t z[RT], s[RT]; };
  auto issue_meta = [&](Meta& mt_, int base) {
    int st = base + c;
    st = st < S ? st : S - 1;   // (past the end of K: a valid dword, never used for a live step)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      mt_.z[rt] = __builtin_amdgcn_raw_buffer_load_b32(rz, mbase[rt] + static_cast<uint32_t>(st) * 4u, 0, 0);
      mt_.s[rt] = __builtin_amdgcn_raw_buffer_load_b32(rs, mbase[rt] + static_cast<uint32_t>(st) * 4u, 0, 0);
    }
  };

  // piece j of a step for this lane: the bytes lane (rp, h = j, c) loaded
  auto piece = [&](uint32_t v, int j) -> uint32_t {
    if constexpr (PER == 1) return v;
    else if constexpr (PER == 2) {
      // row_ror:8 — banks 2-3 (lanes 8-15 of a 16-lane row) take from lanes 0-7 (j = 0) or banks 0-1 from lanes 8-15 (j = 1); the rest keep their own
      return j == 0 ? static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(v), static_cast<int>(v), 0x128, 0xF, 0xC, false))
                    : static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(v), static_cast<int>(v), 0x128, 0xF, 0x3, false));
    } else {
      // ds_swizzle, bit-mask mode inside 32 lanes: source lane = (lane & 0b10011) | 4 j — same rp, same c, piece j
      switch (j) {
        case 0: return static_cast<uint32_t>(__builtin_amdgcn_ds_swizzle(static_cast<int>(v), 0x13 | (0 << 5)));
        case 1: return static_cast<uint32_t>(__builtin_amdgcn_ds_swizzle(static_cast<int>(v), 0x13 | (4 << 5)));
        case 2: return static_cast<uint32_t>(__builtin_amdgcn_ds_swizzle(static_cast<int>(v), 0x13 | (8 << 5)));
        default: return static_cast<uint32_t>(__builtin_amdgcn_ds_swizzle(static_cast<int>(v), 0x13 | (12 << 5)));
      }
    }
  };

  // ---- one step: rebuild the lane's slab of every tile (4 A fragments of 8 k each), contract with the 4 MT x fragments ----
  auto consume = [&](const Unit& un, const Meta& mt_, int jr /* step inside its round: which lanes' constants */) {
    frag_t B[MT][4];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) B[t][q] = __builtin_bit_cast(frag_t, W3 ? un.x[t][q] : kw_permute_x8(un.x[t][q]));
#ifdef KW_LAB_NOARITH   // lab (floor probe): every loaded register is touched, nothing is rebuilt or contracted
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      float f = 0.f;
#pragma unroll
      for (int d = 0; d < kw_wv(NBITS); ++d) f += __uint_as_float(un.w[rt][d] & 0x3F800000u);
      f += __uint_as_float((mt_.z[rt] ^ mt_.s[rt]) & 0x3F800000u);
#pragma unroll
      for (int t = 0; t < MT; ++t) {
#pragma unroll
        for (int q = 0; q < 4; ++q) f += __uint_as_float((un.x[t][q].x ^ un.x[t][q].y ^ un.x[t][q].z ^ un.x[t][q].w) & 0x3F800000u);
        acc[rt][t][0] += f;
      }
    }
    return;
#endif
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      // the step's (zero, zero') / (scale, scale') of the lane's output row sit in lane R + 16 jr
      const uint32_t zp = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute((R + 16 * jr) * 4, static_cast<int>(mt_.z[rt])));
      const uint32_t sp = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute((R + 16 * jr) * 4, static_cast<int>(mt_.s[rt])));
      uint32_t zq = zp, sq = sp;
      if constexpr (SUB && !BF16 && !W3) {
        zq = kw_u32(as_h2(zp) * lc.fz);   // exact for every group of a layer that passed hqq_hip_meta_check
        sq = kw_u32(as_h2(sp) * lc.fs);
      }
      uint32_t o[4][4];   // [fragment q][dword]
      if constexpr (NBITS == 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t w2[2] = {piece(un.w[rt][0], j), piece(un.w[rt][1], j)};
          const uint32_t zg = (j >> 1) ? zq >> 16 : zq & 0xFFFFu, sg = (j >> 1) ? sq >> 16 : sq & 0xFFFFu;
          if constexpr (BF16) kw_rebuild_bf16<2>(w2, zg, sg, lc, o[j]);
          else kw_rebuild_f16<2, SUB>(w2, as_h2(zg | (zg << 16)), as_h2(sg | (sg << 16)), lc, magic, o[j]);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const uint32_t zg = j ? zq >> 16 : zq & 0xFFFFu, sg = j ? sq >> 16 : sq & 0xFFFFu;
          uint32_t o8[8];
          if constexpr (W3) {
            kw_rebuild_w3s<BF16, SUB>(piece(un.w[rt][0], j), piece(un.w[rt][1], j), piece(un.w[rt][2], j), zg, sg, lc, magic, o8);
          } else {
            uint32_t w4[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) w4[d] = NBITS == 8 ? un.w[rt][4 * j + d] : piece(un.w[rt][d], j);
            if constexpr (BF16) kw_rebuild_bf16<4>(w4, zg, sg, lc, o8);
            else kw_rebuild_f16<4, SUB>(w4, as_h2(zg | (zg << 16)), as_h2(sg | (sg << 16)), lc, magic, o8);
          }
#pragma unroll
          for (int d = 0; d < 4; ++d) { o[2 * j][d] = o8[d]; o[2 * j + 1][d] = o8[4 + d]; }
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const frag_t A = __builtin_bit_cast(frag_t, u32x4{o[q][0], o[q][1], o[q][2], o[q][3]});
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          if constexpr (BF16) acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B[t][q], acc[rt][t], 0, 0, 0);
          else acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B[t][q], acc[rt][t], 0, 0, 0);
        }
      }
    }
  };

  // ---- prologue: the first round's constants, D units; then rounds of four steps: request the NEXT round's constants, and per step
  //      consume a unit and refill it D steps ahead.  (sched_barrier: left alone the scheduler lifts the first instructions of the next
  //      consume — and the wait they drag along — in front of the requests.) ----
  Unit un[D];
  Meta mcur, mnext;
  issue_meta(mcur, s0);
#pragma unroll
  for (int k = 0; k < D; ++k) issue(un[k], s0 + k);
  __builtin_amdgcn_sched_barrier(0);
  for (int i = s0; i < s1; i += 4) {
    issue_meta(mnext, i + 4);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      consume(un[k % D], mcur, k);   // (steps past s1: zero weights x zero activations on valid constants: adds exactly 0)
      __builtin_amdgcn_sched_barrier(0);
      issue(un[k % D], i + k + D);
      __builtin_amdgcn_sched_barrier(0);
    }
    mcur = mnext;
  }

  // ---- the eight waves' partial tiles meet in LDS; output (tile, token, row) sums them in wave order, rounds once, stores ----
  f32x4* red = reinterpret_cast<f32x4*>(smem);   // [wave][rt][t][lane]
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int t = 0; t < MT; ++t) red[((wave * RT + rt) * MT + t) * 64 + lane] = acc[rt][t];
  __syncthreads();
  // bias / y of the layer: one indexed scalar load each, here at the end
  const half_t* bias = a.bias[li];
  half_t* y = a.y[li];
  const float* redf = reinterpret_cast<const float*>(smem);
  constexpr int NOUT = RT * MT * 256;
#pragma unroll
  for (int it = 0; it < (NOUT + KW_T - 1) / KW_T; ++it) {
    const int oi = it * KW_T + tid;
    if (NOUT % KW_T != 0 && oi >= NOUT) break;
    const int tile = oi >> 8, e = oi & 255, tok = e >> 4, Rr = e & 15;   // D layout: element (row Rr, column tok) sits in lane tok + 16 (Rr >> 2), register Rr & 3
    const int src = (tile * 64 + tok + 16 * (Rr >> 2)) * 4 + (Rr & 3);
    float sum = redf[src];
#pragma unroll
    for (int w = 1; w < KW_WAVES; ++w) sum += redf[w * (RT * MT * 256) + src];
    const int rt = tile / MT, t = tile % MT;
    const int m = 16 * t + tok;
    const int prow = (tile0 + rt) * RP + (Rr % RP);
    if (m < M && prow < rows_per_slab) {
      const int n = (Rr / RP) * rows_per_slab + prow;
      if constexpr (BF16) {
        uint16_t o = f32_to_bf16(sum);
        if (bias) o = f32_to_bf16(bf16_to_f32(o) + bf16_to_f32(reinterpret_cast<const uint16_t*>(bias)[n]));
        reinterpret_cast<uint16_t*>(y)[static_cast<int64_t>(m) * N + n] = o;
      } else {
        half_t o = static_cast<half_t>(sum);
        if (bias) o = o + bias[n];   // `out += bias` on the rounded matmul result (quantize.py:896-897)
        y[static_cast<int64_t>(m) * N + n] = o;
      }
    }
  }
}

// ---- host side ----------------------------------------------------------------------------------------------------------
static int kw_num_cus() {
  static int n_cus = 0;
  if (n_cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) n_cus = n;
    else n_cus = 256;
  }
  return n_cus;
}

// tiles per workgroup: the candidate with the least (rounds of workgroups over the CUs) x (bytes a workgroup pulls through its L1:
// RT tiles of weights + constants, and all of x).  Speed only: no sum depends on it.  forced: HQQ_OPT_SKINNY_KS(n) (tuning).
static int kw_choose_rt(int nbits, int n_layers, const int64_t* N, int64_t M, int64_t K, int forced) {
  const int per = kw_per(nbits), rp = 16 / per, mt = static_cast<int>((M + 15) / 16);
  const int cand[5] = {1, 2, 3, 4, 6};
  const int cus = kw_num_cus();
  const double wtile = static_cast<double>(rp) * static_cast<double>(K) * kw_row_bytes_num(nbits) / 4.0 + 16.0 * (K / 64) * 4.0;
  const double xb = static_cast<double>(M) * static_cast<double>(K) * 2.0;
  int best = 1;
  double best_cost = 0;
  for (int ci = 0; ci < 5; ++ci) {
    const int rt = cand[ci];
    if (!kw_combo(nbits, mt, rt)) continue;
    if (forced == rt) return rt;
    int64_t wgs = 0;
    for (int i = 0; i < n_layers; ++i) { const int64_t tiles = (N[i] / per + rp - 1) / rp; wgs += (tiles + rt - 1) / rt; }
    const double rounds = static_cast<double>((wgs + cus - 1) / cus);
    const double cost = rounds * (rt * wtile + xb);
    if (ci == 0 || cost < best_cost) { best = rt; best_cost = cost; }
  }
  return best;
}

template <int NBITS, int MT, int RT, bool BF16, bool SUB>
static int kw_launch1(const KwArgs& a, int wgs, hipStream_t st) {
  auto kern = kwave_kernel<NBITS, MT, RT, BF16, SUB>;
  const size_t lds = static_cast<size_t>(KW_WAVES) * RT * MT * 64 * sizeof(f32x4);
  if (lds > 64 * 1024) {
    static LdsRaised raised;
    if (const int rc = raise_lds_limit(raised, reinterpret_cast<const void*>(kern), 160 * 1024, "hqq_hip_gemv")) return rc;
  }
  hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(wgs)), dim3(KW_T), lds, st, a);
  return check_launch("hqq_hip_gemv");
}

template <int NBITS, bool BF16, bool SUB>
static int kw_launch(const KwArgs& a, int rt, int wgs, hipStream_t st) {
  const int mt = (a.M + 15) / 16;
#define KW_CASE(MT_, RT_) if constexpr (kw_combo(NBITS, MT_, RT_)) { if (mt == MT_ && rt == RT_) return kw_launch1<NBITS, MT_, RT_, BF16, SUB>(a, wgs, st); }
  KW_CASE(1, 1) KW_CASE(1, 2) KW_CASE(1, 3) KW_CASE(1, 4) KW_CASE(1, 6)
  KW_CASE(2, 1) KW_CASE(2, 2) KW_CASE(2, 3) KW_CASE(2, 4) KW_CASE(2, 6)
  KW_CASE(3, 1) KW_CASE(3, 2) KW_CASE(3, 3) KW_CASE(3, 4)
  KW_CASE(4, 1) KW_CASE(4, 2) KW_CASE(4, 3)
#undef KW_CASE
  set_error("hqq_hip_gemv: no batched-decode kernel for %d m-tiles x %d row tiles", mt, rt);
  return HQQ_ERR_SHAPE;
}

#define KW_CAT2(a, b) a##b
#define KW_CAT(a, b) KW_CAT2(a, b)
// one object per bit width (Makefile): kwave_run_8 / _4 / _3 / _2
int KW_CAT(kwave_run_, KW_NBITS)(const KwArgs& a, int rt, int wgs, int dtype, uint32_t opts, hipStream_t st) {
  if (dtype == HQQ_BF16) return kw_launch<KW_NBITS, true, false>(a, rt, wgs, st);
  if (opts & HQQ_OPT_META_SCALABLE) return kw_launch<KW_NBITS, false, true>(a, rt, wgs, st);
  return kw_launch<KW_NBITS, false, false>(a, rt, wgs, st);
}

#if KW_NBITS == 4
int kwave_run_8(const KwArgs& a, int rt, int wgs, int dtype, uint32_t opts, hipStream_t st);
int kwave_run_3(const KwArgs& a, int rt, int wgs, int dtype, uint32_t opts, hipStream_t st);
int kwave_run_2(const KwArgs& a, int rt, int wgs, int dtype, uint32_t opts, hipStream_t st);
#endif

}  // namespace kw

#if KW_NBITS == 4
// shapes the kernel covers (nbits = 3: the stream layout).  Everything else stays where it was (skinny.hip / gemv_mfma.hip).
bool kwave_covers(int nbits, int64_t M, int64_t K, int64_t group_size, const int64_t* N, int n_layers) {
  if ((nbits != 8 && nbits != 4 && nbits != 3 && nbits != 2) || group_size != 64 || M < 5 || M > 64 || K % kw::KW_KSTEP != 0 || K < kw::KW_KSTEP * kw::KW_WAVES) return false;
  const int per = kw::kw_per(nbits);
  for (int i = 0; i < n_layers; ++i) {
    if (N[i] % per != 0 || N[i] / per < 1) return false;
    // 32-bit byte offsets inside a layer and inside x
    if ((N[i] / per + 16) * (K / 4 * kw::kw_row_bytes_num(nbits)) > static_cast<int64_t>(UINT32_MAX) || N[i] * (K / 64) * 2 > static_cast<int64_t>(UINT32_MAX)) return false;
  }
  return M * K * 2 <= static_cast<int64_t>(UINT32_MAX);
}

int kwave_run(int nbits, int n_layers, const void* x, const void* const* Wq, const void* const* scale, const void* const* zero,
              const void* const* bias, void* const* y, const int64_t* N, int64_t M, int64_t K, int dtype, uint32_t opts, hipStream_t st) {
  using namespace kw;
  const int per = kw_per(nbits), rp = 16 / per;
  const int rt = kw_choose_rt(nbits, n_layers, N, M, K, static_cast<int>(opts >> 24));
  KwArgs a;
  int64_t wgs = 0;
  for (int i = 0; i < n_layers; ++i) {
    const int64_t tiles = (N[i] / per + rp - 1) / rp;
    wgs += (tiles + rt - 1) / rt;
    if (wgs > INT32_MAX / 2) { set_error("hqq_hip_gemv: size overflow"); return HQQ_ERR_SHAPE; }
    a.Wq[i] = static_cast<const uint8_t*>(Wq[i]);
    a.scale[i] = static_cast<const half_t*>(scale[i]);
    a.zero[i] = static_cast<const half_t*>(zero[i]);
    a.bias[i] = bias ? static_cast<const half_t*>(bias[i]) : nullptr;
    a.y[i] = static_cast<half_t*>(y[i]);
    a.N[i] = static_cast<int>(N[i]);
    a.wg_end[i] = static_cast<int>(wgs);
  }
  for (int i = n_layers; i < KW_MAXL; ++i) {
    a.Wq[i] = a.Wq[n_layers - 1]; a.scale[i] = a.scale[n_layers - 1]; a.zero[i] = a.zero[n_layers - 1]; a.bias[i] = a.bias[n_layers - 1];
    a.y[i] = a.y[n_layers - 1]; a.N[i] = a.N[n_layers - 1]; a.wg_end[i] = a.wg_end[n_layers - 1];
  }
  a.x = static_cast<const half_t*>(x);
  a.M = static_cast<int>(M);
  a.K = static_cast<int>(K);
  a.S = static_cast<int>(K / KW_KSTEP);
  switch (nbits) {
    case 8: return kwave_run_8(a, rt, static_cast<int>(wgs), dtype, opts, st);
    case 4: return kwave_run_4(a, rt, static_cast<int>(wgs), dtype, opts, st);
    case 3: return kwave_run_3(a, rt, static_cast<int>(wgs), dtype, opts, st);
    default: return kwave_run_2(a, rt, static_cast<int>(wgs), dtype, opts, st);
  }
}
#endif

}  // namespace hqq
