// decode_common.h — device pieces shared by the decode kernels (gemv.hip: one launch per layer group; engine.hip: the persistent
// cross-layer engine): the exact-weights rebuild of a 16-byte packed vector on the matrix core, the x staging order, wave reductions.
#pragma once
#include <type_traits>

#include "hqq_common.h"

namespace hqq {

// sum over the 64 lanes, result valid in every lane: four DPP adds inside each row of 16, then the four row totals
// through SGPRs.  No LDS traffic (a ds_bpermute butterfly costs ~100 cycles of latency per step).
__device__ __forceinline__ float wave_sum(float v) {
  auto dpp_add = [](float x, auto ctrl) {
    const int y = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xF, 0xF, true);
    return x + __builtin_bit_cast(float, y);
  };
  v = dpp_add(v, std::integral_constant<int, 0xB1>{});    // quad_perm [1,0,3,2]
  v = dpp_add(v, std::integral_constant<int, 0x4E>{});    // quad_perm [2,3,0,1]
  v = dpp_add(v, std::integral_constant<int, 0x141>{});   // row_half_mirror
  v = dpp_add(v, std::integral_constant<int, 0x140>{});   // row_mirror
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return (r0 + r1) + (r2 + r3);
}

__device__ __forceinline__ half2_t as_h2(uint32_t u) { return __builtin_bit_cast(half2_t, u); }

// biased levels of slab S for the byte pairs (b0,b2) [word] / (b1,b3) [word >> 8] of one packed dword: the fp16 pair
// (1024 + F q, 1024 + F q'), F = 2^shift(S) — the masked nibbles OR-ed onto the exponent 0x6400, one VALU op.
template <int NBITS, int S>
__device__ __forceinline__ half2_t biased_levels(uint32_t word_or_shifted, uint32_t magic /* 0x64006400 held in a VGPR */) {
  constexpr int per = 8 / NBITS;
  constexpr int sh = NBITS * (per - 1 - S);
  constexpr uint32_t m1 = ((NBITS == 8) ? 0xFFu : ((1u << NBITS) - 1u)) << sh;
  constexpr uint32_t m = m1 | (m1 << 16);
  uint32_t b;
  // hipcc emits v_and + v_or for (w & m) | magic (GFX9 VOP3 takes no literals); the mask rides in an SGPR here
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(b) : "v"(word_or_shifted), "s"(m), "v"(magic));
  return as_h2(b);
}
template <int NBITS, int S> struct SlabF {   // F and 1/F of slab S
  static constexpr int sh = NBITS * (8 / NBITS - 1 - S);
  static constexpr float F = static_cast<float>(1 << sh);
  static constexpr float invF = 1.0f / static_cast<float>(1 << sh);
};

// x staging order: a lane's chunk of 16 k-values is kept as two 16-byte planes (conflict-free ds_read_b128);
// inside a plane the 8 halfs are (k0,k2,k1,k3,k4,k6,k5,k7) so that half2 j pairs with the levels<> of byte pair j.
__device__ __forceinline__ u32x4 permute_x8(u32x4 v) {
  u32x4 r;
  r.x = (v.x & 0xFFFFu) | (v.y << 16);
  r.y = (v.x >> 16) | (v.y & 0xFFFF0000u);
  r.z = (v.z & 0xFFFFu) | (v.w << 16);
  r.w = (v.z >> 16) | (v.w & 0xFFFF0000u);
  return r;
}

// EXACT mode: the lane's 16 weights of slab S are rebuilt exactly as Quantizer.dequantize does (two fp16 roundings, 4 packed
// ops per pair) and contracted on the matrix core.  With all 64 lanes holding the SAME output row, lane l = (i = l & 15,
// o = l >> 4) supplies row i / k-octet o of the A operand and column i / k-octet o of the B operand (its own 8 x-values):
// D[i][i] is then the partial dot product of the four lanes {i + 16 o}, and the row's result is the sum of the diagonal.
// 15/16 of the MFMA's flops are discarded — the matrix pipe is idle otherwise — but no VALU slot is spent on the dot
// product, which keeps the kernel under the VALU ceiling (see gemv_mfma.hip for the rates).
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
// SUB = true — the same two roundings in three VALU ops per weight pair instead of four.  The masked field of slab S,
// w & (m << sh), read as fp16 IS the subnormal q * 2^(sh - 24): no exponent has to be OR-ed on and no bias taken off again.
// One v_pk_fma lifts it by 2^15 and subtracts the zero-point in the same (single) rounding,
//     d' = round16(q * 2^-J - z * 2^-J) = round16(q - z) * 2^-J,   J = 9 - sh,
// and one v_pk_mul by s * 2^J gives round16(d * s).  Scaling by a power of two commutes with round-to-nearest as long as nothing
// leaves the fp16 range on the way: z * 2^-J must be exact (then every q - z that lands in the subnormal range is itself a
// multiple of 2^-24 * 2^J and exact on both sides) and s * 2^J finite.  hqq_hip_meta_check tests exactly that, per layer, once;
// layers that pass are launched with HQQ_OPT_META_SCALABLE, all others keep the four-op sequence below.
template <int NBITS, int M, int S, int PER, bool SUB>
struct SlabExact {
  static __device__ __forceinline__ void run(const u32x4& w, const uint32_t (&zs)[PER], const h8_t (&b0)[M], const h8_t (&b1)[M],
                                             f32x4 (&acc)[M][PER], uint32_t magic) {
    constexpr int sh = NBITS * (PER - 1 - S);
    const half2_t pr = as_h2(zs[S]);
    const half2_t zz = {pr.x, pr.x}, ss = {pr.y, pr.y};
    half2_t q[8];
    uint32_t o[8];
    if constexpr (SUB) {
      constexpr uint32_t m1 = ((NBITS == 8) ? 0xFFu : ((1u << NBITS) - 1u)) << sh;
      constexpr uint32_t m = m1 | (m1 << 16);
      const half2_t lift = {static_cast<half_t>(32768.0f), static_cast<half_t>(32768.0f)};
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        q[2 * d] = as_h2(w[d] & m);              // bytes (4d+0, 4d+2): q * 2^(sh-24), a subnormal pair
        q[2 * d + 1] = as_h2((w[d] >> 8) & m);   // bytes (4d+1, 4d+3)
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) q[i] = __builtin_elementwise_fma(q[i], lift, -zz);   // rounding 1 (zz = z * 2^-J)
    } else {
      constexpr float inv = 1.0f / static_cast<float>(1 << sh);
      const half2_t k1 = {static_cast<half_t>(inv), static_cast<half_t>(inv)};
      const half2_t k2 = {static_cast<half_t>(-1024.0f * inv), static_cast<half_t>(-1024.0f * inv)};
      // stage-wise over the eight weight pairs (not pair by pair): eight independent chains keep the packed-fp16 pipe busy
      // instead of stalling on each fma -> add -> mul dependency
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        q[2 * d] = biased_levels<NBITS, S>(w[d], magic);            // bytes (4d+0, 4d+2)
        q[2 * d + 1] = biased_levels<NBITS, S>(w[d] >> 8, magic);   // bytes (4d+1, 4d+3)
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) q[i] = __builtin_elementwise_fma(q[i], k1, k2);   // exact integer level
#pragma unroll
      for (int i = 0; i < 8; ++i) q[i] = q[i] - zz;                                  // rounding 1
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = __builtin_bit_cast(uint32_t, q[i] * ss);    // rounding 2 (SUB: ss = s * 2^J)
    const h8_t a0 = __builtin_bit_cast(h8_t, u32x4{o[0], o[1], o[2], o[3]});
    const h8_t a1 = __builtin_bit_cast(h8_t, u32x4{o[4], o[5], o[6], o[7]});
#pragma unroll
    for (int m = 0; m < M; ++m) {
      acc[m][S] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0[m], acc[m][S], 0, 0, 0);
      acc[m][S] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1[m], acc[m][S], 0, 0, 0);
    }
    if constexpr (S + 1 < PER) SlabExact<NBITS, M, S + 1, PER, SUB>::run(w, zs, b0, b1, acc, magic);
  }
};
// The rebuild half of SlabExact on its own: the lane's 16 weights of every slab as MFMA A operands (out[s][0] = k 0..7 of the
// lane's chunk, out[s][1] = k 8..15), nothing contracted yet.  gemv_chain.hip rebuilds a launch's first units while the activation
// row they will meet is still being produced by the previous launch.  Same operations in the same order as SlabExact: same bits.
template <int NBITS, int S, int PER, bool SUB>
struct SlabRebuild {
  static __device__ __forceinline__ void run(const u32x4& w, const uint32_t (&zs)[PER], h8_t (&out)[PER][2], uint32_t magic) {
    constexpr int sh = NBITS * (PER - 1 - S);
    const half2_t pr = as_h2(zs[S]);
    const half2_t zz = {pr.x, pr.x}, ss = {pr.y, pr.y};
    half2_t q[8];
    uint32_t o[8];
    if constexpr (SUB) {
      constexpr uint32_t m1 = ((NBITS == 8) ? 0xFFu : ((1u << NBITS) - 1u)) << sh;
      constexpr uint32_t m = m1 | (m1 << 16);
      const half2_t lift = {static_cast<half_t>(32768.0f), static_cast<half_t>(32768.0f)};
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        q[2 * d] = as_h2(w[d] & m);
        q[2 * d + 1] = as_h2((w[d] >> 8) & m);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) q[i] = __builtin_elementwise_fma(q[i], lift, -zz);   // rounding 1 (zz = z * 2^-J)
    } else {
      constexpr float inv = 1.0f / static_cast<float>(1 << sh);
      const half2_t k1 = {static_cast<half_t>(inv), static_cast<half_t>(inv)};
      const half2_t k2 = {static_cast<half_t>(-1024.0f * inv), static_cast<half_t>(-1024.0f * inv)};
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        q[2 * d] = biased_levels<NBITS, S>(w[d], magic);
        q[2 * d + 1] = biased_levels<NBITS, S>(w[d] >> 8, magic);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) q[i] = __builtin_elementwise_fma(q[i], k1, k2);   // exact integer level
#pragma unroll
      for (int i = 0; i < 8; ++i) q[i] = q[i] - zz;                                  // rounding 1
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = __builtin_bit_cast(uint32_t, q[i] * ss);    // rounding 2 (SUB: ss = s * 2^J)
    out[S][0] = __builtin_bit_cast(h8_t, u32x4{o[0], o[1], o[2], o[3]});
    out[S][1] = __builtin_bit_cast(h8_t, u32x4{o[4], o[5], o[6], o[7]});
    if constexpr (S + 1 < PER) SlabRebuild<NBITS, S + 1, PER, SUB>::run(w, zs, out, magic);
  }
};
// (z, s) of slab `slab` as fetched -> (z * 2^-J, s * 2^J): one packed multiply on the fetching lane, before the hand-round
// (`slab` is a constant after unrolling)
template <int NBITS>
__device__ __forceinline__ uint32_t scale_meta_sub(uint32_t zs_raw, int slab) {
  const int J = 9 - NBITS * (8 / NBITS - 1 - slab);
  const half2_t f = {static_cast<half_t>(1.0f / static_cast<float>(1 << J)), static_cast<half_t>(static_cast<float>(1 << J))};
  return __builtin_bit_cast(uint32_t, as_h2(zs_raw) * f);
}

// bf16 compute dtype: the reference's two roundings are to bf16 (quantize.py:198 on bf16 tensors).  gfx950 has no packed
// bf16 arithmetic, so the weight goes through fp32: v_cvt_f32_ubyteN lifts the masked byte F q, one fma forms q - z,
// v_cvt_pk_bf16_f32 rounds it (RNE), v_dot2_f32_bf16 against (s, 0) / (0, s) forms the exact product with s, a second v_cvt_pk rounds again.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
template <int B>
__device__ __forceinline__ float ubyte_f32(uint32_t v) { return static_cast<float>((v >> (8 * B)) & 0xFFu); }   // v_cvt_f32_ubyteB
template <int NBITS, int M, int S, int PER>
struct SlabExactBF16 {
  static __device__ __forceinline__ void run(const u32x4& w, const uint32_t (&zs)[PER], const bf16x8_t (&b0)[M], const bf16x8_t (&b1)[M],
                                             f32x4 (&acc)[M][PER], uint32_t magic) {
    constexpr int sh = NBITS * (PER - 1 - S);
    constexpr float inv = 1.0f / static_cast<float>(1 << sh);
    const float zf = __uint_as_float(zs[S] << 16);
    const bf16x2_t s_lo = __builtin_bit_cast(bf16x2_t, zs[S] >> 16);          // (s, 0)
    const bf16x2_t s_hi = __builtin_bit_cast(bf16x2_t, zs[S] & 0xFFFF0000u);  // (0, s)
    constexpr uint32_t m1 = ((1u << NBITS) - 1u) << sh;
    uint32_t o[8];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const uint32_t fq = NBITS == 8 ? w[d] : (w[d] & (m1 * 0x01010101u));   // the word's four bytes reduced to slab S's field (F q each)
      // fma(F q, 1 / F, -z) is q - z with ONE fp32 rounding (none unless z is below 2^-15): a bias folded into the addend
      // (-(1024 / F) - z) would itself round when z is small and cost an ulp after rounding 1
      const f32x2_t dq[2] = {{__builtin_fmaf(ubyte_f32<0>(fq), inv, -zf), __builtin_fmaf(ubyte_f32<2>(fq), inv, -zf)},    // bytes (4d+0, 4d+2)
                             {__builtin_fmaf(ubyte_f32<1>(fq), inv, -zf), __builtin_fmaf(ubyte_f32<3>(fq), inv, -zf)}};   // bytes (4d+1, 4d+3)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const bf16x2_t dr = __builtin_convertvector(dq[h], bf16x2_t);                 // rounding 1
        const f32x2_t pw = {__builtin_amdgcn_fdot2_f32_bf16(dr, s_lo, 0.f, false), __builtin_amdgcn_fdot2_f32_bf16(dr, s_hi, 0.f, false)};
        o[2 * d + h] = __builtin_bit_cast(uint32_t, __builtin_convertvector(pw, bf16x2_t));   // rounding 2
      }
    }
    const bf16x8_t a0 = __builtin_bit_cast(bf16x8_t, u32x4{o[0], o[1], o[2], o[3]});
    const bf16x8_t a1 = __builtin_bit_cast(bf16x8_t, u32x4{o[4], o[5], o[6], o[7]});
#pragma unroll
    for (int m = 0; m < M; ++m) {
      acc[m][S] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0[m], acc[m][S], 0, 0, 0);
      acc[m][S] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1[m], acc[m][S], 0, 0, 0);
    }
    if constexpr (S + 1 < PER) SlabExactBF16<NBITS, M, S + 1, PER>::run(w, zs, b0, b1, acc, magic);
  }
};

}  // namespace hqq
