// block_math.h — element arithmetic of the decode step's glue (block.hip) with the roundings torch's elementwise ops make; shared with the
// decode kernels that fold those steps into their prologue / epilogue (gemv_block.hip).  Translation units that use it are compiled with
// -ffp-contract=off: a fused multiply-add would remove a rounding HF's separate ops make.
#pragma once
#include "hqq_common.h"

namespace hqq {

// element arithmetic on raw 16-bit values, with the roundings torch's elementwise ops make
template <bool BF>
struct El {
  static __device__ __forceinline__ float f(uint16_t a) {
    if constexpr (BF) return bf16_to_f32(a);
    else return static_cast<float>(__builtin_bit_cast(half_t, a));
  }
  static __device__ __forceinline__ uint16_t r(float v) {   // round to nearest even
    if constexpr (BF) return f32_to_bf16(v);
    else return __builtin_bit_cast(uint16_t, static_cast<half_t>(v));
  }
  static __device__ __forceinline__ uint16_t add(uint16_t a, uint16_t b) {
    if constexpr (BF) return f32_to_bf16(bf16_to_f32(a) + bf16_to_f32(b));
    else { const half_t s = __builtin_bit_cast(half_t, a) + __builtin_bit_cast(half_t, b); return __builtin_bit_cast(uint16_t, s); }
  }
  static __device__ __forceinline__ uint16_t mul(uint16_t a, uint16_t b) {
    if constexpr (BF) return f32_to_bf16(bf16_to_f32(a) * bf16_to_f32(b));
    else { const half_t s = __builtin_bit_cast(half_t, a) * __builtin_bit_cast(half_t, b); return __builtin_bit_cast(uint16_t, s); }
  }
  static __device__ __forceinline__ uint16_t neg(uint16_t a) { return static_cast<uint16_t>(a ^ 0x8000u); }
  // T(a * b) for an fp32 product the way torch's two ops round it: the product is an fp32 VALUE first (one rounding), the cast rounds it again.  Written
  // as `r(a * b)` hipcc folds the pair into ONE v_fma_mixlo_f16 where it can (not under SLP vectorisation, which takes v_pk_mul_f32 + v_cvt_pk_f16_f32): the
  // same source then rounds differently from one build to the next (round 6: -fno-slp-vectorize moved 1 normalised activation in ~16,000 by an ulp and a
  // tiny model's 27th greedy token with it).  The empty asm makes the product opaque: v_mul_f32, then the conversion.
  static __device__ __forceinline__ uint16_t r_prod(float a, float b) {
    float p = a * b;
    asm("" : "+v"(p));
    return r(p);
  }
};

// q_embed = (q * cos) + (rotate_half(q) * sin), rotate_half = cat(-x2, x1): every product and the sum round to T
template <bool BF>
__device__ __forceinline__ void rope_pair(uint16_t x1, uint16_t x2, uint16_t c1, uint16_t c2, uint16_t s1, uint16_t s2, uint16_t& o1, uint16_t& o2) {
  using E = El<BF>;
  o1 = E::add(E::mul(x1, c1), E::mul(E::neg(x2), s1));
  o2 = E::add(E::mul(x2, c2), E::mul(x1, s2));
}

// LlamaMLP: act_fn(gate) * up — silu in fp32 (x / (1 + exp(-x))), rounded to T, then the product in T (transformers models/llama/modeling_llama.py LlamaMLP.forward)
template <bool BF>
__device__ __forceinline__ uint16_t silu_mul_el(uint16_t gate, uint16_t up) {
  using E = El<BF>;
  const float x = E::f(gate);
  return E::mul(E::r(x / (1.0f + expf(-x))), up);
}

}  // namespace hqq
