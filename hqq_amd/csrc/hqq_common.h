// hqq_common.h — shared device/host helpers for the gfx950 HQQ kernels (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/hqq_hip.h"

namespace hqq {

using half_t = _Float16;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct bf16_t { uint16_t v; };

// ---- error plumbing (host) -------------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);
// hipLaunchKernelGGL reports through the sticky last-error slot: drop whatever an earlier, unrelated failure (e.g. an invalidated
// stream capture in the caller) left there, so that check_launch() reports this call's launch only
static inline void clear_stale_error() { (void)hipGetLastError(); }

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

static inline int per_of(int nbits) {
  switch (nbits) { case 8: return 1; case 4: return 2; case 2: return 4; case 1: return 8; case 3: return 10; default: return 0; }
}

// ---- bf16 <-> f32, round-to-nearest-even (device) -------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float(static_cast<uint32_t>(h) << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
  uint32_t x = __float_as_uint(f);
  if ((x & 0x7FFFFFFFu) > 0x7F800000u) return static_cast<uint16_t>((x >> 16) | 0x40u);
  x += 0x7FFFu + ((x >> 16) & 1u);
  return static_cast<uint16_t>(x >> 16);
}

// ---- one "compute dtype" arithmetic policy per element type ---------------------------------------
// dequant(q, z, s) = round_T(round_T(q - z) * s): the two roundings of Quantizer.dequantize
// (hqq/core/quantize.py:198) done in the tensor dtype.
template <typename T> struct CD;

template <> struct CD<float> {
  using store_t = float;
  static __device__ __forceinline__ float load(const float* p, int64_t i) { return p[i]; }
  static __device__ __forceinline__ float dequant(float q, float z, float s) { float d = q - z; return d * s; }
  static __device__ __forceinline__ float from_f32(float v) { return v; }
  static __device__ __forceinline__ float to_f32(float v) { return v; }
};
template <> struct CD<half_t> {
  using store_t = half_t;
  static __device__ __forceinline__ half_t load(const half_t* p, int64_t i) { return p[i]; }
  static __device__ __forceinline__ half_t dequant(float q, half_t z, half_t s) {
    half_t d = static_cast<half_t>(q) - z;   // v_sub_f16, RNE
    return d * s;                            // v_mul_f16, RNE
  }
  static __device__ __forceinline__ half_t from_f32(float v) { return static_cast<half_t>(v); }
  static __device__ __forceinline__ float to_f32(half_t v) { return static_cast<float>(v); }
};
template <> struct CD<bf16_t> {
  using store_t = uint16_t;
  static __device__ __forceinline__ bf16_t load(const bf16_t* p, int64_t i) { return p[i]; }
  static __device__ __forceinline__ bf16_t dequant(float q, bf16_t z, bf16_t s) {
    // q - z is exact in float (q small integer, z has 8 significant bits); one rounding to bf16
    float d = bf16_to_f32(f32_to_bf16(q - bf16_to_f32(z.v)));
    // product of two bf16 values is exact in float; one rounding to bf16
    return bf16_t{f32_to_bf16(d * bf16_to_f32(s.v))};
  }
  static __device__ __forceinline__ bf16_t from_f32(float v) { return bf16_t{f32_to_bf16(v)}; }
  static __device__ __forceinline__ float to_f32(bf16_t v) { return bf16_to_f32(v.v); }
};

// Layout of the caller-owned decode workspace (hqq_hip_gemv_workspace_bytes): [arrival counters | fp32 partial sums].  The head holds
// one int per (panel, row group) of a split-K launch (skinny.hip) and must be zero whenever a call starts — every kernel that uses
// counters leaves them zero — so kernels that only park partial sums (gemv3s.hip) keep out of it.
constexpr size_t WS_COUNTER_BYTES = size_t(256) << 10;

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace hqq
