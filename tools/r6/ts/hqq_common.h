// hqq_common.h — shared device/host helpers for the gfx950 HQQ kernels (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <type_traits>

#include "../../../include/hqq_hip.h"

namespace hqq {

using half_t = _Float16;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct bf16_t { uint16_t v; };

// ---- error plumbing (host) -------------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);
// hipLaunchKernelGGL reports through the sticky last-error slot: drop whatever an earlier, unrelated failure (e.g. an invalidated
// stream capture in the caller) left there, so that check_launch() reports this call's launch only
static inline void clear_stale_error() { (void)hipGetLastError(); }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) takes effect on the CURRENT device only: one flag per device and call site (a
// process-wide flag left the second GPU of a process without the raised limit: its first launch above 64 KiB of LDS failed)
struct LdsRaised { bool done[64] = {}; };
static inline int raise_lds_limit(LdsRaised& st, const void* kern, int bytes, const char* who) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = -1; }
  if (dev >= 0 && dev < 64 && st.done[dev]) return 0;
  hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) { set_error("%s: cannot raise the dynamic LDS limit: %s", who, hipGetErrorString(e)); return static_cast<int>(e); }
  if (dev >= 0 && dev < 64) st.done[dev] = true;
  return 0;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

static inline int per_of(int nbits) {
  switch (nbits) { case 8: return 1; case 4: return 2; case 2: return 4; case 1: return 8; case 3: return 10; default: return 0; }
}

// ---- bf16 <-> f32, round-to-nearest-even (device) -------------------------------------------------
// select between VALUES (arguments by value).  `c ? x : y` on two lvalues is itself an lvalue: the compiler selects the ADDRESS and
// loads through it — a dependent scalar load (kernel-argument struct) or a scratch access (local struct) per select.
template <class T>
__device__ __forceinline__ T pick(bool c, T x, T y) { return c ? x : y; }

// raw buffer descriptor over a whole allocation: base pointer, stride 0, no bound (offsets stay below 4 GiB per layer: checked on the host)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buffer_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, -1, 0x00020000);
}

// Sum of the diagonal of a 16x16 MFMA accumulator tile, the same value in every lane.  D[i][j] sits in lane j + 16 (i / 4), register
// i % 4, so the diagonal is register i of lane 20 r + i (r = 0..3): three DPP adds fold a quad's (reg0, reg1, reg2, reg3) of lanes
// (0, 1, 2, 3) into its lane 0 — no per-lane selects —, the four quads that hold diagonal elements are read out and added.
// Association: ((d0 + d1) + (d2 + d3)) per quad, (q0 + q1) + (q2 + q3) across — the order wave_sum() gives the masked tile.
__device__ __forceinline__ float diag_sum(const f32x4& t) {
  auto dpp = [](float x, auto ctrl) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xF, 0xF, true));
  };
  const float a = t[0] + dpp(t[1], std::integral_constant<int, 0x55>{});   // quad_perm [1,1,1,1]: lane 0 of the quad: reg0 + lane 1's reg1
  const float b = t[2] + dpp(t[3], std::integral_constant<int, 0xFF>{});   // quad_perm [3,3,3,3]: lane 2: reg2 + lane 3's reg3
  const float w = a + dpp(b, std::integral_constant<int, 0xAA>{});         // quad_perm [2,2,2,2]: lane 0: a + lane 2's b
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w), 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w), 20));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w), 40));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w), 60));
  return (r0 + r1) + (r2 + r3);
}

// The same sum, valid in LANE 63 ONLY, without leaving the vector side (round 6): the quad fold towards the quad's lane 3, everything but the four diagonal
// quads masked to zero, two row shifts bring a row's value to its lane 15, row_bcast:15 / row_bcast:31 add the rows into lane 63.  8 VALU and no
// v_readlane (diag_sum: 3 DPP adds + 4 v_readlane + 5 VALU).  Same association — ((d0 + d1) + (d2 + d3)) per quad, (q0 + q1) + (q2 + q3) across, with the
// operands of each addition swapped — hence the same bits; the zeros added on the way are exact (an accumulator that starts at +0 never holds -0).
template <int NV>
__device__ __forceinline__ void diag_sum63(const f32x4 (&t)[NV], float (&out)[NV], bool diag_lane /* lane == 3, 23, 43, 63 */) {
  auto dpp = [](float x, auto ctrl) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xF, 0xF, true));
  };
  // stage by stage over the NV tiles: every DPP operation needs its source two instructions old, and NV independent chains fill those slots with each other
  // (one chain after the other the compiler pads each step with s_nop)
  float a[NV], b[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) a[i] = t[i][1] + dpp(t[i][0], std::integral_constant<int, 0x00>{});   // quad_perm [0,0,0,0]: lane 1 of the quad: reg1 + lane 0's reg0
#pragma unroll
  for (int i = 0; i < NV; ++i) b[i] = t[i][3] + dpp(t[i][2], std::integral_constant<int, 0xAA>{});   // quad_perm [2,2,2,2]: lane 3: reg3 + lane 2's reg2
#pragma unroll
  for (int i = 0; i < NV; ++i) a[i] = b[i] + dpp(a[i], std::integral_constant<int, 0x55>{});         // quad_perm [1,1,1,1]: lane 3: b + lane 1's a
#pragma unroll
  for (int i = 0; i < NV; ++i) a[i] = diag_lane ? a[i] : 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) a[i] = a[i] + dpp(a[i], std::integral_constant<int, 0x114>{});        // row_shr:4
#pragma unroll
  for (int i = 0; i < NV; ++i) a[i] = a[i] + dpp(a[i], std::integral_constant<int, 0x118>{});        // row_shr:8 -> lane 15 of a row: its diagonal quad's sum
  // (every row enabled: rows without a source add the zero bound_ctrl supplies; what the other lanes end up with does not matter — and the compiler can fuse the
  //  move into the add, which it cannot when some rows must keep an old value)
#pragma unroll
  for (int i = 0; i < NV; ++i) a[i] = a[i] + dpp(a[i], std::integral_constant<int, 0x142>{});        // row_bcast:15: row r += lane 15 of row r - 1
#pragma unroll
  for (int i = 0; i < NV; ++i) out[i] = a[i] + dpp(a[i], std::integral_constant<int, 0x143>{});      // row_bcast:31: rows 2, 3 += lane 31: lane 63 = (q3 + q2) + (q1 + q0)
}

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
