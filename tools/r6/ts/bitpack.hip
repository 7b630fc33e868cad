// bitpack.hip — BitPack.pack_* / unpack_* and Quantizer.dequantize for gfx950.
//
// Reference semantics: hqq/core/bitpack.py:14-144, hqq/core/quantize.py:183-199; supersedes the
// axis=0-only CUDA kernels hqq/kernels/hqq_aten_cuda_kernel.cu:35-428 (1 thread per packed byte, scalar
// stores).  Here every lane moves 16 packed bytes (one global_load_dwordx4) and writes `per` dense
// 16/32/64-byte runs, so both directions stream at HBM rate; both axes are covered.
//
// Layout fact used throughout: the packed tensor is flat.  With n = packed element count
// (= step*cols), slab s of the unpacked matrix is the flat range [s*n, (s+1)*n), and packed element i
// holds unpacked elements {s*n + i}.  No 2-D indexing is needed for pack/unpack.
#include <type_traits>

#include "hqq_common.h"
#include "w3s.h"

namespace hqq {

template <int NBITS> struct Pk {
  static constexpr int per = (NBITS == 3) ? 10 : 8 / NBITS;
  static constexpr uint32_t mask = (NBITS == 8) ? 0xFFu : ((1u << NBITS) - 1u);
  // shift of slab s inside the container (slab 0 most significant)
  static __device__ __forceinline__ int shift(int s) { return (NBITS == 3) ? (27 - 3 * s) : NBITS * (per - 1 - s); }
};

// ---- input element readers for pack (uint8 levels, or float32 holding integer levels) ------------
__device__ __forceinline__ uint32_t level_of(uint8_t v) { return v; }
__device__ __forceinline__ uint32_t level_of(float v) { return static_cast<uint32_t>(static_cast<uint8_t>(static_cast<int>(v))); }

// =================================================================================================
// pack: u8 containers, VEC packed bytes per thread (16 when the slab size allows, else 1)
// =================================================================================================
template <int NBITS, typename IN, int VEC>
__global__ __launch_bounds__(256) void pack_u8_kernel(const IN* __restrict__ U, uint8_t* __restrict__ out, int64_t n) {
  using P = Pk<NBITS>;
  const int64_t i0 = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) * VEC;
  if (i0 >= n) return;
  uint8_t acc[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[j] = 0;
#pragma unroll
  for (int s = 0; s < P::per; ++s) {
    const IN* src = U + static_cast<int64_t>(s) * n + i0;
    IN v[VEC];
    if constexpr (VEC == 16 && sizeof(IN) == 1) {
      *reinterpret_cast<u32x4*>(v) = *reinterpret_cast<const u32x4*>(src);
    } else if constexpr (VEC == 16) {
#pragma unroll
      for (int q = 0; q < 4; ++q) reinterpret_cast<f32x4*>(v)[q] = reinterpret_cast<const f32x4*>(src)[q];
    } else {
      v[0] = src[0];
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j)  // uint8 `<<` in torch wraps modulo 256; the cast reproduces it
      acc[j] |= static_cast<uint8_t>(level_of(v[j]) << P::shift(s));
  }
  if constexpr (VEC == 16) *reinterpret_cast<u32x4*>(out + i0) = *reinterpret_cast<u32x4*>(acc);
  else out[i0] = acc[0];
}

// pack, uint8 levels, 16-byte chunks (round 5): dword arithmetic instead of sixteen byte-wise shifts — a byte's `<<` wraps modulo 256 in torch, i.e. per dword
// ((v & (0xFF >> shift) x 0x01010101) << shift) — and CH chunks per thread (256 threads apart: coalesced) so that at least four 16-byte loads are in flight per
// thread whatever the number of slabs (the 4-bit pack, two loads per thread, ran at 0.35 of the HBM roofline against 0.57 for the 2-bit one with four)
template <int NBITS, int CH>
__global__ __launch_bounds__(256) void pack_u8x16_kernel(const uint8_t* __restrict__ U, uint8_t* __restrict__ out, int64_t n) {
  using P = Pk<NBITS>;
  const int64_t base = (static_cast<int64_t>(blockIdx.x) * (256 * CH) + threadIdx.x) * 16;
  u32x4 v[CH][P::per];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int64_t i0 = base + static_cast<int64_t>(c) * 256 * 16;
#pragma unroll
    for (int s = 0; s < P::per; ++s)
      v[c][s] = i0 < n ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(U + static_cast<int64_t>(s) * n + i0)) : u32x4{0u, 0u, 0u, 0u};
  }
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int64_t i0 = base + static_cast<int64_t>(c) * 256 * 16;
    if (i0 >= n) continue;
    u32x4 acc{0u, 0u, 0u, 0u};
#pragma unroll
    for (int s = 0; s < P::per; ++s) {
      const int sh = NBITS * (P::per - 1 - s);
      const uint32_t keep = (0xFFu >> sh) * 0x01010101u;
#pragma unroll
      for (int d = 0; d < 4; ++d) acc[d] |= (v[c][s][d] & keep) << sh;
    }
    *reinterpret_cast<u32x4*>(out + i0) = acc;
  }
}

// pack: 3-bit into int32, VEC packed words per thread (16, 4 or 1); slabs past `total` are the zero padding.  VEC = 16: a slab's 16 levels in
// one 16-byte load (four for float levels) — n is a multiple of 64, so every slab's run starts 16-byte aligned — and four 16-byte stores
template <typename IN, int VEC>
__global__ __launch_bounds__(256) void pack_3bit_kernel(const IN* __restrict__ U, int32_t* __restrict__ out, int64_t n, int64_t total) {
  const int64_t i0 = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) * VEC;
  if (i0 >= n) return;
  uint32_t acc[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[j] = 0;
#pragma unroll
  for (int s = 0; s < 10; ++s) {
    const int64_t e0 = static_cast<int64_t>(s) * n + i0;
    if constexpr (VEC == 16) {
      if (e0 + 15 < total) {
        IN v[16];
        if constexpr (sizeof(IN) == 1) {
          *reinterpret_cast<u32x4*>(v) = *reinterpret_cast<const u32x4*>(U + e0);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) reinterpret_cast<f32x4*>(v)[q] = reinterpret_cast<const f32x4*>(U + e0)[q];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] |= level_of(v[j]) << (27 - 3 * s);
        continue;
      }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const uint32_t v = (e0 + j < total) ? level_of(U[e0 + j]) : 0u;
      acc[j] |= v << (27 - 3 * s);
    }
  }
  if constexpr (VEC == 16) {
#pragma unroll
    for (int q = 0; q < 4; ++q) reinterpret_cast<u32x4*>(out + i0)[q] = u32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
  } else {
#pragma unroll
    for (int j = 0; j < VEC; ++j) out[i0 + j] = static_cast<int32_t>(acc[j]);
  }
}

// =================================================================================================
// unpack / dequantize share one kernel: DEQ=false writes the integer levels converted to OUT,
// DEQ=true applies ((q - zero) * scale) in the compute dtype.  VEC packed containers per thread.
//   AXIS 1: meta index of unpacked element e is e / gs          (unpacked matrix [R, gs])
//   AXIS 0: meta index is e % cols_u, cols_u = total / gs         (unpacked matrix [gs, R])
// =================================================================================================
template <typename OUT> struct Conv;
template <> struct Conv<uint8_t> { static __device__ __forceinline__ uint8_t of(uint32_t q) { return static_cast<uint8_t>(q); } };
template <> struct Conv<float> { static __device__ __forceinline__ float of(uint32_t q) { return static_cast<float>(q); } };
template <> struct Conv<half_t> { static __device__ __forceinline__ half_t of(uint32_t q) { return static_cast<half_t>(static_cast<float>(q)); } };
template <> struct Conv<bf16_t> { static __device__ __forceinline__ bf16_t of(uint32_t q) { return bf16_t{f32_to_bf16(static_cast<float>(q))}; } };

template <typename OUT, int VEC>
__device__ __forceinline__ void store_vec(OUT* dst, const OUT* v) {
  constexpr int bytes = VEC * sizeof(OUT);
  if constexpr (bytes % 16 == 0) {
#pragma unroll
    for (int q = 0; q < bytes / 16; ++q) reinterpret_cast<u32x4*>(dst)[q] = reinterpret_cast<const u32x4*>(v)[q];
  } else if constexpr (bytes == 8) {
    *reinterpret_cast<u32x2*>(dst) = *reinterpret_cast<const u32x2*>(v);
  } else if constexpr (bytes == 4) {
    *reinterpret_cast<uint32_t*>(dst) = *reinterpret_cast<const uint32_t*>(v);
  } else {
#pragma unroll
    for (int j = 0; j < VEC; ++j) dst[j] = v[j];
  }
}

template <int NBITS, typename OUT, bool DEQ, int AXIS, int VEC>
__global__ __launch_bounds__(256) void unpack_kernel(const void* __restrict__ packed, const OUT* __restrict__ scale,
                                                     const OUT* __restrict__ zero, OUT* __restrict__ out,
                                                     int64_t n, int64_t limit, int64_t gs, int64_t cols_u) {
  using P = Pk<NBITS>;
  using CT = typename std::conditional<NBITS == 3, uint32_t, uint8_t>::type;
  const int64_t i0 = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) * VEC;
  if (i0 >= n) return;
  CT w[VEC];
  if constexpr (VEC * sizeof(CT) == 16) *reinterpret_cast<u32x4*>(w) = *reinterpret_cast<const u32x4*>(static_cast<const CT*>(packed) + i0);
  else w[0] = static_cast<const CT*>(packed)[i0];
  // sub-chunks over which the meta index is constant for AXIS 1: gs is a multiple of 8 (quantize.py:1088-1091)
  constexpr int SUB = (VEC >= 8) ? 8 : VEC;
#pragma unroll
  for (int s = 0; s < P::per; ++s) {
    const int64_t e0 = static_cast<int64_t>(s) * n + i0;
    if (e0 >= limit) continue;   // 3-bit padding rows (quantize.py:190-195 slices them off) / unpack limit = per*n
    OUT v[VEC];
#pragma unroll
    for (int c = 0; c < VEC; c += SUB) {
      if constexpr (DEQ && AXIS == 1) {
        const int64_t r = (e0 + c) / gs;
        const OUT z = zero[r];
        const OUT sc = scale[r];
#pragma unroll
        for (int j = 0; j < SUB; ++j) {
          const uint32_t q = (static_cast<uint32_t>(w[c + j]) >> P::shift(s)) & P::mask;
          v[c + j] = CD<OUT>::dequant(static_cast<float>(q), z, sc);
        }
      } else if constexpr (DEQ) {
#pragma unroll
        for (int j = 0; j < SUB; ++j) {
          const int64_t r = (e0 + c + j) % cols_u;
          const uint32_t q = (static_cast<uint32_t>(w[c + j]) >> P::shift(s)) & P::mask;
          v[c + j] = CD<OUT>::dequant(static_cast<float>(q), zero[r], scale[r]);
        }
      } else {
#pragma unroll
        for (int j = 0; j < SUB; ++j)
          v[c + j] = Conv<OUT>::of((static_cast<uint32_t>(w[c + j]) >> P::shift(s)) & P::mask);
      }
    }
    store_vec<OUT, VEC>(out + e0, v);
  }
}

// ---- host launchers ------------------------------------------------------------------------------
static inline dim3 grid_for(int64_t n, int vec) { return dim3(static_cast<unsigned>((n + 256LL * vec - 1) / (256LL * vec))); }

template <int NBITS, typename IN>
static int launch_pack_u8(const void* U, void* out, int64_t n, hipStream_t st) {
  if constexpr (std::is_same_v<IN, uint8_t>) {
    if (n % 16 == 0 && aligned16(U) && aligned16(out)) {
      constexpr int CH = Pk<NBITS>::per >= 4 ? 1 : (Pk<NBITS>::per == 2 ? 2 : 4);
      hipLaunchKernelGGL((pack_u8x16_kernel<NBITS, CH>), grid_for(n, 16 * CH), dim3(256), 0, st, static_cast<const uint8_t*>(U), static_cast<uint8_t*>(out), n);
      return check_launch("hqq_hip_pack");
    }
  }
  if (n % 16 == 0) hipLaunchKernelGGL((pack_u8_kernel<NBITS, IN, 16>), grid_for(n, 16), dim3(256), 0, st, static_cast<const IN*>(U), static_cast<uint8_t*>(out), n);
  else hipLaunchKernelGGL((pack_u8_kernel<NBITS, IN, 1>), grid_for(n, 1), dim3(256), 0, st, static_cast<const IN*>(U), static_cast<uint8_t*>(out), n);
  return check_launch("hqq_hip_pack");
}
template <typename IN>
static int launch_pack_3(const void* U, void* out, int64_t n, int64_t total, hipStream_t st) {
  if (n % 16 == 0 && aligned16(U) && aligned16(out)) hipLaunchKernelGGL((pack_3bit_kernel<IN, 16>), grid_for(n, 16), dim3(256), 0, st, static_cast<const IN*>(U), static_cast<int32_t*>(out), n, total);
  else if (n % 4 == 0) hipLaunchKernelGGL((pack_3bit_kernel<IN, 4>), grid_for(n, 4), dim3(256), 0, st, static_cast<const IN*>(U), static_cast<int32_t*>(out), n, total);
  else hipLaunchKernelGGL((pack_3bit_kernel<IN, 1>), grid_for(n, 1), dim3(256), 0, st, static_cast<const IN*>(U), static_cast<int32_t*>(out), n, total);
  return check_launch("hqq_hip_pack");
}

template <int NBITS, typename OUT, bool DEQ, int AXIS>
static int launch_unpack(const void* packed, const void* scale, const void* zero, void* out, int64_t n, int64_t limit,
                         int64_t gs, int64_t cols_u, hipStream_t st, const char* what) {
  constexpr int V = (NBITS == 3) ? 4 : 16;
  constexpr int SUB = (V >= 8) ? 8 : V;
  // the vector path needs every slab start (s*n) to stay V-aligned and, for AXIS 1 dequant, each
  // SUB-element sub-chunk to sit inside one group
  const bool vec_ok = (n % V == 0) && (!(DEQ && AXIS == 1) || gs % SUB == 0);
  if (vec_ok)
    hipLaunchKernelGGL((unpack_kernel<NBITS, OUT, DEQ, AXIS, V>), grid_for(n, V), dim3(256), 0, st, packed,
                       static_cast<const OUT*>(scale), static_cast<const OUT*>(zero), static_cast<OUT*>(out), n, limit, gs, cols_u);
  else
    hipLaunchKernelGGL((unpack_kernel<NBITS, OUT, DEQ, AXIS, 1>), grid_for(n, 1), dim3(256), 0, st, packed,
                       static_cast<const OUT*>(scale), static_cast<const OUT*>(zero), static_cast<OUT*>(out), n, limit, gs, cols_u);
  return check_launch(what);
}

template <typename OUT, bool DEQ, int AXIS>
static int dispatch_bits(int nbits, const void* packed, const void* scale, const void* zero, void* out, int64_t n,
                         int64_t limit, int64_t gs, int64_t cols_u, hipStream_t st, const char* what) {
  switch (nbits) {
    case 8: return launch_unpack<8, OUT, DEQ, AXIS>(packed, scale, zero, out, n, limit, gs, cols_u, st, what);
    case 4: return launch_unpack<4, OUT, DEQ, AXIS>(packed, scale, zero, out, n, limit, gs, cols_u, st, what);
    case 3: return launch_unpack<3, OUT, DEQ, AXIS>(packed, scale, zero, out, n, limit, gs, cols_u, st, what);
    case 2: return launch_unpack<2, OUT, DEQ, AXIS>(packed, scale, zero, out, n, limit, gs, cols_u, st, what);
    case 1: return launch_unpack<1, OUT, DEQ, AXIS>(packed, scale, zero, out, n, limit, gs, cols_u, st, what);
  }
  set_error("%s: nbits=%d not in {8,4,3,2,1}", what, nbits);
  return HQQ_ERR_NBITS;
}


// =================================================================================================
// 3-bit: the reference container <-> the stream layout of this build (w3s.h).  Patch-time work (HQQLinearHIP), once per layer.
//   reference: unpacked row r = n G + k / 64 (G = K / 64) sits in slab t = r / step of word (r % step, k % 64), bits [27 - 3 t, +3)
// =================================================================================================
// one thread per (packed row p, chunk c): 2 x 16 consecutive words of the reference container in, 12 bytes out
__global__ __launch_bounds__(256) void w3s_pack_kernel(const uint32_t* __restrict__ ref, uint32_t* __restrict__ out, int N, int K, int64_t step) {
  const int chunks = K / W3S_CHUNK_K, G = K / 64;
  const int64_t id = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (id >= static_cast<int64_t>(N / 2) * chunks) return;
  const int p = static_cast<int>(id / chunks), c = static_cast<int>(id % chunks);
  uint32_t D[3] = {0u, 0u, 0u};
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int64_t r = static_cast<int64_t>(p + s * (N / 2)) * G + (c >> 2);
    const int t = static_cast<int>(r / step);
    const uint32_t* src = ref + (r - t * step) * 64 + (c & 3) * 16;
#pragma unroll
    for (int i = 0; i < 16; ++i) w3s_put(D, s, i, (src[i] >> (27 - 3 * t)) & 7u);
  }
  uint32_t* dst = out + id * 3;
  dst[0] = D[0]; dst[1] = D[1]; dst[2] = D[2];
}
// one thread per word of the reference container: its ten levels gathered from the stream layout (rows past R: the zero padding)
__global__ __launch_bounds__(256) void w3s_unpack_kernel(const uint32_t* __restrict__ w3s, uint32_t* __restrict__ ref, int N, int K, int64_t step, int64_t R) {
  const int chunks = K / W3S_CHUNK_K, G = K / 64;
  const int64_t id = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (id >= step * 64) return;
  const int64_t i = id >> 6;
  const int col = static_cast<int>(id & 63);
  uint32_t word = 0u;
#pragma unroll
  for (int t = 0; t < 10; ++t) {
    const int64_t r = t * step + i;
    if (r < R) {
      const int n = static_cast<int>(r / G), k = static_cast<int>(r % G) * 64 + col;
      const int s = n >= N / 2 ? 1 : 0, p = n - s * (N / 2);
      const uint32_t* src = w3s + (static_cast<int64_t>(p) * chunks + (k >> 4)) * 3;
      const uint32_t D[3] = {src[0], src[1], src[2]};
      word |= w3s_get(D, s, k & 15) << (27 - 3 * t);
    }
  }
  ref[id] = word;
}

}  // namespace hqq

using namespace hqq;

extern "C" {

int64_t hqq_hip_packed_rows(int nbits, int64_t rows) {
  const int per = per_of(nbits);
  if (!per) return HQQ_ERR_NBITS;
  if (nbits == 3) return (rows + 9) / 10;
  if (rows % per) return HQQ_ERR_SHAPE;
  return rows / per;
}

int hqq_hip_pack(int nbits, const void* U, int in_dtype, int64_t rows, int64_t cols, void* out, void* stream) {
  clear_stale_error();
  const int64_t prow = hqq_hip_packed_rows(nbits, rows);
  if (prow < 0) { set_error("hqq_hip_pack: nbits=%d rows=%lld not packable", nbits, (long long)rows); return static_cast<int>(prow); }
  if (in_dtype != HQQ_U8 && in_dtype != HQQ_F32) { set_error("hqq_hip_pack: in_dtype must be U8 or F32"); return HQQ_ERR_DTYPE; }
  if (!aligned16(U) || !aligned16(out)) { set_error("hqq_hip_pack: pointers must be 16-byte aligned"); return HQQ_ERR_ALIGN; }
  const int64_t n = prow * cols, total = rows * cols;
  if (n == 0) return 0;
  hipStream_t st = as_stream(stream);
  const bool f = in_dtype == HQQ_F32;
  switch (nbits) {
    case 8: return f ? launch_pack_u8<8, float>(U, out, n, st) : launch_pack_u8<8, uint8_t>(U, out, n, st);
    case 4: return f ? launch_pack_u8<4, float>(U, out, n, st) : launch_pack_u8<4, uint8_t>(U, out, n, st);
    case 2: return f ? launch_pack_u8<2, float>(U, out, n, st) : launch_pack_u8<2, uint8_t>(U, out, n, st);
    case 1: return f ? launch_pack_u8<1, float>(U, out, n, st) : launch_pack_u8<1, uint8_t>(U, out, n, st);
    case 3: return f ? launch_pack_3<float>(U, out, n, total, st) : launch_pack_3<uint8_t>(U, out, n, total, st);
  }
  return HQQ_ERR_NBITS;
}

int hqq_hip_unpack(int nbits, const void* packed, int64_t packed_rows, int64_t cols, void* out, int out_dtype, void* stream) {
  clear_stale_error();
  const int per = per_of(nbits);
  if (!per) { set_error("hqq_hip_unpack: nbits=%d not in {8,4,3,2,1}", nbits); return HQQ_ERR_NBITS; }
  if (!aligned16(packed) || !aligned16(out)) { set_error("hqq_hip_unpack: pointers must be 16-byte aligned"); return HQQ_ERR_ALIGN; }
  const int64_t n = packed_rows * cols, limit = n * per;
  if (n == 0) return 0;
  hipStream_t st = as_stream(stream);
  switch (out_dtype) {
    case HQQ_U8: return dispatch_bits<uint8_t, false, 1>(nbits, packed, nullptr, nullptr, out, n, limit, 8, 1, st, "hqq_hip_unpack");
    case HQQ_F32: return dispatch_bits<float, false, 1>(nbits, packed, nullptr, nullptr, out, n, limit, 8, 1, st, "hqq_hip_unpack");
    case HQQ_F16: return dispatch_bits<half_t, false, 1>(nbits, packed, nullptr, nullptr, out, n, limit, 8, 1, st, "hqq_hip_unpack");
    case HQQ_BF16: return dispatch_bits<bf16_t, false, 1>(nbits, packed, nullptr, nullptr, out, n, limit, 8, 1, st, "hqq_hip_unpack");
  }
  set_error("hqq_hip_unpack: bad out_dtype %d", out_dtype);
  return HQQ_ERR_DTYPE;
}

int hqq_hip_dequantize(int nbits, const void* Wq, const void* scale, const void* zero, void* out,
                       int64_t N, int64_t K, int64_t group_size, int axis, int dtype, void* stream) {
  clear_stale_error();
  const int per = per_of(nbits);
  if (!per) { set_error("hqq_hip_dequantize: nbits=%d not in {8,4,3,2,1}", nbits); return HQQ_ERR_NBITS; }
  const int64_t total = N * K;
  if (group_size <= 0 || total % group_size || (axis != 0 && axis != 1)) {
    set_error("hqq_hip_dequantize: N*K=%lld not divisible by group_size=%lld, or bad axis %d", (long long)total, (long long)group_size, axis);
    return HQQ_ERR_SHAPE;
  }
  if (!aligned16(Wq) || !aligned16(out)) { set_error("hqq_hip_dequantize: pointers must be 16-byte aligned"); return HQQ_ERR_ALIGN; }
  if (total == 0) return 0;
  // unpacked matrix: axis 1 -> [R, gs], axis 0 -> [gs, R]; packing always slabs dim 0
  const int64_t R = total / group_size;
  const int64_t urows = (axis == 1) ? R : group_size, ucols = (axis == 1) ? group_size : R;
  const int64_t prow = hqq_hip_packed_rows(nbits, urows);
  if (prow < 0) { set_error("hqq_hip_dequantize: %lld unpacked rows not packable at %d bits", (long long)urows, nbits); return HQQ_ERR_SHAPE; }
  const int64_t n = prow * ucols;
  hipStream_t st = as_stream(stream);
  const char* what = "hqq_hip_dequantize";
#define HQQ_DQ(T)                                                                                               \
  return (axis == 1) ? dispatch_bits<T, true, 1>(nbits, Wq, scale, zero, out, n, total, group_size, R, st, what) \
                     : dispatch_bits<T, true, 0>(nbits, Wq, scale, zero, out, n, total, group_size, R, st, what)
  switch (dtype) {
    case HQQ_F32: HQQ_DQ(float);
    case HQQ_F16: HQQ_DQ(half_t);
    case HQQ_BF16: HQQ_DQ(bf16_t);
  }
#undef HQQ_DQ
  set_error("hqq_hip_dequantize: bad dtype %d", dtype);
  return HQQ_ERR_DTYPE;
}

static int w3s_check(const char* who, const void* a, const void* b, int64_t N, int64_t K) {
  if (N <= 0 || K <= 0 || N % 2 || K % 64) { set_error("%s: the 3-bit stream layout needs N %% 2 == 0 and K %% 64 == 0 (got %lld x %lld)", who, (long long)N, (long long)K); return HQQ_ERR_SHAPE; }
  if (N * (K / 64) > INT32_MAX || N * K / 2 > static_cast<int64_t>(INT32_MAX) * 8) { set_error("%s: size overflow", who); return HQQ_ERR_SHAPE; }
  if (!a || !b) { set_error("%s: null argument", who); return HQQ_ERR_SHAPE; }
  if (!aligned16(a) || !aligned16(b)) { set_error("%s: pointers must be 16-byte aligned", who); return HQQ_ERR_ALIGN; }
  return 0;
}

int hqq_hip_w3s_pack(const void* Wq_ref, void* w3s_out, int64_t N, int64_t K, void* stream) {
  clear_stale_error();
  if (const int rc = w3s_check("hqq_hip_w3s_pack", Wq_ref, w3s_out, N, K)) return rc;
  const int64_t R = N * (K / 64), step = (R + 9) / 10, n = (N / 2) * (K / W3S_CHUNK_K);
  hipLaunchKernelGGL(w3s_pack_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, as_stream(stream), static_cast<const uint32_t*>(Wq_ref),
                     static_cast<uint32_t*>(w3s_out), static_cast<int>(N), static_cast<int>(K), step);
  return check_launch("hqq_hip_w3s_pack");
}

int hqq_hip_w3s_unpack(const void* w3s, void* Wq_ref_out, int64_t N, int64_t K, void* stream) {
  clear_stale_error();
  if (const int rc = w3s_check("hqq_hip_w3s_unpack", w3s, Wq_ref_out, N, K)) return rc;
  const int64_t R = N * (K / 64), step = (R + 9) / 10, n = step * 64;
  hipLaunchKernelGGL(w3s_unpack_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, as_stream(stream), static_cast<const uint32_t*>(w3s),
                     static_cast<uint32_t*>(Wq_ref_out), static_cast<int>(N), static_cast<int>(K), step, R);
  return check_launch("hqq_hip_w3s_unpack");
}

}  // extern "C"
