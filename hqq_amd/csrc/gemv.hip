// gemv.hip — fused unpack -> dequantize -> GEMV for decode-shaped HQQLinear.forward (M <= 8), gfx950.
//
// Replaces, for axis=1 layers, the reference's per-call chain
//   BitPack.unpack_*  -> (W_r - zero) * scale -> torch.matmul(x, W.t()) (+ bias)
//   (hqq/core/bitpack.py:31-64, hqq/core/quantize.py:183-199, :880-898; patching.py:82-86 "TODO GEMV use-case")
// which moves ~12.5 B/param through HBM, by one pass over the packed weights (0.5625 B/param at 4-bit).
//
// HBM-bandwidth bound, no MFMA.  Data layout consumed as stored by the reference (no repacking):
//   Wq     [N/per, K] bytes; byte (p, k) holds W_q[p + s*N/per, k] for slab s at bit 8 - nbits*(s+1)
//   scale  [N*G] , zero [N*G] in the compute dtype, G = K/group_size; row n uses [n*G, (n+1)*G)
//
// Work decomposition: one wave streams one packed row (K bytes -> `per` output rows) with coalesced
// 16-byte-per-lane non-temporal loads (1 KiB per wave instruction); x is staged once per workgroup in
// LDS in the order the nibble extraction produces values; scale/zero are fetched per 16-element lane
// chunk (4 lanes share a 64-wide group, the loads coalesce in the TA).  Each lane keeps per*M fp32
// accumulators; one wave reduction per packed row at the end.  No inter-wave communication.
//
// Numerics: the weight is rebuilt exactly as Quantizer.dequantize does it — w = round(round(q - z) * s)
// in the compute dtype, two packed-fp16 instructions for two weights — then accumulated in fp32 with
// v_dot2_f32_f16.  The dequantised weights are therefore bit-identical to hqq_hip_dequantize / the
// reference; only the fp32 summation order differs from a BLAS.
#include "hqq_common.h"

namespace hqq {

constexpr int GEMV_WAVES = 8;              // waves per workgroup (512 threads)
constexpr int GEMV_KSTEP = 1024;           // k covered by one wave instruction: 64 lanes x 16 bytes
constexpr int GEMV_LDS_HALFS = 32768;      // x staging budget per workgroup: 64 KiB

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ half2_t as_h2(uint32_t u) { return __builtin_bit_cast(half2_t, u); }

// integer levels of slab S for the byte pairs (b0,b2) [odd=0] / (b1,b3) [odd=1] of one packed dword,
// returned as exact fp16 values.  (word & mask) | 0x6400 is the fp16 number 1024 + q*2^sh; one packed
// fma removes the bias exactly.
template <int NBITS, int S>
__device__ __forceinline__ half2_t levels(uint32_t word_or_shifted) {
  constexpr int per = 8 / NBITS;
  constexpr int sh = NBITS * (per - 1 - S);
  constexpr uint32_t m1 = ((NBITS == 8) ? 0xFFu : ((1u << NBITS) - 1u)) << sh;
  constexpr uint32_t m = m1 | (m1 << 16);
  const half2_t biased = as_h2((word_or_shifted & m) | 0x64006400u);
  constexpr float inv = 1.0f / static_cast<float>(1 << sh);
  const half2_t a = {static_cast<half_t>(inv), static_cast<half_t>(inv)};
  const half2_t b = {static_cast<half_t>(-1024.0f * inv), static_cast<half_t>(-1024.0f * inv)};
  return __builtin_elementwise_fma(biased, a, b);
}

// x staging order: lane chunk of 16 k-values is kept as two 16-byte planes (conflict-free ds_read_b128);
// inside a plane the 8 halfs are (k0,k2,k1,k3,k4,k6,k5,k7) so that half2 j pairs with levels<>(.., odd=j&1).
__device__ __forceinline__ u32x4 permute_x8(u32x4 v) {
  u32x4 r;
  r.x = (v.x & 0xFFFFu) | (v.y << 16);
  r.y = (v.x >> 16) | (v.y & 0xFFFF0000u);
  r.z = (v.z & 0xFFFFu) | (v.w << 16);
  r.w = (v.z >> 16) | (v.w & 0xFFFF0000u);
  return r;
}

template <int NBITS, int M, int S, int PER>
struct SlabLoop {
  static __device__ __forceinline__ void run(const u32x4& w, const half2_t (&zz)[PER], const half2_t (&ss)[PER],
                                             const half2_t (&xr)[M][8], float (&acc)[M][PER]) {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const uint32_t word = w[d];
      const half2_t q0 = levels<NBITS, S>(word);        // bytes (4d+0, 4d+2)
      const half2_t q1 = levels<NBITS, S>(word >> 8);   // bytes (4d+1, 4d+3)
      const half2_t w0 = (q0 - zz[S]) * ss[S];          // two roundings, as Quantizer.dequantize
      const half2_t w1 = (q1 - zz[S]) * ss[S];
#pragma unroll
      for (int m = 0; m < M; ++m) {
        acc[m][S] = __builtin_amdgcn_fdot2(w0, xr[m][2 * d], acc[m][S], false);
        acc[m][S] = __builtin_amdgcn_fdot2(w1, xr[m][2 * d + 1], acc[m][S], false);
      }
    }
    if constexpr (S + 1 < PER) SlabLoop<NBITS, M, S + 1, PER>::run(w, zz, ss, xr, acc);
  }
};

template <int NBITS, int M>
__global__ __launch_bounds__(GEMV_WAVES * 64) void gemv_f16_kernel(
    const half_t* __restrict__ x, const uint8_t* __restrict__ Wq, const half_t* __restrict__ scale,
    const half_t* __restrict__ zero, const half_t* __restrict__ bias, half_t* __restrict__ y,
    int N, int K, int gs, int n_prow, int kc /* k staged per pass, multiple of GEMV_KSTEP */) {
  constexpr int PER = 8 / NBITS;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  u32x4* xs = reinterpret_cast<u32x4*>(smem);   // [M][kc/1024][2 planes][64 lanes] x 16 B

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int G = K / gs;
  const int rows_per_slab = N / PER;
  const int tiles = (n_prow + GEMV_WAVES - 1) / GEMV_WAVES;
  const int nchunk = (K + kc - 1) / kc;
  const int planes_per_m = (kc / GEMV_KSTEP) * 2 * 64;

  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int pr = tile * GEMV_WAVES + wave;
    const bool row_ok = pr < n_prow;
    float acc[M][PER];
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
      for (int s = 0; s < PER; ++s) acc[m][s] = 0.f;

    for (int ch = 0; ch < nchunk; ++ch) {
      const int kb = ch * kc;
      // ---- stage x[:, kb : kb+kc] into LDS (once per workgroup when K fits one chunk) ----
      if (nchunk > 1 || tile == static_cast<int>(blockIdx.x)) {
        if (nchunk > 1) __syncthreads();   // previous chunk fully consumed
        const int vec_per_m = kc / 8;
        for (int v = tid; v < M * vec_per_m; v += GEMV_WAVES * 64) {
          const int m = v / vec_per_m, j = v - m * vec_per_m;
          const int k = kb + j * 8;
          u32x4 val = {0u, 0u, 0u, 0u};
          if (k < K) val = *reinterpret_cast<const u32x4*>(x + static_cast<int64_t>(m) * K + k);
          const int it = j >> 7, rem = j & 127, ln = rem >> 1, h = rem & 1;
          xs[m * planes_per_m + (it * 2 + h) * 64 + ln] = permute_x8(val);
        }
        __syncthreads();
      }
      if (!row_ok) continue;
      const uint8_t* wrow = Wq + static_cast<int64_t>(pr) * K;
      const int nit = (min(K - kb, kc) + GEMV_KSTEP - 1) / GEMV_KSTEP;
      constexpr int U = 4;
      for (int it0 = 0; it0 < nit; it0 += U) {
        u32x4 w[U];
        half_t zr[U][PER], sc[U][PER];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int k0 = kb + (it0 + u) * GEMV_KSTEP + lane * 16;
          ok[u] = (it0 + u < nit) && (k0 < K);
          if (ok[u]) {
            w[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow + k0));
            const int g = k0 / gs;
#pragma unroll
            for (int s = 0; s < PER; ++s) {
              const int64_t r = static_cast<int64_t>(pr + s * rows_per_slab) * G + g;
              zr[u][s] = zero[r];
              sc[u][s] = scale[r];
            }
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (!ok[u]) continue;
          half2_t xr[M][8];
#pragma unroll
          for (int m = 0; m < M; ++m) {
            const u32x4 a = xs[m * planes_per_m + ((it0 + u) * 2 + 0) * 64 + lane];
            const u32x4 b = xs[m * planes_per_m + ((it0 + u) * 2 + 1) * 64 + lane];
            xr[m][0] = as_h2(a.x); xr[m][1] = as_h2(a.y); xr[m][2] = as_h2(a.z); xr[m][3] = as_h2(a.w);
            xr[m][4] = as_h2(b.x); xr[m][5] = as_h2(b.y); xr[m][6] = as_h2(b.z); xr[m][7] = as_h2(b.w);
          }
          half2_t zz[PER], ss[PER];
#pragma unroll
          for (int s = 0; s < PER; ++s) { zz[s] = half2_t{zr[u][s], zr[u][s]}; ss[s] = half2_t{sc[u][s], sc[u][s]}; }
          SlabLoop<NBITS, M, 0, PER>::run(w[u], zz, ss, xr, acc);
        }
      }
    }
    // ---- one wave reduction per output row; lane 0 writes ----
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
      for (int s = 0; s < PER; ++s) {
        const float v = wave_sum(acc[m][s]);
        if (lane == 0 && row_ok) {
          const int n = pr + s * rows_per_slab;
          half_t o = static_cast<half_t>(v);
          if (bias) o = o + bias[n];   // `out += bias` on the rounded matmul result (quantize.py:896-897)
          y[static_cast<int64_t>(m) * N + n] = o;
        }
      }
  }
}

template <int NBITS, int M>
static int launch_gemv_f16(const void* x, const void* Wq, const void* scale, const void* zero, const void* bias, void* y,
                           int N, int K, int gs, hipStream_t st) {
  constexpr int PER = 8 / NBITS;
  const int n_prow = N / PER;
  const int kpad = (K + GEMV_KSTEP - 1) / GEMV_KSTEP * GEMV_KSTEP;
  const int kc_max = GEMV_LDS_HALFS / M / GEMV_KSTEP * GEMV_KSTEP;
  const int kc = kpad < kc_max ? kpad : kc_max;
  const size_t lds = static_cast<size_t>(M) * kc * 2;
  const int tiles = (n_prow + GEMV_WAVES - 1) / GEMV_WAVES;
  const int grid = tiles < 512 ? tiles : 512;
  hipLaunchKernelGGL((gemv_f16_kernel<NBITS, M>), dim3(grid), dim3(GEMV_WAVES * 64), lds, st,
                     static_cast<const half_t*>(x), static_cast<const uint8_t*>(Wq), static_cast<const half_t*>(scale),
                     static_cast<const half_t*>(zero), static_cast<const half_t*>(bias), static_cast<half_t*>(y),
                     N, K, gs, n_prow, kc);
  return check_launch("hqq_hip_gemv");
}

template <int NBITS>
static int dispatch_m(int M, const void* x, const void* Wq, const void* scale, const void* zero, const void* bias, void* y,
                      int N, int K, int gs, hipStream_t st) {
  switch (M) {
    case 1: return launch_gemv_f16<NBITS, 1>(x, Wq, scale, zero, bias, y, N, K, gs, st);
    case 2: return launch_gemv_f16<NBITS, 2>(x, Wq, scale, zero, bias, y, N, K, gs, st);
    case 3: return launch_gemv_f16<NBITS, 3>(x, Wq, scale, zero, bias, y, N, K, gs, st);
    case 4: return launch_gemv_f16<NBITS, 4>(x, Wq, scale, zero, bias, y, N, K, gs, st);
    case 5: return launch_gemv_f16<NBITS, 5>(x, Wq, scale, zero, bias, y, N, K, gs, st);
    case 6: return launch_gemv_f16<NBITS, 6>(x, Wq, scale, zero, bias, y, N, K, gs, st);
    case 7: return launch_gemv_f16<NBITS, 7>(x, Wq, scale, zero, bias, y, N, K, gs, st);
    case 8: return launch_gemv_f16<NBITS, 8>(x, Wq, scale, zero, bias, y, N, K, gs, st);
  }
  return HQQ_ERR_SHAPE;
}

}  // namespace hqq

using namespace hqq;

extern "C" int hqq_hip_gemv(int nbits, const void* x, const void* Wq, const void* scale, const void* zero, const void* bias,
                            void* y, int64_t M, int64_t N, int64_t K, int64_t group_size, int dtype, void* stream) {
  if (M < 1 || M > HQQ_GEMV_MAX_M) { set_error("hqq_hip_gemv: M=%lld outside [1,%d]", (long long)M, HQQ_GEMV_MAX_M); return HQQ_ERR_SHAPE; }
  if (N <= 0 || K <= 0 || group_size <= 0 || K % group_size) { set_error("hqq_hip_gemv: bad N/K/group_size"); return HQQ_ERR_SHAPE; }
  if (N > INT32_MAX || K > INT32_MAX || N * (K / group_size) > INT32_MAX) { set_error("hqq_hip_gemv: size overflow"); return HQQ_ERR_SHAPE; }
  if (!aligned16(x) || !aligned16(Wq) || !aligned16(y)) { set_error("hqq_hip_gemv: pointers must be 16-byte aligned"); return HQQ_ERR_ALIGN; }
  if (nbits != 4 && nbits != 2 && nbits != 8 && nbits != 1) { set_error("hqq_hip_gemv: nbits=%d not covered by the fused GEMV", nbits); return HQQ_ERR_UNSUPPORTED; }
  const int per = 8 / nbits;
  if (N % per || group_size % 16 || K % 16) {
    set_error("hqq_hip_gemv: needs N %% %d == 0, group_size %% 16 == 0 (got N=%lld gs=%lld)", per, (long long)N, (long long)group_size);
    return HQQ_ERR_UNSUPPORTED;
  }
  if (dtype != HQQ_F16) { set_error("hqq_hip_gemv: dtype %d not covered (fp16 only for now)", dtype); return HQQ_ERR_UNSUPPORTED; }
  hipStream_t st = as_stream(stream);
  const int n = static_cast<int>(N), k = static_cast<int>(K), gs = static_cast<int>(group_size), m = static_cast<int>(M);
  switch (nbits) {
    case 8: return dispatch_m<8>(m, x, Wq, scale, zero, bias, y, n, k, gs, st);
    case 4: return dispatch_m<4>(m, x, Wq, scale, zero, bias, y, n, k, gs, st);
    case 2: return dispatch_m<2>(m, x, Wq, scale, zero, bias, y, n, k, gs, st);
    case 1: return dispatch_m<1>(m, x, Wq, scale, zero, bias, y, n, k, gs, st);
  }
  return HQQ_ERR_NBITS;
}
