// gemm.hip — fused unpack -> dequantize -> MFMA GEMM for prefill-shaped HQQLinear.forward (large M), gfx950.
//
// Reference path replaced: Quantizer.dequantize materialising the fp16 [N,K] weight in HBM followed by
// torch.matmul(x, W.t()) (hqq/core/quantize.py:183-199, :880-898).  Here the packed weights are the
// only weight bytes that leave HBM: each workgroup dequantises its [BN x 64] weight tile in registers
// (bit-identical to hqq_hip_dequantize: two fp16 roundings), writes it to LDS in MFMA operand order and
// contracts it with the activation tile on the matrix cores (v_mfma_f32_16x16x32_f16, fp32 accumulate).
//
// Tile:   BM = 128 or 256 tokens (template; 256 halves the dequantisation VALU work per flop and is used for large M),
//         BN = 128 output columns (PER slabs x 128/PER packed rows, because one packed byte holds rows
//         p, p+N/PER, ...), BM = 128 tokens, BK = 64.  4 waves as 2(n) x 2(m), 64x64 per wave =
//         4x4 MFMA tiles.  W is the MFMA "A" operand (rows = output features) so every lane ends up with
//         4 consecutive output features of one token: 8-byte stores.
// LDS:    [128 rows][64 k] fp16 for W and for x, 16-byte chunk index XOR (row & 7): conflict-free
//         ds_read_b128 fragment reads and ds_write_b128 fills.
// k order inside each 4-k quad is (k0,k2,k1,k3) for BOTH operands — the order nibble extraction
//         produces two-at-a-time — which leaves the dot product unchanged and saves the re-interleave.
// Pipeline v1: register prefetch of the next K-step's global loads while the current one is in the
//         MFMA phase; one LDS buffer, two barriers per K-step.
#include <stdlib.h>

#include "hqq_common.h"

namespace hqq {

constexpr int GB_N = 128, GB_K = 64, G_THREADS = 256;   // token-tile height BM (128 or 256) is a template parameter

typedef _Float16 h8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ half2_t g_as_h2(uint32_t u) { return __builtin_bit_cast(half2_t, u); }
__device__ __forceinline__ uint32_t g_as_u32(half2_t h) { return __builtin_bit_cast(uint32_t, h); }

template <int NBITS, int S>
__device__ __forceinline__ half2_t g_levels(uint32_t word_or_shifted) {
  constexpr int per = 8 / NBITS;
  constexpr int sh = NBITS * (per - 1 - S);
  constexpr uint32_t m1 = ((NBITS == 8) ? 0xFFu : ((1u << NBITS) - 1u)) << sh;
  constexpr uint32_t m = m1 | (m1 << 16);
  const half2_t biased = g_as_h2((word_or_shifted & m) | 0x64006400u);
  constexpr float inv = 1.0f / static_cast<float>(1 << sh);
  const half2_t a = {static_cast<half_t>(inv), static_cast<half_t>(inv)};
  const half2_t b = {static_cast<half_t>(-1024.0f * inv), static_cast<half_t>(-1024.0f * inv)};
  return __builtin_elementwise_fma(biased, a, b);
}

__device__ __forceinline__ u32x4 g_permute_x8(u32x4 v) {   // (k0..k7) -> (k0,k2,k1,k3,k4,k6,k5,k7)
  u32x4 r;
  r.x = (v.x & 0xFFFFu) | (v.y << 16);
  r.y = (v.x >> 16) | (v.y & 0xFFFF0000u);
  r.z = (v.z & 0xFFFFu) | (v.w << 16);
  r.w = (v.z >> 16) | (v.w & 0xFFFF0000u);
  return r;
}

// byte offset of 16-byte chunk `c` (0..7) of row `row` in a [rows][64] fp16 LDS tile
__device__ __forceinline__ int lds_off(int row, int c) { return row * 128 + ((c ^ (row & 7)) << 4); }

template <int NBITS, int S, int PER>
struct DeqSlab {
  // dequantise the 16 k-values of slab S held in `w` and write them (2 chunks) to the W tile
  static __device__ __forceinline__ void run(const u32x4& w, const half_t (&z)[PER], const half_t (&s)[PER], uint8_t* ldsW,
                                             int prow_in_tile, int kchunk16) {
    const half2_t zz = {z[S], z[S]}, ss = {s[S], s[S]};
    uint32_t o[8];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const half2_t q0 = g_levels<NBITS, S>(w[d]);
      const half2_t q1 = g_levels<NBITS, S>(w[d] >> 8);
      o[2 * d] = g_as_u32((q0 - zz) * ss);        // (k0,k2) of quad d
      o[2 * d + 1] = g_as_u32((q1 - zz) * ss);    // (k1,k3)
    }
    const int row = S * (GB_N / PER) + prow_in_tile;
    *reinterpret_cast<u32x4*>(ldsW + lds_off(row, kchunk16 * 2)) = u32x4{o[0], o[1], o[2], o[3]};
    *reinterpret_cast<u32x4*>(ldsW + lds_off(row, kchunk16 * 2 + 1)) = u32x4{o[4], o[5], o[6], o[7]};
    if constexpr (S + 1 < PER) DeqSlab<NBITS, S + 1, PER>::run(w, z, s, ldsW, prow_in_tile, kchunk16);
  }
};

template <int NBITS, int BM>
__global__ __launch_bounds__(G_THREADS, BM == 256 ? 2 : 1) void gemm_f16_kernel(
    const half_t* __restrict__ x, const uint8_t* __restrict__ Wq, const half_t* __restrict__ scale,
    const half_t* __restrict__ zero, const half_t* __restrict__ bias, half_t* __restrict__ y,
    int M, int N, int K, int gs, int n_tiles) {
  constexpr int PER = 8 / NBITS;
  constexpr int PROWS = GB_N / PER;                  // packed rows per tile
  constexpr int WLOADS = (PROWS * GB_K) / (16 * G_THREADS) > 0 ? (PROWS * GB_K) / (16 * G_THREADS) : 1;
  constexpr int WTHREADS = (PROWS * GB_K) / 16 / WLOADS;   // threads that carry a packed chunk
  constexpr int MT = BM / 32;       // 16-token MFMA tiles per wave along M (each wave covers BM/2 tokens)
  constexpr int XROWS = BM / 128;   // x rows staged per thread
  __shared__ __attribute__((aligned(16))) uint8_t lds[(GB_N + BM) * GB_K * 2];
  uint8_t* ldsW = lds;
  uint8_t* ldsX = lds + GB_N * GB_K * 2;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wm = wave & 1;
  // n tiles vary fastest so that consecutive workgroups share the x tile (L2) and sweep the weights once per m tile
  const int nt = blockIdx.x % n_tiles, mt = blockIdx.x / n_tiles;
  const int rows_per_slab = N / PER;
  const int p0 = nt * PROWS;                         // first packed row of the tile
  const int m0 = mt * BM;
  const int G = K / gs;

  // ---- per-thread global->register staging assignment ----
  const int wp = tid / 4, wk = tid & 3;              // packed row in tile / 16-k chunk (PROWS*4 threads active)
  const bool w_active = tid < WTHREADS && (p0 + wp) < rows_per_slab;
  const int xr_ = tid >> 1, xh = tid & 1;            // x row in tile (+128 per extra row), half (32 k = 4 chunks)

  u32x4 wreg = {0u, 0u, 0u, 0u};
  half_t zreg[PER], sreg[PER];
  u32x4 xreg[XROWS][4];

  auto load_regs = [&](int kt) {
    const int k0 = kt * GB_K;
    if (w_active) {
      wreg = *reinterpret_cast<const u32x4*>(Wq + static_cast<int64_t>(p0 + wp) * K + k0 + wk * 16);
      const int g = (k0 + wk * 16) / gs;
#pragma unroll
      for (int s = 0; s < PER; ++s) {
        const int64_t r = static_cast<int64_t>(p0 + wp + s * rows_per_slab) * G + g;
        zreg[s] = zero[r];
        sreg[s] = scale[r];
      }
    }
#pragma unroll
    for (int xr = 0; xr < XROWS; ++xr) {
      const int row = m0 + xr * 128 + xr_;
      if (row < M) {
        const half_t* src = x + static_cast<int64_t>(row) * K + k0 + xh * 32;
#pragma unroll
        for (int c = 0; c < 4; ++c) xreg[xr][c] = *reinterpret_cast<const u32x4*>(src + c * 8);
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) xreg[xr][c] = u32x4{0u, 0u, 0u, 0u};
      }
    }
  };

  auto write_lds = [&]() {
    if (tid < WTHREADS) {
      if (w_active) {
        DeqSlab<NBITS, 0, PER>::run(wreg, zreg, sreg, ldsW, wp, wk);
      } else {   // rows past the end of the slab: zero weights
#pragma unroll
        for (int s = 0; s < PER; ++s) {
          *reinterpret_cast<u32x4*>(ldsW + lds_off(s * PROWS + wp, wk * 2)) = u32x4{0u, 0u, 0u, 0u};
          *reinterpret_cast<u32x4*>(ldsW + lds_off(s * PROWS + wp, wk * 2 + 1)) = u32x4{0u, 0u, 0u, 0u};
        }
      }
    }
#pragma unroll
    for (int xr = 0; xr < XROWS; ++xr)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        *reinterpret_cast<u32x4*>(ldsX + lds_off(xr * 128 + xr_, xh * 4 + c)) = g_permute_x8(xreg[xr][c]);
  };

  f32x4 acc[4][MT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = K / GB_K;
  load_regs(0);
  const int fr = lane & 15, fq = lane >> 4;          // fragment row / k-octet
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();                                 // previous MFMA phase done with the LDS tiles
    write_lds();
    __syncthreads();
    if (kt + 1 < nk) load_regs(kt + 1);              // global loads in flight during the MFMA phase
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      h8_t a[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        a[i] = *reinterpret_cast<const h8_t*>(ldsW + lds_off(wn * 64 + i * 16 + fr, ks * 4 + fq));
#pragma unroll
      for (int jh = 0; jh < MT; jh += 4) {   // four token tiles at a time keeps the B fragments at 16 VGPRs
        h8_t b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          b[j] = *reinterpret_cast<const h8_t*>(ldsX + lds_off(wm * (BM / 2) + (jh + j) * 16 + fr, ks * 4 + fq));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][jh + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][jh + j], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: C tile (i,j): lane holds features (fq*4 .. +3) of tile-row block i, token fr of block j ----
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int trow = wn * 64 + i * 16 + fq * 4;      // tile row (feature) of acc[.][.][0]
    const int slab = trow / PROWS, pin = trow % PROWS;
    const int prow = p0 + pin;
    if (prow >= rows_per_slab) continue;             // whole quad out of range
    const int n = slab * rows_per_slab + prow;
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const int m = m0 + wm * (BM / 2) + j * 16 + fr;
      if (m >= M) continue;
      half_t o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        o[r] = static_cast<half_t>(acc[i][j][r]);
        if (bias && prow + r < rows_per_slab) o[r] = o[r] + bias[n + r];
      }
      half_t* dst = y + static_cast<int64_t>(m) * N + n;
      if (prow + 3 < rows_per_slab) {
        *reinterpret_cast<u32x2*>(dst) = *reinterpret_cast<u32x2*>(o);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (prow + r < rows_per_slab) dst[r] = o[r];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Wave-specialised variant for large M: 256 tokens x 128 features per workgroup, 8 waves.  Waves 0-3 are *producers* (packed
// weights -> registers -> exact dequantisation -> LDS; x tile -> LDS), waves 4-7 are *consumers* (LDS -> MFMA only, 64 MFMAs per
// K-step each).  A workgroup's waves are dealt over the SIMDs cyclically, so every SIMD hosts one producer and one consumer:
// the unpack arithmetic (VALU) and the contraction (matrix pipe) run concurrently instead of alternating inside one wave —
// the single-role kernel above measures MFMA busy 36 % / VALU busy 40 % with almost no overlap (profiles/r01_prefill_*).
// Two LDS stages (2 x 48 KiB), one workgroup barrier per K-step: producers fill stage (kt+1)&1 while consumers read kt&1.
// ------------------------------------------------------------------------------------------------------------------
constexpr int WS_BM = 256, WS_THREADS = 512;

template <int NBITS>
__global__ __launch_bounds__(WS_THREADS, 2) void gemm_ws_f16_kernel(
    const half_t* __restrict__ x, const uint8_t* __restrict__ Wq, const half_t* __restrict__ scale,
    const half_t* __restrict__ zero, const half_t* __restrict__ bias, half_t* __restrict__ y,
    int M, int N, int K, int gs, int n_tiles) {
  constexpr int PER = 8 / NBITS;
  constexpr int PROWS = GB_N / PER;                       // packed rows per tile
  constexpr int WTHREADS = (PROWS * GB_K) / 16;           // producer threads that carry a packed 16-byte chunk (<= 256)
  constexpr int STAGE = (GB_N + WS_BM) * GB_K * 2;        // W tile + x tile
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];   // two stages

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave < 4;
  // XCD-aware tile order: workgroup b runs on XCD b % 8 (observed dispatch order; speed only, never correctness).  All n tiles
  // of one token tile are given to the same XCD, so the 256 x K activation tile is fetched into one L2 and shared by the
  // 32 CUs working on it, instead of being pulled into all eight L2s.
  int nt, mt;
  {
    const int m_tiles = gridDim.x / n_tiles;
    if ((m_tiles & 7) == 0) {
      const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
      nt = local % n_tiles;
      mt = (local / n_tiles) * 8 + xcd;
    } else {
      nt = blockIdx.x % n_tiles;
      mt = blockIdx.x / n_tiles;
    }
  }
  const int rows_per_slab = N / PER;
  const int p0 = nt * PROWS;
  const int m0 = mt * WS_BM;
  const int G = K / gs;
  const int nk = K / GB_K;

  // ---- producer state ----
  const int ptid = tid & 255;
  const int wp = ptid / 4, wk = ptid & 3;
  const bool w_active = ptid < WTHREADS && (p0 + wp) < rows_per_slab;
  const int xr_ = ptid >> 1, xh = ptid & 1;
  // three K-steps of global data in flight per producer thread (a K-step of MFMA work is ~1000 cycles, an HBM round trip 2-4x that)
  struct PStage { u32x4 wreg; half_t zreg[PER], sreg[PER]; u32x4 xreg[2][4]; };
  PStage ring[3];

  auto load_regs = [&](PStage& st, int kt) {
    const int k0 = kt * GB_K;
    st.wreg = u32x4{0u, 0u, 0u, 0u};
    if (w_active) {
      st.wreg = *reinterpret_cast<const u32x4*>(Wq + static_cast<int64_t>(p0 + wp) * K + k0 + wk * 16);
      const int g = (k0 + wk * 16) / gs;
#pragma unroll
      for (int s = 0; s < PER; ++s) {
        const int64_t r = static_cast<int64_t>(p0 + wp + s * rows_per_slab) * G + g;
        st.zreg[s] = zero[r];
        st.sreg[s] = scale[r];
      }
    }
#pragma unroll
    for (int xr = 0; xr < 2; ++xr) {
      const int row = m0 + xr * 128 + xr_;
      if (row < M) {
        const half_t* src = x + static_cast<int64_t>(row) * K + k0 + xh * 32;
#pragma unroll
        for (int c = 0; c < 4; ++c) st.xreg[xr][c] = *reinterpret_cast<const u32x4*>(src + c * 8);
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) st.xreg[xr][c] = u32x4{0u, 0u, 0u, 0u};
      }
    }
  };
  auto write_lds = [&](const PStage& st, uint8_t* ldsW) {
    uint8_t* ldsX = ldsW + GB_N * GB_K * 2;
    if (ptid < WTHREADS) {
      if (w_active) {
        DeqSlab<NBITS, 0, PER>::run(st.wreg, st.zreg, st.sreg, ldsW, wp, wk);
      } else {
#pragma unroll
        for (int s = 0; s < PER; ++s) {
          *reinterpret_cast<u32x4*>(ldsW + lds_off(s * PROWS + wp, wk * 2)) = u32x4{0u, 0u, 0u, 0u};
          *reinterpret_cast<u32x4*>(ldsW + lds_off(s * PROWS + wp, wk * 2 + 1)) = u32x4{0u, 0u, 0u, 0u};
        }
      }
    }
#pragma unroll
    for (int xr = 0; xr < 2; ++xr)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        *reinterpret_cast<u32x4*>(ldsX + lds_off(xr * 128 + xr_, xh * 4 + c)) = g_permute_x8(st.xreg[xr][c]);
  };

  // The two roles run separate loops (separate register allocation: the consumer's 128 accumulator registers are not live in
  // the producer's code) and meet at the same number of workgroup barriers: 1 + nk.
  if (producer) {
    load_regs(ring[0], 0);
    if (nk > 1) load_regs(ring[1], 1);
    if (nk > 2) load_regs(ring[2], 2);
    write_lds(ring[0], lds);
    if (nk > 3) load_regs(ring[0], 3);
    __syncthreads();
    // iteration kt: fill stage (kt+1)&1 from ring slot (kt+1)%3, then refill that slot with step kt+4
    auto step = [&](PStage& st, int kt) {
      if (kt + 1 < nk) write_lds(st, lds + ((kt + 1) & 1) * STAGE);
      if (kt + 4 < nk) load_regs(st, kt + 4);
      __syncthreads();                                             // stage (kt+1)&1 filled, stage kt&1 drained
    };
    int kt = 0;
    for (; kt + 2 < nk; kt += 3) {
      step(ring[1], kt);
      step(ring[2], kt + 1);
      step(ring[0], kt + 2);
    }
    if (kt < nk) { step(ring[1], kt); ++kt; }
    if (kt < nk) { step(ring[2], kt); ++kt; }
    return;
  }

  // ---- consumer: wave (wn, wm) owns features wn*64..+63 x tokens wm*128..+127 ----
  const int cw = wave & 3;
  const int wn = cw & 1, wm = cw >> 1;
  const int fr = lane & 15, fq = lane >> 4;
  f32x4 acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const uint8_t* ldsW = lds + (kt & 1) * STAGE;
    const uint8_t* ldsX = ldsW + GB_N * GB_K * 2;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      h8_t a[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        a[i] = *reinterpret_cast<const h8_t*>(ldsW + lds_off(wn * 64 + i * 16 + fr, ks * 4 + fq));
#pragma unroll
      for (int jh = 0; jh < 8; jh += 4) {
        h8_t b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          b[j] = *reinterpret_cast<const h8_t*>(ldsX + lds_off(wm * 128 + (jh + j) * 16 + fr, ks * 4 + fq));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][jh + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][jh + j], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // ---- epilogue (consumers): lane holds features (fq*4 .. +3) of feature block i, token fr of token block j ----
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int trow = wn * 64 + i * 16 + fq * 4;
    const int slab = trow / PROWS, pin = trow % PROWS;
    const int prow = p0 + pin;
    if (prow >= rows_per_slab) continue;
    const int n = slab * rows_per_slab + prow;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int m = m0 + wm * 128 + j * 16 + fr;
      if (m >= M) continue;
      half_t o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        o[r] = static_cast<half_t>(acc[i][j][r]);
        if (bias && prow + r < rows_per_slab) o[r] = o[r] + bias[n + r];
      }
      half_t* dst = y + static_cast<int64_t>(m) * N + n;
      if (prow + 3 < rows_per_slab) {
        *reinterpret_cast<u32x2*>(dst) = *reinterpret_cast<u32x2*>(o);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (prow + r < rows_per_slab) dst[r] = o[r];
      }
    }
  }
}

template <int NBITS>
static int launch_gemm_ws_f16(const void* x, const void* Wq, const void* scale, const void* zero, const void* bias, void* y,
                              int M, int N, int K, int gs, hipStream_t st) {
  constexpr int PER = 8 / NBITS;
  const int rows_per_slab = N / PER;
  const int n_tiles = (rows_per_slab + GB_N / PER - 1) / (GB_N / PER);
  const int m_tiles = (M + WS_BM - 1) / WS_BM;
  const int64_t blocks = static_cast<int64_t>(n_tiles) * m_tiles;
  if (blocks > INT32_MAX) { set_error("hqq_hip_gemm: grid too large"); return HQQ_ERR_SHAPE; }
  constexpr int lds_bytes = 2 * (GB_N + WS_BM) * GB_K * 2;
  auto kern = gemm_ws_f16_kernel<NBITS>;
  static bool raised = false;   // per instantiation
  if (!raised) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) { set_error("hqq_hip_gemm: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e)); return static_cast<int>(e); }
    raised = true;
  }
  hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(blocks)), dim3(WS_THREADS), lds_bytes, st,
                     static_cast<const half_t*>(x), static_cast<const uint8_t*>(Wq), static_cast<const half_t*>(scale),
                     static_cast<const half_t*>(zero), static_cast<const half_t*>(bias), static_cast<half_t*>(y),
                     M, N, K, gs, n_tiles);
  return check_launch("hqq_hip_gemm");
}

template <int NBITS, int BM>
static int launch_gemm_f16(const void* x, const void* Wq, const void* scale, const void* zero, const void* bias, void* y,
                           int M, int N, int K, int gs, hipStream_t st) {
  constexpr int PER = 8 / NBITS;
  const int rows_per_slab = N / PER;
  const int n_tiles = (rows_per_slab + GB_N / PER - 1) / (GB_N / PER);
  const int m_tiles = (M + BM - 1) / BM;
  const int64_t blocks = static_cast<int64_t>(n_tiles) * m_tiles;
  if (blocks > INT32_MAX) { set_error("hqq_hip_gemm: grid too large"); return HQQ_ERR_SHAPE; }
  hipLaunchKernelGGL((gemm_f16_kernel<NBITS, BM>), dim3(static_cast<unsigned>(blocks)), dim3(G_THREADS), 0, st,
                     static_cast<const half_t*>(x), static_cast<const uint8_t*>(Wq), static_cast<const half_t*>(scale),
                     static_cast<const half_t*>(zero), static_cast<const half_t*>(bias), static_cast<half_t*>(y),
                     M, N, K, gs, n_tiles);
  return check_launch("hqq_hip_gemm");
}

}  // namespace hqq

using namespace hqq;

extern "C" {

int hqq_hip_gemm(int nbits, const void* x, const void* Wq, const void* scale, const void* zero, const void* bias,
                 void* y, int64_t M, int64_t N, int64_t K, int64_t group_size, int dtype, void* stream) {
  clear_stale_error();
  if (M < 1 || N <= 0 || K <= 0 || group_size <= 0 || K % group_size) { set_error("hqq_hip_gemm: bad M/N/K/group_size"); return HQQ_ERR_SHAPE; }
  if (M > INT32_MAX || N > INT32_MAX || K > INT32_MAX || N * (K / group_size) > INT32_MAX) { set_error("hqq_hip_gemm: size overflow"); return HQQ_ERR_SHAPE; }
  if (!aligned16(x) || !aligned16(Wq) || !aligned16(y)) { set_error("hqq_hip_gemm: pointers must be 16-byte aligned"); return HQQ_ERR_ALIGN; }
  if (nbits != 4 && nbits != 2) { set_error("hqq_hip_gemm: nbits=%d not covered by the fused GEMM", nbits); return HQQ_ERR_UNSUPPORTED; }
  const int per = 8 / nbits;
  if (N % per || (N / per) % 4 || group_size % 16 || K % GB_K) {
    set_error("hqq_hip_gemm: needs N %% %d == 0, K %% 64 == 0, group_size %% 16 == 0 (got N=%lld K=%lld gs=%lld)", 4 * per, (long long)N, (long long)K, (long long)group_size);
    return HQQ_ERR_UNSUPPORTED;
  }
  if (dtype != HQQ_F16) { set_error("hqq_hip_gemm: dtype %d not covered (fp16 only for now)", dtype); return HQQ_ERR_UNSUPPORTED; }
  hipStream_t st = as_stream(stream);
  const int m = static_cast<int>(M), n = static_cast<int>(N), k = static_cast<int>(K), gs = static_cast<int>(group_size);
  // opt-in (HQQ_HIP_GEMM_WS=1): the wave-specialised 256x128 kernel.  Round 1 status: correct, 0.58-0.74 PFLOP/s — not yet
  // ahead of the single-role kernels below (0.65-0.83); PMC: MFMA busy 29 %, waves waiting 50 % of their cycles.
  if (getenv("HQQ_HIP_GEMM_WS") && static_cast<int64_t>((M + 255) / 256) * ((N + GB_N - 1) / GB_N) >= 512)
    return nbits == 4 ? launch_gemm_ws_f16<4>(x, Wq, scale, zero, bias, y, m, n, k, gs, st) : launch_gemm_ws_f16<2>(x, Wq, scale, zero, bias, y, m, n, k, gs, st);
  // 256-token tiles halve the dequantisation work per flop; keep 128 when M is too small to fill the chip with them
  const bool big = static_cast<int64_t>((M + 255) / 256) * ((N + GB_N - 1) / GB_N) >= 1536;   // >= 3 full waves of 256-token tiles
  if (nbits == 4) return big ? launch_gemm_f16<4, 256>(x, Wq, scale, zero, bias, y, m, n, k, gs, st) : launch_gemm_f16<4, 128>(x, Wq, scale, zero, bias, y, m, n, k, gs, st);
  return big ? launch_gemm_f16<2, 256>(x, Wq, scale, zero, bias, y, m, n, k, gs, st) : launch_gemm_f16<2, 128>(x, Wq, scale, zero, bias, y, m, n, k, gs, st);
}

int hqq_hip_forward(int nbits, const void* x, const void* Wq, const void* scale, const void* zero, const void* bias,
                    void* y, int64_t M, int64_t N, int64_t K, int64_t group_size, int dtype, void* stream) {
  if (M >= 1 && M <= (nbits == 3 ? 4 : HQQ_GEMV_MAX_M)) return hqq_hip_gemv(nbits, x, Wq, scale, zero, bias, y, M, N, K, group_size, dtype, stream);
  return hqq_hip_gemm(nbits, x, Wq, scale, zero, bias, y, M, N, K, group_size, dtype, stream);
}

}  // extern "C"
