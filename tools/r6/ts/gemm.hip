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

template <int NBITS, int S, int PER, int BN = GB_N>
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
    const int row = S * (BN / PER) + prow_in_tile;
    *reinterpret_cast<u32x4*>(ldsW + lds_off(row, kchunk16 * 2)) = u32x4{o[0], o[1], o[2], o[3]};
    *reinterpret_cast<u32x4*>(ldsW + lds_off(row, kchunk16 * 2 + 1)) = u32x4{o[4], o[5], o[6], o[7]};
    if constexpr (S + 1 < PER) DeqSlab<NBITS, S + 1, PER, BN>::run(w, z, s, ldsW, prow_in_tile, kchunk16);
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
// Large-M kernel: 256 tokens x (16*PER*8 = 256 features at 4 bits) per workgroup, 8 waves.  Weights never touch LDS: wave w owns
// 16 packed rows (-> 16*PER output features), loads their 16 packed bytes per lane and K-step straight in MFMA operand layout
// (lane (r = lane & 15, c = lane >> 4): row r, k = 16c .. 16c+15, as in gemv_mfma.hip), dequantises them exactly in registers
// and uses them as the A operand against all 256 tokens.  Only the activation tile goes through LDS (two 32 KiB stages, one
// workgroup barrier per K-step), staged cooperatively by all waves.
// Why (in-kernel cycle accounting, tools/gemm_lab.hip, on two producer/consumer variants of the LDS-staged design): writing the
// dequantised fp16 weight tile to LDS costs 4x its packed size in LDS-write bandwidth (64-85 B/clk/CU) and pins the producers at
// 2700-4400 cycles per K-step against 1000-2000 cycles of MFMA work; the CU's address/L1 path (64 B/clk) is the second limit,
// which the 256x256 tile relieves (40 KiB per step for 8.4 MFLOP).  Here every wave does both jobs — 64 exact-dequant VALU ops
// and 64 MFMAs per K-step — so the two pipes overlap inside each wave without a producer/consumer hand-off.
// ------------------------------------------------------------------------------------------------------------------
constexpr int RT_WAVES = 8, RT_THREADS = 64 * RT_WAVES;   // token-tile height BM: 256 at 4 bits, 128 at 2 bits (accumulator registers)

template <int NBITS, int S, int PER>
struct RtSlab {   // A fragments (two k-octets) of slab S from the lane's 16 packed bytes: exact, two fp16 roundings
  static __device__ __forceinline__ void run(const u32x4& w, const half_t (&z)[PER], const half_t (&s)[PER], h8_t (&a0)[PER], h8_t (&a1)[PER]) {
    const half2_t zz = {z[S], z[S]}, ss = {s[S], s[S]};
    uint32_t o[8];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      o[2 * d] = g_as_u32((g_levels<NBITS, S>(w[d]) - zz) * ss);            // bytes (4d+0, 4d+2)
      o[2 * d + 1] = g_as_u32((g_levels<NBITS, S>(w[d] >> 8) - zz) * ss);   // bytes (4d+1, 4d+3)
    }
    a0[S] = __builtin_bit_cast(h8_t, u32x4{o[0], o[1], o[2], o[3]});   // k = 16c + 0..7 (permuted inside the octet, like x in LDS)
    a1[S] = __builtin_bit_cast(h8_t, u32x4{o[4], o[5], o[6], o[7]});   // k = 16c + 8..15
    if constexpr (S + 1 < PER) RtSlab<NBITS, S + 1, PER>::run(w, z, s, a0, a1);
  }
};

template <int NBITS, int RT_BM>
__global__ __launch_bounds__(RT_THREADS, 2) void gemm_rt_f16_kernel(
    const half_t* __restrict__ x, const uint8_t* __restrict__ Wq, const half_t* __restrict__ scale,
    const half_t* __restrict__ zero, const half_t* __restrict__ bias, half_t* __restrict__ y,
    int M, int N, int K, int gs, int n_tiles) {
  constexpr int PER = 8 / NBITS;
  constexpr int PROWS = 16 * RT_WAVES;                    // packed rows per workgroup tile (128) -> 128*PER features
  constexpr int XSTAGE = RT_BM * GB_K * 2;                // 32 KiB
  constexpr int MT = RT_BM / 16;                          // token tiles
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];   // two x stages

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, c = lane >> 4;
  // XCD-aware tile order (workgroup b runs on XCD b % 8; speed only): all n tiles of a token tile on one XCD -> its x tile in one L2
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
  const int m0 = mt * RT_BM;
  const int G = K / gs;
  const int nk = K / GB_K;
  int prow = nt * PROWS + wave * 16 + r;                  // this lane's packed row
  const bool row_ok = prow < rows_per_slab;
  prow = row_ok ? prow : rows_per_slab - 1;               // ragged last tile: duplicate the last row (masked at the store)

  // x staging assignment: XCH consecutive 16-byte chunks of one row per thread (a row of the tile is 8 chunks)
  constexpr int XCH = RT_BM * 8 / RT_THREADS;             // 4 (BM = 256) or 2 (BM = 128)
  const int xr_ = tid / (8 / XCH), xc0 = (tid % (8 / XCH)) * XCH;
  const half_t* xsrc = x + static_cast<int64_t>(m0 + xr_ < M ? m0 + xr_ : 0) * K + xc0 * 8;
  const bool x_ok = m0 + xr_ < M;

  struct WStage { u32x4 w; half_t z[PER], s[PER]; };
  auto load_w = [&](WStage& st, int kt) {
    const int k0 = kt * GB_K;
    st.w = *reinterpret_cast<const u32x4*>(Wq + static_cast<int64_t>(prow) * K + k0 + c * 16);
    const int g = (k0 + c * 16) / gs;
#pragma unroll
    for (int s = 0; s < PER; ++s) {
      const int64_t q = static_cast<int64_t>(prow + s * rows_per_slab) * G + g;
      st.z[s] = zero[q];
      st.s[s] = scale[q];
    }
  };
  u32x4 xreg[XCH];
  auto load_x = [&](int kt) {
#pragma unroll
    for (int i = 0; i < XCH; ++i) xreg[i] = x_ok ? *reinterpret_cast<const u32x4*>(xsrc + kt * GB_K + i * 8) : u32x4{0u, 0u, 0u, 0u};
  };
  auto write_x = [&](uint8_t* ldsX) {
#pragma unroll
    for (int i = 0; i < XCH; ++i) *reinterpret_cast<u32x4*>(ldsX + lds_off(xr_, xc0 + i)) = g_permute_x8(xreg[i]);
  };

  f32x4 acc[PER][MT];
#pragma unroll
  for (int s = 0; s < PER; ++s)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[s][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  WStage wa, wb;
  load_w(wa, 0);
  load_x(0);
  if (nk > 1) load_w(wb, 1);
  write_x(lds);
  if (nk > 1) load_x(1);
  __syncthreads();

  auto step = [&](WStage& cur, int kt) {
    const uint8_t* ldsX = lds + (kt & 1) * XSTAGE;
    if (kt + 1 < nk) write_x(lds + ((kt + 1) & 1) * XSTAGE);       // registers hold x of step kt+1
    h8_t a0[PER], a1[PER];
    RtSlab<NBITS, 0, PER>::run(cur.w, cur.z, cur.s, a0, a1);
    if (kt + 2 < nk) { load_w(cur, kt + 2); load_x(kt + 2); }       // two steps ahead, in flight across the barrier
    // four token tiles at a time: 4*PER independent MFMAs between two MFMAs on the same accumulator (a dependent pair issued
    // back to back stalls for the full MFMA latency)
#pragma unroll
    for (int jh = 0; jh < MT; jh += 4) {
      h8_t b0[4], b1[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        b0[j] = *reinterpret_cast<const h8_t*>(ldsX + lds_off((jh + j) * 16 + r, c * 2));
        b1[j] = *reinterpret_cast<const h8_t*>(ldsX + lds_off((jh + j) * 16 + r, c * 2 + 1));
      }
#pragma unroll
      for (int s = 0; s < PER; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[s][jh + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[s], b0[j], acc[s][jh + j], 0, 0, 0);
#pragma unroll
      for (int s = 0; s < PER; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[s][jh + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[s], b1[j], acc[s][jh + j], 0, 0, 0);
    }
    __syncthreads();   // x stage (kt+1)&1 filled, stage kt&1 drained
  };
  int kt = 0;
  for (; kt + 1 < nk; kt += 2) {
    step(wa, kt);
    step(wb, kt + 1);
  }
  if (kt < nk) step(wa, kt);

  // ---- epilogue: D layout — lane holds packed rows 4c + i (i = 0..3) of the wave's 16, token r of token tile j ----
  const int p_base = nt * PROWS + wave * 16 + c * 4;
#pragma unroll
  for (int s = 0; s < PER; ++s) {
    if (p_base >= rows_per_slab) continue;
    const int n = s * rows_per_slab + p_base;
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const int m = m0 + j * 16 + r;
      if (m >= M) continue;
      half_t o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        o[i] = static_cast<half_t>(acc[s][j][i]);
        if (bias && p_base + i < rows_per_slab) o[i] = o[i] + bias[n + i];
      }
      half_t* dst = y + static_cast<int64_t>(m) * N + n;
      if (p_base + 3 < rows_per_slab) {
        *reinterpret_cast<u32x2*>(dst) = *reinterpret_cast<u32x2*>(o);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (p_base + i < rows_per_slab) dst[i] = o[i];
      }
    }
  }
}

template <int NBITS, int RT_BM>
static int launch_gemm_rt_f16(const void* x, const void* Wq, const void* scale, const void* zero, const void* bias, void* y,
                              int M, int N, int K, int gs, hipStream_t st) {
  constexpr int PER = 8 / NBITS;
  const int rows_per_slab = N / PER;
  const int n_tiles = (rows_per_slab + 16 * RT_WAVES - 1) / (16 * RT_WAVES);
  const int m_tiles = (M + RT_BM - 1) / RT_BM;
  const int64_t blocks = static_cast<int64_t>(n_tiles) * m_tiles;
  if (blocks > INT32_MAX) { set_error("hqq_hip_gemm: grid too large"); return HQQ_ERR_SHAPE; }
  constexpr int lds_bytes = 2 * RT_BM * GB_K * 2;
  hipLaunchKernelGGL((gemm_rt_f16_kernel<NBITS, RT_BM>), dim3(static_cast<unsigned>(blocks)), dim3(RT_THREADS), lds_bytes, st,
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

// gemm_pipe.hip: the pipelined split-K kernel for the rows between decode and long prefill
struct GpPlan;
size_t gemm_pipe_workspace_bytes(int nbits, int64_t M, int64_t N, int64_t K, uint32_t opts);
bool gemm_pipe_covers(int nbits, int64_t M, int64_t N, int64_t K, int64_t gs, int dtype);
bool gemm_pipe_wins(int nbits, int64_t M, int64_t N, int64_t K);
bool skinny_covers(int nbits, int64_t M, int64_t K, int64_t group_size, const int64_t* N, int n_layers);   // skinny.hip
void gemm_pipe_describe(int nbits, int64_t M, int64_t N, int64_t K, uint32_t opts, int out[8]);
int gemm_pipe_run(int nbits, const void* x, const void* Wq, const void* scale, const void* zero, const void* bias, void* y,
                  int64_t M, int64_t N, int64_t K, int64_t gs, int dtype, uint32_t opts, void* workspace, size_t workspace_bytes, hipStream_t st);
size_t gemm_pipe_workspace_bytes_grouped(int nbits, int n_layers, const int64_t* N, int64_t M, int64_t K, uint32_t opts);
int gemm_pipe_run_grouped(int nbits, int n_layers, const void* x, const void* const* Wq, const void* const* scale, const void* const* zero, const void* const* bias,
                          void* const* y, const int64_t* N, int64_t M, int64_t K, int64_t gs, int dtype, uint32_t opts, void* workspace, size_t workspace_bytes, hipStream_t st);

// which fused GEMM serves a call: the pipelined kernel (gemm_pipe.hip) wherever it applies — it is ahead of the output-tile kernels
// below at every M (0.8-1.14 PFLOP/s against 0.5-0.84 from 2048 rows on, 2-4x below 512) —, the output-tile kernels for the group sizes
// and K it does not cover
#ifndef GEMM_PIPE_MAX_M_VALUE
#define GEMM_PIPE_MAX_M_VALUE (int64_t(1) << 40)
#endif
constexpr int64_t GEMM_PIPE_MAX_M = GEMM_PIPE_MAX_M_VALUE;
static bool use_pipe(int nbits, int64_t M, int64_t N, int64_t K, int64_t gs, int dtype, uint32_t opts) {
  if (opts & (HQQ_OPT_GEMM_REGTILE | HQQ_OPT_GEMM_CLASSIC)) return false;
  if (nbits == 3 && !(opts & HQQ_OPT_W3S)) return false;   // (the reference's 3-bit container has no fused GEMM; the stream layout runs like a 4-bit layer)
  return (M <= GEMM_PIPE_MAX_M || nbits == 8) && gemm_pipe_covers(nbits, M, N, K, gs, dtype);   // (8-bit: the only fused GEMM there is)
}

}  // namespace hqq

using namespace hqq;

extern "C" {

size_t hqq_hip_gemm_workspace_bytes(int nbits, int64_t M, int64_t N, int64_t K, int64_t group_size, int dtype, uint32_t opts) {
  if (M < 1 || N <= 0 || K <= 0 || group_size <= 0) return 0;
  return use_pipe(nbits, M, N, K, group_size, dtype, opts) ? gemm_pipe_workspace_bytes(nbits, M, N, K, opts) : 0;
}

size_t hqq_hip_forward_workspace_bytes(int nbits, int64_t M, int64_t N, int64_t K, int64_t group_size, int dtype, uint32_t opts) {
  if (M < 1 || N <= 0 || K <= 0 || group_size <= 0) return 0;
  // (the same test as hqq_hip_forward's dispatch)
  if (M <= (nbits == 3 ? 4 : HQQ_GEMV_MAX_M)) return hqq_hip_gemv_workspace_bytes(nbits, 1, &N, M, K, group_size, dtype, opts);
  const bool w3s = nbits == 3 && (opts & HQQ_OPT_W3S);   // the 3-bit stream layout: two row slabs per packed row, served like a 4-bit layer
  if (M <= HQQ_GEMV_MAX_M_SKINNY && (dtype == HQQ_F16 || dtype == HQQ_BF16) && (nbits == 8 || nbits == 4 || nbits == 2 || w3s) && group_size == 64 && K % 256 == 0 && K >= 512 &&
      N % (w3s ? 2 : 8 / nbits) == 0)
    return hqq_hip_gemv_workspace_bytes(nbits, 1, &N, M, K, group_size, dtype, opts);
  return hqq_hip_gemm_workspace_bytes(nbits, M, N, K, group_size, dtype, opts);
}

int hqq_hip_gemm_plan(int nbits, int64_t M, int64_t N, int64_t K, int64_t group_size, int dtype, uint32_t opts, int* out8) {
  if (!out8) return HQQ_ERR_SHAPE;
  for (int i = 0; i < 8; ++i) out8[i] = 0;
  if (M < 1 || N <= 0 || K <= 0 || group_size <= 0 || !use_pipe(nbits, M, N, K, group_size, dtype, opts)) return HQQ_ERR_UNSUPPORTED;
  gemm_pipe_describe(nbits, M, N, K, opts, out8);
  return 0;
}

int hqq_hip_forward_prefers_fused(int nbits, int64_t M, int64_t N, int64_t K, int64_t group_size, int dtype) {
  if (M < 1 || N <= 0 || K <= 0 || group_size <= 0) return 0;
  if (M <= HQQ_GEMV_MAX_M) return 1;          // decode: always the weight-streaming kernels (hqq_hip_gemv reports what it does not cover)
  if (nbits == 3) return 0;   // the reference's 3-bit container has no fused kernel beyond the decode rows; a layer in the stream layout asks with nbits = 4 (same kernels, same plan)
  if (M <= HQQ_GEMV_MAX_M_SKINNY && (dtype == HQQ_F16 || dtype == HQQ_BF16) && skinny_covers(nbits, M, K, group_size, &N, 1)) return 1;
  return gemm_pipe_covers(nbits, M, N, K, group_size, dtype) && gemm_pipe_wins(nbits, M, N, K) ? 1 : 0;
}

int hqq_hip_gemm(int nbits, const void* x, const void* Wq, const void* scale, const void* zero, const void* bias,
                 void* y, int64_t M, int64_t N, int64_t K, int64_t group_size, int dtype, uint32_t opts, void* workspace, size_t workspace_bytes,
                 void* stream) {
  clear_stale_error();
  if (opts & ~HQQ_OPT_ALL) { set_error("hqq_hip_gemm: unknown option bits 0x%x", opts & ~HQQ_OPT_ALL); return HQQ_ERR_SHAPE; }
  if (M < 1 || N <= 0 || K <= 0 || group_size <= 0 || K % group_size) { set_error("hqq_hip_gemm: bad M/N/K/group_size"); return HQQ_ERR_SHAPE; }
  if (M > INT32_MAX || N > INT32_MAX || K > INT32_MAX || N * (K / group_size) > INT32_MAX) { set_error("hqq_hip_gemm: size overflow"); return HQQ_ERR_SHAPE; }
  if (!aligned16(x) || !aligned16(Wq) || !aligned16(y)) { set_error("hqq_hip_gemm: pointers must be 16-byte aligned"); return HQQ_ERR_ALIGN; }
  hipStream_t st = as_stream(stream);
  if (use_pipe(nbits, M, N, K, group_size, dtype, opts)) return gemm_pipe_run(nbits, x, Wq, scale, zero, bias, y, M, N, K, group_size, dtype, opts, workspace, workspace_bytes, st);
  if (nbits != 4 && nbits != 2) { set_error("hqq_hip_gemm: nbits=%d not covered by the fused GEMM", nbits); return HQQ_ERR_UNSUPPORTED; }
  const int per = 8 / nbits;
  if (N % per || (N / per) % 4 || group_size % 16 || K % GB_K) {
    set_error("hqq_hip_gemm: needs N %% %d == 0, K %% 64 == 0, group_size %% 16 == 0 (got N=%lld K=%lld gs=%lld)", 4 * per, (long long)N, (long long)K, (long long)group_size);
    return HQQ_ERR_UNSUPPORTED;
  }
  if (dtype != HQQ_F16) { set_error("hqq_hip_gemm: dtype %d not covered (fp16 only for now)", dtype); return HQQ_ERR_UNSUPPORTED; }
  const int m = static_cast<int>(M), n = static_cast<int>(N), k = static_cast<int>(K), gs = static_cast<int>(group_size);
  // opt-in (HQQ_OPT_GEMM_REGTILE): register-tile kernel (weights never touch LDS).  Round-1 status: correct, 0.62-0.81 PFLOP/s — level
  // with the LDS-staged kernels below (0.65-0.83), not ahead; PMC: waves stall on issue 33 % and wait 45 % of their cycles.
  if ((opts & HQQ_OPT_GEMM_REGTILE) && static_cast<int64_t>((M + 255) / 256) * ((N / per + 127) / 128) >= 256)
    return nbits == 4 ? launch_gemm_rt_f16<4, 256>(x, Wq, scale, zero, bias, y, m, n, k, gs, st) : launch_gemm_rt_f16<2, 128>(x, Wq, scale, zero, bias, y, m, n, k, gs, st);
  // 256-token tiles halve the dequantisation work per flop; keep 128 when M is too small to fill the chip with them
  const bool big = static_cast<int64_t>((M + 255) / 256) * ((N + GB_N - 1) / GB_N) >= 1536;   // >= 3 full waves of 256-token tiles
  if (nbits == 4) return big ? launch_gemm_f16<4, 256>(x, Wq, scale, zero, bias, y, m, n, k, gs, st) : launch_gemm_f16<4, 128>(x, Wq, scale, zero, bias, y, m, n, k, gs, st);
  return big ? launch_gemm_f16<2, 256>(x, Wq, scale, zero, bias, y, m, n, k, gs, st) : launch_gemm_f16<2, 128>(x, Wq, scale, zero, bias, y, m, n, k, gs, st);
}

// A group of layers that read the same x (q | k | v, gate | up) through ONE launch of the pipelined fused GEMM (+ one split-K reduce): ABI 8
static bool group_on_pipe(int nbits, int n_layers, const int64_t* N, int64_t M, int64_t K, int64_t group_size, int dtype, uint32_t opts) {
  if (!N || n_layers < 1 || n_layers > HQQ_GEMV_MAX_GROUP || M < 1 || K <= 0 || group_size <= 0) return false;
  for (int i = 0; i < n_layers; ++i)
    if (N[i] <= 0 || !use_pipe(nbits, M, N[i], K, group_size, dtype, opts)) return false;
  return true;
}
int hqq_hip_gemm_grouped_covers(int nbits, int n_layers, const int64_t* N, int64_t M, int64_t K, int64_t group_size, int dtype, uint32_t opts) {
  return group_on_pipe(nbits, n_layers, N, M, K, group_size, dtype, opts) ? 1 : 0;
}
size_t hqq_hip_gemm_grouped_workspace_bytes(int nbits, int n_layers, const int64_t* N, int64_t M, int64_t K, int64_t group_size, int dtype, uint32_t opts) {
  return group_on_pipe(nbits, n_layers, N, M, K, group_size, dtype, opts) ? gemm_pipe_workspace_bytes_grouped(nbits, n_layers, N, M, K, opts) : 0;
}
int hqq_hip_gemm_grouped(int nbits, int n_layers, const void* x, const void* const* Wq, const void* const* scale, const void* const* zero, const void* const* bias,
                         void* const* y, const int64_t* N, int64_t M, int64_t K, int64_t group_size, int dtype, uint32_t opts, void* workspace, size_t workspace_bytes,
                         void* stream) {
  clear_stale_error();
  if (opts & ~HQQ_OPT_ALL) { set_error("hqq_hip_gemm_grouped: unknown option bits 0x%x", opts & ~HQQ_OPT_ALL); return HQQ_ERR_SHAPE; }
  if (n_layers < 1 || n_layers > HQQ_GEMV_MAX_GROUP) { set_error("hqq_hip_gemm_grouped: n_layers=%d outside [1,%d]", n_layers, HQQ_GEMV_MAX_GROUP); return HQQ_ERR_SHAPE; }
  if (!x || !Wq || !scale || !zero || !y || !N) { set_error("hqq_hip_gemm_grouped: null argument"); return HQQ_ERR_SHAPE; }
  if (M < 1 || K <= 0 || group_size <= 0 || K % group_size) { set_error("hqq_hip_gemm_grouped: bad M/K/group_size"); return HQQ_ERR_SHAPE; }
  if (M > INT32_MAX || K > INT32_MAX) { set_error("hqq_hip_gemm_grouped: size overflow"); return HQQ_ERR_SHAPE; }
  int64_t ntot = 0;
  for (int i = 0; i < n_layers; ++i) {
    if (N[i] <= 0 || N[i] > INT32_MAX || N[i] * (K / group_size) > INT32_MAX) { set_error("hqq_hip_gemm_grouped: bad N / size overflow"); return HQQ_ERR_SHAPE; }
    if (!Wq[i] || !scale[i] || !zero[i] || !y[i]) { set_error("hqq_hip_gemm_grouped: null layer pointer"); return HQQ_ERR_SHAPE; }
    if (!aligned16(Wq[i]) || !aligned16(y[i])) { set_error("hqq_hip_gemm_grouped: pointers must be 16-byte aligned"); return HQQ_ERR_ALIGN; }
    ntot += N[i];
  }
  if (!aligned16(x)) { set_error("hqq_hip_gemm_grouped: pointers must be 16-byte aligned"); return HQQ_ERR_ALIGN; }
  if (ntot > INT32_MAX) { set_error("hqq_hip_gemm_grouped: size overflow"); return HQQ_ERR_SHAPE; }
  if (!group_on_pipe(nbits, n_layers, N, M, K, group_size, dtype, opts)) {
    set_error("hqq_hip_gemm_grouped: every layer of the group must be served by the pipelined fused GEMM (fp16 / bf16, nbits 8 / 4 / 2 or the 3-bit stream layout, group_size 64, K %% 128 == 0)");
    return HQQ_ERR_UNSUPPORTED;
  }
  return gemm_pipe_run_grouped(nbits, n_layers, x, Wq, scale, zero, bias, y, N, M, K, group_size, dtype, opts, workspace, workspace_bytes, as_stream(stream));
}

int hqq_hip_forward(int nbits, const void* x, const void* Wq, const void* scale, const void* zero, const void* bias,
                    void* y, int64_t M, int64_t N, int64_t K, int64_t group_size, int dtype, uint32_t opts, void* workspace, size_t workspace_bytes,
                    void* stream) {
  if (M >= 1 && M <= (nbits == 3 ? 4 : HQQ_GEMV_MAX_M)) return hqq_hip_gemv(nbits, x, Wq, scale, zero, bias, y, M, N, K, group_size, dtype, opts, workspace, workspace_bytes, stream);
  // a batch of 17..64 rows is still weight-streaming work: the skinny-GEMM kernel where it applies (same conditions as skinny_covers)
  const bool w3s = nbits == 3 && (opts & HQQ_OPT_W3S);
  if (M <= HQQ_GEMV_MAX_M_SKINNY && (dtype == HQQ_F16 || dtype == HQQ_BF16) && (nbits == 8 || nbits == 4 || nbits == 2 || w3s) && group_size == 64 && K % 256 == 0 && K >= 512 &&
      N % (w3s ? 2 : 8 / nbits) == 0)
    return hqq_hip_gemv(nbits, x, Wq, scale, zero, bias, y, M, N, K, group_size, dtype, opts, workspace, workspace_bytes, stream);
  return hqq_hip_gemm(nbits, x, Wq, scale, zero, bias, y, M, N, K, group_size, dtype, opts, workspace, workspace_bytes, stream);
}

}  // extern "C"
