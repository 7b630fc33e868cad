// gemm_dense.hip — y[M,N] = x[M,K] . Wd[N,K]^T (+ bias): the dense contraction of HQQLinear.matmul (hqq/core/quantize.py:880-882,
// torch.matmul(x, W.t())) on the matrix cores, fp16 / bf16, fp32 accumulation, gfx950.  The GEMM half of the prefill path: hqq_hip_dequantize
// rebuilds the layer's weights ONCE (bit-identical to Quantizer.dequantize), this kernel contracts them with any number of tokens —
// no library call on the hot path (round 3's long-prompt route was hqq_hip_dequantize + hipBLASLt through torch.matmul).
//
// Why not the fused kernel here (gemm_pipe.hip): it rebuilds every weight once per 256-token tile — 32 times at 8192 tokens, 1.4 VALU per
// MFMA, B-fragment reads as long as the MFMAs — and measured 1.07-1.10 PFLOP/s where dequantise-once + a plain GEMM gives 1.22-1.24
// (profiles/r03_pipe8k_ablation.txt).  Beyond ~1000 tokens rebuilding once and streaming fp16 weights wins; below, gemm_pipe.hip.
//
// Structure (the 256 x 256 x 64 tile of the CDNA guide, four phases per K tile, every byte by LDS-DMA, counted vmcnt, one raw barrier per phase):
//   tile     256 tokens x 256 features x 64 k per workgroup; 8 waves as 2 (tokens) x 4 (features): a wave owns 128 tokens x 64 features =
//            8 x 4 MFMA tiles of 16 x 16 (v_mfma_f32_16x16x32: A operand = 16 features, B operand = 16 tokens, so a lane ends up with 4
//            consecutive features of one token), 128 accumulator registers.
//   LDS      two K-tile buffers of [256 x rows | 256 Wd rows] x 128 bytes (64 KiB each).  A 1-KiB DMA piece = 8 rows x 128 bytes; the 16-byte
//            chunk a position holds is XOR-ed with a function of the row (gemm_pipe.hip's map) on the SOURCE address, so that the fragment
//            reads (lane (r, c): chunks 2c, 2c + 1 of row 16 j + r) are conflict-free.
//   phases   a K tile's 64 MFMAs per wave go in four quadrants (token half x feature half of the wave's sub-tile), ordered so that the
//            fragments a quadrant needs beyond its predecessor's are few and the last quadrant (X1, W1) leaves the registers of the next
//            fragments a quadrant needs beyond its predecessor's are few:   Q0 = (X0, W0)   Q1 = (X0, W1)   Q2 = (X1, W1)   Q3 = (X1, W0)
//            phase q:  LOAD part: wait (the DMA pieces the NEXT phase reads have landed) -> issue the DMA pieces whose LDS rows were read two
//                      phases ago -> issue the reads of quadrant q's missing fragments | barrier | reads returned, 16 MFMAs | barrier.
//            The two waves of a SIMD run half a phase apart: one's LOAD part beside the other's MFMA part (the guide's ping-pong).
//            A staging unit = the rows all waves read in one phase (X0: tokens 128 wm + [0, 64); W0: features 64 wn + [0, 32); ...): 16
//            pieces, 2 per wave.  Every unit is issued 6-8 phases (1.5-2 K tiles) before it is read; 8 pieces stay in flight across
//            every barrier (s_waitcnt vmcnt(8), never 0 in the loop).
//   order    workgroup -> tile through an XCD-aware bijection (a band of token tiles x all feature tiles per XCD: its L2 holds the band's
//            x rows and one pass over Wd).
#include <type_traits>

#include "hqq_common.h"

namespace hqq {
namespace gd {

constexpr int BM = 256, BN = 256, BK = 64;
#ifndef GD_BAND
#define GD_BAND 4
#endif
constexpr int NWAVES = 8, NT = NWAVES * 64;
constexpr int XBYTES = BM * BK * 2, WBYTES = BN * BK * 2, BUF = XBYTES + WBYTES;   // 32 KiB + 32 KiB per K tile

struct Args {
  const uint16_t* x;
  const uint16_t* w;
  const uint16_t* bias;
  uint16_t* y;
  int M, N, K, m_tiles, n_tiles;
};

typedef __attribute__((address_space(3))) void* lds_t;
typedef const __attribute__((address_space(1))) void* glb_t;
__device__ __forceinline__ void dma16(const void* src, uint8_t* lds_wave_base) { __builtin_amdgcn_global_load_lds((glb_t)src, (lds_t)lds_wave_base, 16, 0, 0); }
__device__ __forceinline__ int swz(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 2); }

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));

template <bool BF>
__global__ __launch_bounds__(NT, 2) void dense_gemm_kernel(const Args a) {
  extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];   // [2 buffers][X tile | W tile]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int r = lane & 15, c = lane >> 4;
  // ---- XCD-aware tile order (workgroup b runs on XCD b % 8 — observed; a speed assumption only): XCD x gets a contiguous run of logical
  //      tiles; logical tile L = token tile (L / n_tiles), feature tile (L % n_tiles).  Bijective for any grid size ----
  int mt, nt;
  {
    const int nwg = static_cast<int>(gridDim.x), h = static_cast<int>(blockIdx.x), xcd = h & 7, idx = h >> 3;
    const int q8 = nwg >> 3, r8 = nwg & 7;
    const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    // inside the run: bands of GD_BAND token tiles, walked down the band first — the 32 workgroups an XCD runs at a time cover 4 token tiles x 8
    // feature tiles (12 operand slices per K tile through its L2) rather than 2 x 16 (18): +2-5 % (profiles/r04_dense_gemm_lab.txt)
    const int band = L / (GD_BAND * a.n_tiles), idxb = L - band * GD_BAND * a.n_tiles;
    const int gm = a.m_tiles - band * GD_BAND < GD_BAND ? a.m_tiles - band * GD_BAND : GD_BAND;
    nt = idxb / gm;
    mt = band * GD_BAND + (idxb - nt * gm);
  }
  const int M = a.M, N = a.N, K = a.K, nk = K / BK;
  const int m0 = mt * BM, n0 = nt * BN;

  // ---- DMA sources.  Staging unit u (0: X0, 1: W0, 2: W1, 3: X1), piece p (0, 1) of this wave: 8 rows x 128 bytes.
  //      Unit rows (workgroup tile): X0: 128 g + [0, 64), X1: 128 g + [64, 128), g = 0, 1;  W0: 64 g + [0, 32), W1: 64 g + [32, 64), g = 0..3.
  //      Piece q = 2 wave + p of a unit (0..15) covers 8 consecutive rows of it, first row row0 (bit 3 of row0 = p in every unit).
  //      Buffer loads: one descriptor per operand over the tile's valid rows (rows past the end of x / Wd read as zeros: no clamping, their
  //      outputs are never stored), a per-lane offset that depends on p only (row inside the piece, swizzled chunk) and a wave-uniform
  //      offset per (unit, piece) + 128 bytes per K tile ----
  const int xrows = M - m0 < BM ? M - m0 : BM, wrows = N - n0 < BN ? N - n0 : BN;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.x + static_cast<int64_t>(m0) * K), 0, xrows * K * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.w + static_cast<int64_t>(n0) * K), 0, wrows * K * 2, 0x00020000);
  int voff[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) voff[p] = (lane >> 3) * K * 2 + (((lane & 7) ^ (((lane >> 4) & 1) | (p << 2))) << 4);   // chunk ^ swz(row0 + (lane >> 3))
  int soff[4][2], dst[4][2];   // wave-uniform: byte offset of the piece's first row in the operand / inside an LDS buffer
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int q = 2 * wave + p;
      const bool is_x = (u == 0 || u == 3);
      const int sub = (u == 0 || u == 1) ? 0 : 1;
      const int row0 = is_x ? (((8 * q) >> 6) * 128 + sub * 64 + ((8 * q) & 63)) : (((8 * q) >> 5) * 64 + sub * 32 + ((8 * q) & 31));
      soff[u][p] = row0 * K * 2;
      dst[u][p] = (is_x ? 0 : XBYTES) + row0 * 128;
    }
  auto stage = [&](int u, int kt) {   // unit u of K tile kt (kt < nk) into buffer kt & 1
    uint8_t* base = lds + (kt & 1) * BUF;
    const bool is_x = (u == 0 || u == 3);
#pragma unroll
    for (int p = 0; p < 2; ++p)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(is_x ? rx : rw, (lds_t)(base + dst[u][p]), 16, voff[p], soff[u][p] + kt * (BK * 2), 0, 0);
  };

  // ---- fragments: X (tokens, the MFMA's B operand): 8 tiles of 16 tokens, sub-half s = tiles 4 s .. 4 s + 3;
  //      W (features, A operand): 4 tiles, sub-half = 2.  Both sub-halves of both operands have their own registers (96): the reads of a
  //      quadrant are issued one phase ahead, under the MFMAs of the quadrant before it ----
  u32x4 xf[2][4][2], wf[2][2][2];   // [sub-half][tile][k half (chunks 2c, 2c + 1)]
  const int xrd = (wm * 128 + r) * 128, wrd = XBYTES + (wn * 64 + r) * 128;   // swz(row) = swz(r): tile bases are multiples of 16 rows
  const int ch0 = ((2 * c) ^ swz(r)) << 4, ch1 = ((2 * c + 1) ^ swz(r)) << 4;
  auto read_x = [&](int s, int kt) {
    const uint8_t* base = lds + (kt & 1) * BUF + xrd;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      xf[s][t][0] = *reinterpret_cast<const u32x4*>(base + (4 * s + t) * 2048 + ch0);
      xf[s][t][1] = *reinterpret_cast<const u32x4*>(base + (4 * s + t) * 2048 + ch1);
    }
  };
  auto read_w = [&](int s, int kt) {
    const uint8_t* base = lds + (kt & 1) * BUF + wrd;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      wf[s][t][0] = *reinterpret_cast<const u32x4*>(base + (2 * s + t) * 2048 + ch0);
      wf[s][t][1] = *reinterpret_cast<const u32x4*>(base + (2 * s + t) * 2048 + ch1);
    }
  };
  f32x4 acc[4][8];   // [feature tile][token tile]
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[f][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto mfma = [&](const u32x4& A, const u32x4& B, f32x4 C) {
    if constexpr (BF) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8, A), __builtin_bit_cast(b8, B), C, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, A), __builtin_bit_cast(h8, B), C, 0, 0, 0);
  };
  auto quadrant = [&](int xs, int ws) {   // 16 MFMAs: 2 feature tiles x 4 token tiles x 2 k halves
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[2 * ws + f][4 * xs + t] = mfma(wf[ws][f][h], xf[xs][t][h], acc[2 * ws + f][4 * xs + t]);
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- prologue: K tiles 0 and 1 whole, drained once; the fragments of the first quadrant ----
  // (unit numbering: 0 = X0, 1 = W0, 2 = W1, 3 = X1)
  stage(0, 0); stage(1, 0); stage(2, 0); stage(3, 0);
  if (nk > 1) { stage(0, 1); stage(1, 1); stage(2, 1); stage(3, 1); }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  read_x(0, 0);
  read_w(0, 0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // ---- main loop.  A phase = a LOAD part (wait, stage one unit, issue the fragment reads of the NEXT quadrant) | barrier | an MFMA part
  //      (16 MFMAs of this phase's quadrant, whose fragments were read a phase ago; then: this phase's reads have returned) | barrier.
  //      Quadrants of K tile t, with A = t & 1 the W sub-half the tile starts with and B = 1 - A:
  //            Q0 = (X0, W_A)   Q1 = (X0, W_B)   Q2 = (X1, W_B)   Q3 = (X1, W_A)           (the next tile starts with the sub-half this one ends
  //      with being idle: consecutive quadrants share one operand, and the fragments a load part reads are never those its own MFMA part uses)
  //        load part of phase   0: stages X0 (t + 2),  reads W_B (t)        1: stages W_A (t + 2), reads X1 (t)
  //                             2: stages W_B (t + 2), reads X0 (t + 1)     3: stages X1 (t + 2),  reads W_B (t + 1)
  //      The two waves of a SIMD — wave w and w + 4, token halves wm = 0 / 1 — run half a phase apart (the wm = 1 half takes one barrier
  //      more up front, the other one more at the end): one's load part (60-185 cycles per DMA piece, the read issue) sits beside the
  //      other's MFMA part (the guide's ping-pong), and no wave waits for an LDS round trip: reads are issued a phase before their use.
  //      Hazards with the halves half a phase apart:
  //        write after read   a load part's reads have returned when its phase ends (the lgkmcnt(0) after the MFMAs: free, the reads are
  //                           ~300 cycles old by then); the later half's phase ends one barrier after the earlier half's next load part
  //                           begins: a unit is staged TWO phases after the phase that read its rows (every unit above is);
  //        read after write   a unit must have landed for EVERY wave before ANY wave reads it, and the other half reads half a phase away from
  //                           this wave's own wait: a wave waits, in a load part, for the unit the NEXT load part reads.
  //      Every unit is issued 6 phases (1.5 K tiles) before it is read and 5 before its wait: 4 load parts x 2 pieces issued in between,
  //      s_waitcnt vmcnt(8) in every load part (before its own stage), never 0 in the steady loop ----
  auto tile = [&](auto parity, int t) {
    constexpr int A = decltype(parity)::value, B = 1 - A;
    const bool st = t + 2 < nk, nx = t + 1 < nk;
    // phase 0
    if (st) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (st) stage(0, t + 2);
    read_w(B, t);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    quadrant(0, A);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // phase 1
    if (st) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (st) stage(1 + A, t + 2);
    read_x(1, t);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    quadrant(0, B);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // phase 2
    if (st) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (st) stage(1 + B, t + 2);
    if (nx) read_x(0, t + 1);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    quadrant(1, B);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // phase 3
    if (st) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (st) stage(3, t + 2);
    if (nx) read_w(B, t + 1);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    quadrant(1, A);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  if (wm == 1) __builtin_amdgcn_s_barrier();
  for (int t = 0; t < nk; t += 2) {
    tile(std::integral_constant<int, 0>{}, t);
    if (t + 1 < nk) tile(std::integral_constant<int, 1>{}, t + 1);
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();

  // ---- epilogue.  A lane (r, c) ends with 4 consecutive features (16 ft + 4 c + i) of token r of every tile: 8 bytes, 32 contiguous bytes per
  //      token over the four c lanes.  The 4 x 4 transpose (feature tile ft <-> lane group c) through v_permlane32_swap / v_permlane16_swap
  //      gives the lane 16 consecutive features (16 c + 4 j + i) of its token: two 16-byte stores, 128 contiguous bytes per token row and
  //      half as many store instructions (the store tail is issue-bound) ----
  const bool bf = BF;
  const bool wide = (N & 7) == 0;
#pragma unroll
  for (int tt = 0; tt < 8; ++tt) {
    const int m = m0 + wm * 128 + tt * 16 + r;
    uint32_t o[4][2];   // [feature tile][features 4 c + {0, 1} | {2, 3}]
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) {
      const int n = n0 + wn * 64 + ft * 16 + 4 * c;
      uint16_t e[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float v = acc[ft][tt][i];
        if (bf) {
          uint16_t h = f32_to_bf16(v);
          if (a.bias && n + i < N) h = f32_to_bf16(bf16_to_f32(h) + bf16_to_f32(a.bias[n + i]));
          e[i] = h;
        } else {
          half_t h = static_cast<half_t>(v);
          if (a.bias && n + i < N) h = h + reinterpret_cast<const half_t*>(a.bias)[n + i];   // `out += bias` on the rounded matmul result (quantize.py:896-897)
          e[i] = __builtin_bit_cast(uint16_t, h);
        }
      }
      o[ft][0] = static_cast<uint32_t>(e[0]) | (static_cast<uint32_t>(e[1]) << 16);
      o[ft][1] = static_cast<uint32_t>(e[2]) | (static_cast<uint32_t>(e[3]) << 16);
    }
    if (wide) {
#pragma unroll
      for (int d = 0; d < 2; ++d) {
#pragma unroll
        for (int f0 = 0; f0 < 2; ++f0) {   // bit 1 of the tile index <-> bit 1 of the lane group
          const auto sw = __builtin_amdgcn_permlane32_swap(o[f0][d], o[2 + f0][d], false, false);
          o[f0][d] = sw[0]; o[2 + f0][d] = sw[1];
        }
#pragma unroll
        for (int f1 = 0; f1 < 2; ++f1) {   // bit 0 <-> bit 0
          const auto sw = __builtin_amdgcn_permlane16_swap(o[2 * f1][d], o[2 * f1 + 1][d], false, false);
          o[2 * f1][d] = sw[0]; o[2 * f1 + 1][d] = sw[1];
        }
      }
      // o[j][.] = features 16 c + 4 j + {0..3} of token r
      const int n = n0 + wn * 64 + 16 * c;
      if (m < M && n < N) {
        uint16_t* dstp = a.y + static_cast<int64_t>(m) * N + n;
        if (n + 15 < N) {
          __builtin_nontemporal_store(u32x4{o[0][0], o[0][1], o[1][0], o[1][1]}, reinterpret_cast<u32x4*>(dstp));
          __builtin_nontemporal_store(u32x4{o[2][0], o[2][1], o[3][0], o[3][1]}, reinterpret_cast<u32x4*>(dstp + 8));
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (n + 4 * j + 3 < N) *reinterpret_cast<u32x2*>(dstp + 4 * j) = u32x2{o[j][0], o[j][1]};
        }
      }
    } else if (m < M) {
#pragma unroll
      for (int ft = 0; ft < 4; ++ft) {
        const int n = n0 + wn * 64 + ft * 16 + 4 * c;
        if (n + 3 < N) *reinterpret_cast<u32x2*>(a.y + static_cast<int64_t>(m) * N + n) = u32x2{o[ft][0], o[ft][1]};   // N % 4 == 0: whole or nothing
      }
    }
  }
}

}  // namespace gd
}  // namespace hqq

using namespace hqq;

// y[M, N] = x[M, K] . Wd[N, K]^T (+ bias[N]); fp16 / bf16, fp32 accumulation, one rounding (+ one for the bias add).  K % 64 == 0, N % 4 == 0.
extern "C" int hqq_hip_gemm_dense(const void* x, const void* Wd, const void* bias, void* y, int64_t M, int64_t N, int64_t K, int dtype, void* stream) {
  clear_stale_error();
  if (dtype != HQQ_F16 && dtype != HQQ_BF16) { set_error("hqq_hip_gemm_dense: dtype %d not covered (fp16 / bf16)", dtype); return HQQ_ERR_UNSUPPORTED; }
  if (!x || !Wd || !y || M < 1 || N < 1 || K < gd::BK || M > INT32_MAX || N > INT32_MAX || K > INT32_MAX) { set_error("hqq_hip_gemm_dense: bad arguments"); return HQQ_ERR_SHAPE; }
  if (K % gd::BK || N % 4) { set_error("hqq_hip_gemm_dense: needs K %% 64 == 0 and N %% 4 == 0 (got N=%lld K=%lld)", (long long)N, (long long)K); return HQQ_ERR_UNSUPPORTED; }
  if (!aligned16(x) || !aligned16(Wd) || !aligned16(y)) { set_error("hqq_hip_gemm_dense: pointers must be 16-byte aligned"); return HQQ_ERR_ALIGN; }
  gd::Args a;
  a.x = static_cast<const uint16_t*>(x); a.w = static_cast<const uint16_t*>(Wd); a.bias = static_cast<const uint16_t*>(bias); a.y = static_cast<uint16_t*>(y);
  a.M = static_cast<int>(M); a.N = static_cast<int>(N); a.K = static_cast<int>(K);
  a.m_tiles = static_cast<int>((M + gd::BM - 1) / gd::BM);
  a.n_tiles = static_cast<int>((N + gd::BN - 1) / gd::BN);
  const int64_t blocks = static_cast<int64_t>(a.m_tiles) * a.n_tiles;
  if (blocks > INT32_MAX || K > (1 << 21)) { set_error("hqq_hip_gemm_dense: problem too large (tiles %lld, K %lld; the 32-bit buffer offsets hold 256 rows of K <= 2^21)", (long long)blocks, (long long)K); return HQQ_ERR_SHAPE; }
  constexpr int lds_bytes = 2 * gd::BUF;
  static LdsRaised raised[2];
  const void* kern = dtype == HQQ_BF16 ? reinterpret_cast<const void*>(&gd::dense_gemm_kernel<true>) : reinterpret_cast<const void*>(&gd::dense_gemm_kernel<false>);
  if (const int rc = raise_lds_limit(raised[dtype == HQQ_BF16 ? 1 : 0], kern, lds_bytes, "hqq_hip_gemm_dense")) return rc;
  if (dtype == HQQ_BF16) hipLaunchKernelGGL(gd::dense_gemm_kernel<true>, dim3(static_cast<unsigned>(blocks)), dim3(gd::NT), lds_bytes, as_stream(stream), a);
  else hipLaunchKernelGGL(gd::dense_gemm_kernel<false>, dim3(static_cast<unsigned>(blocks)), dim3(gd::NT), lds_bytes, as_stream(stream), a);
  return check_launch("hqq_hip_gemm_dense");
}
