// skinny.hip — fused unpack -> dequantize -> skinny GEMM on the matrix cores for decode with a batch (5 <= M <= 64 activation
// rows, fp16, 4-/2-bit, group_size 64): the weights are streamed from HBM exactly once, whatever M is.  gfx950.
//
// Reference chain replaced (axis=1): BitPack.unpack_* -> (W_r - zero) * scale -> torch.matmul(x, W.t()) (+ bias)
//   hqq/core/bitpack.py:31-64, hqq/core/quantize.py:183-199, :880-898; patching.py:82-86.
// The weights it multiplies are bit-identical to hqq_hip_dequantize / Quantizer.dequantize (two fp16 roundings); only the
// fp32 summation order differs from a BLAS GEMM, and it is fixed (no atomics), so results are reproducible and a row of y does
// not depend on the batch it was computed in.
//
// Why not the row-per-wave kernel of gemv.hip: there every activation row costs its own MFMAs per weight fragment and the
// matrix pipe passes the dequantisation time around M = 4.  Why not dequantise + library GEMM: that writes and re-reads the
// fp16 matrix (5x the packed bytes) — 28 us for a 4096 x 4096 layer at M = 32 where the packed weights stream in under 3 us.
//
// Data layout, consumed as the reference stores it (no repacking):
//   Wq     [N/per, K] bytes; byte (p, k) holds W_q[p + s*N/per, k] for slab s at bit 8 - nbits*(s+1)
//   scale  [N*G], zero [N*G] fp16, G = K/64; output row n uses [n*G, (n+1)*G)
//
// Work decomposition (the WIDE tile; the narrow one — see SK_NARROW below — has 32-row panels: two row groups, four waves each taking one
// block of a chunk).  A *panel* is 64 packed rows (-> 64*per output rows): one workgroup of eight waves — four row groups of 16
// packed rows, two waves per row group, each dequantising two of a chunk's four 64-k blocks (their partial tiles meet in LDS
// once, at the end; one wave per row group left the SIMDs half idle).  K is walked in chunks of 256: lane (r = lane & 15,
// c = lane >> 4) loads the 16 packed bytes of row r at k = 256*chunk + 64*j + 16*c for its blocks j
// (global_load_dwordx4, non-temporal), two chunks ahead of their use, ping-pong between two register sets; each block is one
// quantisation group, whose (zero, scale) the workgroup copied to LDS for its whole K range before the loop (requested before
// anything else, four lanes per 128-byte line of the two tensors: loads return in order, and as one-lane-per-line gathers behind
// the weights they were the last thing to arrive, 7 us into the launch).
//   x      every wave of the workgroup needs the same activations, so the chunk [M, 256] is staged ONCE per workgroup in LDS
//          (double-buffered; global -> registers one chunk ahead, written behind the MFMAs, one barrier per chunk), already
//          in MFMA B-fragment order: fragment (m-tile, block j, half h) is 64 lanes x 16 bytes, lane (m & 15) + 16 c holding
//          the k-octet 64 j + 16 c + 8 h of activation row m, permuted to the k order the nibble extraction produces.
//          The previous tile kernel (gemv_mfma.hip) re-read x from L2 in every wave: twice the weight bytes through the CU's
//          address path, which held it at 1.0-1.8 TB/s.
//   split-K  small layers do not have 256 panels: the grid is (8, panels / 8, KS), workgroup (panel, ks) walks the ks-th share of
//          the chunks (<= 16) and parks its fp32 partial tile in a scratch buffer; the last split of a row group to arrive
//          (atomic ticket, nobody waits) adds the KS tiles in split order, rounds, adds the bias and stores: fixed order, no
//          second launch.  KS depends on the shapes of the launch only, never on M.
//          KS = 1 stores directly.  blockIdx.x is the XCD (workgroup b runs on XCD b % 8 — observed, a speed assumption only):
//          the KS workgroups of a panel share the lines of zero / scale and the x chunks through one L2.
// Round-1 status (MI355X, int4, graph replay over > 256 MiB of layers): 4096 x 4096: 11.6 us at M = 8, 15 at M = 32, 21 at M = 64
// (dequantise + hipBLASLt: 28-30); 11008 x 4096: 20.6 / 29 / 45 (43-46).  The first 5 us of a launch go into the prologue: the
// CU's L1 issues the misses of 16 rows x 64 B per wave instruction slowly, and x / group constants queue behind them.
// Several layers that read the same x (q/k/v, gate/up) form one launch: their panels are concatenated.
#include <type_traits>

#include "hqq_common.h"
#include "w3s.h"
#include <stdlib.h>

// This file is compiled twice (Makefile): the WIDE tile — 64-packed-row panels, four row groups x two block halves — and, with -DSK_NARROW,
// the NARROW one — 32-row panels, two row groups x four blocks — for 4- / 2-bit launches of at most 2048 packed rows (one 4096-row int4 layer: o, down),
// where the wide tile has 32 panels and must cut K eight ways to fill the chip.  Measured (profiles/r03_skinny_tiles_ks.txt, us per launch,
// wide -> narrow with four K splits): o at 8 / 32 rows 10.3 -> 8.8 / 12.8 -> 10.3; down 13.8 -> 13.1 / 17.4 -> 16.4; the larger 4-bit
// launches (q|k|v, gate|up) are faster on the wide tile (17.0 vs 17.6, 25.1 vs 30.7) and stay there.  2-bit launches (four slabs per byte:
// twice the rebuild work per panel) are faster on the narrow tile at every 7B shape (o 17.7 -> 13.6, q|k|v 23.0 -> 16.3, gate|up 25.0 ->
// 22.0, down 25.2 -> 16.9 at 32 rows): up to 8192 packed rows.  The choice depends on the shapes of the launch only, never on M, so a
// row's bits do not depend on the batch it is computed in.
#ifdef SK_NARROW
#define SK_NS sk_narrow
#ifndef SK_ROW_GROUPS
#define SK_ROW_GROUPS 2
#endif
#ifndef SK_BLOCK_SPLIT
#define SK_BLOCK_SPLIT 4
#endif
#else
#define SK_NS sk_wide
#endif
namespace hqq {
namespace SK_NS {

constexpr int SK_MAXL = HQQ_GEMV_MAX_GROUP;
#ifndef SK_BLOCK_SPLIT
#define SK_BLOCK_SPLIT 2
#endif
#ifndef SK_ROW_GROUPS
#define SK_ROW_GROUPS 4
#endif
constexpr int SK_RG = SK_ROW_GROUPS;                        // row groups of 16 packed rows per panel
constexpr int SK_SPLIT = SK_BLOCK_SPLIT;          // waves sharing a row group; each takes SK_BLK / SK_SPLIT blocks of every chunk
constexpr int SK_WAVES = SK_RG * SK_SPLIT;
constexpr int SK_ROWS = 16 * SK_RG;               // packed rows per panel
constexpr int SK_KC = 256;               // k per chunk
constexpr int SK_BLK = SK_KC / 64;       // 64-k blocks (= groups) per chunk
constexpr int SK_T = SK_WAVES * 64;
constexpr int SK_BPW = SK_BLK / SK_SPLIT;     // blocks per wave and chunk
#ifndef SK_X_EARLY
#define SK_X_EARLY 1
#endif
#ifndef SK_RING_DEPTH
#define SK_RING_DEPTH 2
#endif
constexpr int SK_RING = SK_RING_DEPTH;        // units in flight per wave (even: the two x buffers alternate with the halves); 4 measured no better
constexpr int SK_MAX_CPS = 16;          // chunks per K split (the group constants of a split are fetched in one batch)
// nbits = 3 is the 3-bit STREAM layout (w3s.h): two row slabs per packed row like the 4-bit container, 12 bytes per lane and block
constexpr int sk_per(int nbits) { return nbits == 3 ? 2 : 8 / nbits; }

typedef _Float16 sk_h8_t __attribute__((ext_vector_type(8)));

struct SkArgs {
  const uint8_t* Wq[SK_MAXL];
  const half_t* scale[SK_MAXL];
  const half_t* zero[SK_MAXL];
  const half_t* bias[SK_MAXL];
  half_t* y[SK_MAXL];
  int N[SK_MAXL];           // out_features
  int panel_end[SK_MAXL];   // end (exclusive) of layer i's panels in the concatenated panel space (unused entries repeat the last)
  int n_off[SK_MAXL];       // first column of layer i in the concatenated output space of the scratch buffer
  const half_t* x;
  float* part;              // [KS][panel][row group][PER][MT][4][64 lanes] fp32 partial tiles in accumulator order (KS > 1 only)
  int* cnt;                 // [panel][row group] arrival counters of the K splits, zero between launches (KS > 1 only)
  int M, K, G, total_panels, KS, cps, n_total;
};

struct SkLayer {   // workgroup-uniform -> SGPRs
  const uint8_t* Wq;
  const half_t* scale;
  const half_t* zero;
  const half_t* bias;
  half_t* y;
  int N, panel0, n_off;
};

__device__ __forceinline__ SkLayer sk_select(const SkArgs& a, int panel) {
  SkLayer c{a.Wq[0], a.scale[0], a.zero[0], a.bias[0], a.y[0], a.N[0], 0, a.n_off[0]};
#pragma unroll
  for (int i = 1; i < SK_MAXL; ++i) {
    const bool in = panel >= a.panel_end[i - 1];
    c.Wq = in ? a.Wq[i] : c.Wq;
    c.scale = in ? a.scale[i] : c.scale;
    c.zero = in ? a.zero[i] : c.zero;
    c.bias = in ? a.bias[i] : c.bias;
    c.y = in ? a.y[i] : c.y;
    c.N = in ? a.N[i] : c.N;
    c.panel0 = in ? a.panel_end[i - 1] : c.panel0;
    c.n_off = in ? a.n_off[i] : c.n_off;
  }
  return c;
}

__device__ __forceinline__ half2_t sk_h2(uint32_t u) { return __builtin_bit_cast(half2_t, u); }
__device__ __forceinline__ uint32_t sk_u32(half2_t h) { return __builtin_bit_cast(uint32_t, h); }

// exact integer levels of slab S for the byte pairs (b0,b2) [word] / (b1,b3) [word >> 8] of a packed dword as fp16:
// (word & mask) | 0x6400 is the fp16 number 1024 + q * 2^sh; one packed fma removes the bias exactly.
template <int NBITS, int S>
__device__ __forceinline__ half2_t sk_levels(uint32_t word_or_shifted, uint32_t magic) {
  constexpr int per = 8 / NBITS;
  constexpr int sh = NBITS * (per - 1 - S);
  constexpr uint32_t m1 = ((1u << NBITS) - 1u) << sh;
  constexpr uint32_t m = m1 | (m1 << 16);
  uint32_t b;
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(b) : "v"(word_or_shifted), "s"(m), "v"(magic));
  constexpr float inv = 1.0f / static_cast<float>(1 << sh);
  const half2_t k1 = {static_cast<half_t>(inv), static_cast<half_t>(inv)};
  const half2_t k2 = {static_cast<half_t>(-1024.0f * inv), static_cast<half_t>(-1024.0f * inv)};
  return __builtin_elementwise_fma(sk_h2(b), k1, k2);
}

__device__ __forceinline__ u32x4 sk_permute_x8(u32x4 v) {   // (k0..k7) -> (k0,k2,k1,k3,k4,k6,k5,k7)
  u32x4 r;
  r.x = (v.x & 0xFFFFu) | (v.y << 16);
  r.y = (v.x >> 16) | (v.y & 0xFFFF0000u);
  r.z = (v.z & 0xFFFFu) | (v.w << 16);
  r.w = (v.z >> 16) | (v.w & 0xFFFF0000u);
  return r;
}

// one 64-k block of one slab: dequantise the lane's 16 weights exactly as Quantizer.dequantize does (two fp16 roundings) ONCE,
// then contract them with every m-tile's activation octets on the matrix core
template <int NBITS, int MT, int S, int PER, bool SUB = false>
struct SkSlab {
  static __device__ __forceinline__ void run(const u32x4& w, const uint32_t (&zs)[PER], const sk_h8_t (&b0)[MT], const sk_h8_t (&b1)[MT],
                                             f32x4 (&acc)[PER][MT], uint32_t magic) {
    const half2_t pr = sk_h2(zs[S]);
    const half2_t zz = {pr.x, pr.x}, ss = {pr.y, pr.y};
    half2_t q[8];
    uint32_t o[8];
    if constexpr (SUB) {   // three-op rebuild (decode_common.h): the table holds (z 2^-J, s 2^J); the masked field is the subnormal q 2^(sh-24)
      constexpr int sh = NBITS * (PER - 1 - S);
      constexpr uint32_t m1 = ((NBITS == 8) ? 0xFFu : ((1u << NBITS) - 1u)) << sh;
      constexpr uint32_t m = m1 | (m1 << 16);
      const half2_t lift = {static_cast<half_t>(32768.0f), static_cast<half_t>(32768.0f)};
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        q[2 * d] = sk_h2(w[d] & m);              // bytes (4d+0, 4d+2)
        q[2 * d + 1] = sk_h2((w[d] >> 8) & m);   // bytes (4d+1, 4d+3)
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) q[i] = __builtin_elementwise_fma(q[i], lift, -zz);   // rounding 1
    } else {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      q[2 * d] = sk_levels<NBITS, S>(w[d], magic);            // bytes (4d+0, 4d+2)
      q[2 * d + 1] = sk_levels<NBITS, S>(w[d] >> 8, magic);   // bytes (4d+1, 4d+3)
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) q[i] = q[i] - zz;                                  // rounding 1
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = sk_u32(q[i] * ss);                          // rounding 2
    const sk_h8_t a0 = __builtin_bit_cast(sk_h8_t, u32x4{o[0], o[1], o[2], o[3]});   // k = 16c + 0..7 (permuted inside the octet)
    const sk_h8_t a1 = __builtin_bit_cast(sk_h8_t, u32x4{o[4], o[5], o[6], o[7]});   // k = 16c + 8..15
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      acc[S][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0[t], acc[S][t], 0, 0, 0);
      acc[S][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1[t], acc[S][t], 0, 0, 0);
    }
    if constexpr (S + 1 < PER) SkSlab<NBITS, MT, S + 1, PER, SUB>::run(w, zs, b0, b1, acc, magic);
  }
};

// bf16 compute dtype: the reference's two roundings are to bf16 (quantize.py:198 on bf16 tensors).  gfx950 has no packed bf16
// arithmetic, so the weight goes through fp32: v_cvt_f32_ubyteN lifts the masked byte F q, one fma forms q - z, v_cvt_pk_bf16_f32 rounds it
// (RNE), v_dot2_f32_bf16 against (s, 0) / (0, s) forms the exact product with s, a second v_cvt_pk rounds again (as gemv.hip).
typedef __bf16 sk_bf2_t __attribute__((ext_vector_type(2)));
typedef __bf16 sk_bf8_t __attribute__((ext_vector_type(8)));
typedef float sk_f2_t __attribute__((ext_vector_type(2)));
template <int NBITS, int S>
__device__ __forceinline__ uint32_t sk_masked(uint32_t word) {   // the four bytes of a word reduced to slab S's field (F q each)
  constexpr int per = 8 / NBITS;
  constexpr int sh = NBITS * (per - 1 - S);
  constexpr uint32_t m1 = ((1u << NBITS) - 1u) << sh;
  if constexpr (NBITS == 8) return word;
  return word & (m1 * 0x01010101u);
}
template <int B>
__device__ __forceinline__ float sk_ubyte(uint32_t v) { return static_cast<float>((v >> (8 * B)) & 0xFFu); }   // v_cvt_f32_ubyteB
template <int NBITS, int MT, int S, int PER>
struct SkSlabBF16 {
  static __device__ __forceinline__ void run(const u32x4& w, const uint32_t (&zs)[PER], const sk_bf8_t (&b0)[MT], const sk_bf8_t (&b1)[MT],
                                             f32x4 (&acc)[PER][MT], uint32_t magic) {
    constexpr int sh = NBITS * (PER - 1 - S);
    constexpr float inv = 1.0f / static_cast<float>(1 << sh);
    const float zf = __uint_as_float(zs[S] << 16);
    const sk_bf2_t s_lo = __builtin_bit_cast(sk_bf2_t, zs[S] >> 16);          // (s, 0)
    const sk_bf2_t s_hi = __builtin_bit_cast(sk_bf2_t, zs[S] & 0xFFFF0000u);  // (0, s)
    uint32_t o[8];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const uint32_t fq = sk_masked<NBITS, S>(w[d]);
      // fma(F q, 1 / F, -z) is q - z with ONE fp32 rounding (none unless z is below 2^-15): a bias folded into the addend
      // (-(1024 / F) - z) would itself round when z is small and cost an ulp after rounding 1
      const sk_f2_t dq[2] = {{__builtin_fmaf(sk_ubyte<0>(fq), inv, -zf), __builtin_fmaf(sk_ubyte<2>(fq), inv, -zf)},    // bytes (4d+0, 4d+2)
                             {__builtin_fmaf(sk_ubyte<1>(fq), inv, -zf), __builtin_fmaf(sk_ubyte<3>(fq), inv, -zf)}};   // bytes (4d+1, 4d+3)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const sk_bf2_t dr = __builtin_convertvector(dq[h], sk_bf2_t);                 // rounding 1
        const sk_f2_t pw = {__builtin_amdgcn_fdot2_f32_bf16(dr, s_lo, 0.f, false), __builtin_amdgcn_fdot2_f32_bf16(dr, s_hi, 0.f, false)};
        o[2 * d + h] = __builtin_bit_cast(uint32_t, __builtin_convertvector(pw, sk_bf2_t));   // rounding 2
      }
    }
    const sk_bf8_t a0 = __builtin_bit_cast(sk_bf8_t, u32x4{o[0], o[1], o[2], o[3]});
    const sk_bf8_t a1 = __builtin_bit_cast(sk_bf8_t, u32x4{o[4], o[5], o[6], o[7]});
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      acc[S][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0[t], acc[S][t], 0, 0, 0);
      acc[S][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1[t], acc[S][t], 0, 0, 0);
    }
    if constexpr (S + 1 < PER) SkSlabBF16<NBITS, MT, S + 1, PER>::run(w, zs, b0, b1, acc, magic);
  }
};

// 3-bit stream layout (w3s.h): one 64-k block of both slabs — the lane's 12 bytes rebuilt exactly (fp16: three- or four-op form; bf16
// through fp32) into natural-order A fragments, contracted with every m-tile's activation octets
template <int MT, bool BF16, bool SUB>
struct SkSlabW3s {
  template <class FRAG>
  static __device__ __forceinline__ void run(const u32x4& w, const uint32_t (&zs)[2], const FRAG (&b0)[MT], const FRAG (&b1)[MT], f32x4 (&acc)[2][MT], uint32_t magic) {
    if constexpr (BF16) {
      w3s_bf8_t a0[2], a1[2];
      w3s_rebuild_bf16(w.x, w.y, w.z, zs, a0, a1);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[s], b0[t], acc[s][t], 0, 0, 0);
          acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[s], b1[t], acc[s][t], 0, 0, 0);
        }
    } else {
      h8_t a0[2], a1[2];
      w3s_rebuild_f16<SUB>(w.x, w.y, w.z, zs, magic, a0, a1);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[s], b0[t], acc[s][t], 0, 0, 0);
          acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[s], b1[t], acc[s][t], 0, 0, 0);
        }
    }
  }
};

struct SkUnit {   // one wave's share of one chunk: SK_BPW KiB of packed weights
  u32x4 w[SK_BPW];
};

template <int NBITS, int MT, bool BF16, bool SUB = false>
__global__ __launch_bounds__(SK_T) void skinny_f16_kernel(const SkArgs a) {
  constexpr bool W3 = NBITS == 3;
  constexpr int PER = sk_per(NBITS);
  constexpr int LB = W3 ? 12 : 16;                  // bytes per lane and block (16 k of PER rows)
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  u32x4* xs = reinterpret_cast<u32x4*>(smem);   // [2 buffers][MT][SK_BLK][2 halves][64 lanes] x 16 B
  constexpr int XS_BUF = MT * SK_BLK * 2 * 64;  // u32x4 per buffer
  uint32_t* mz = reinterpret_cast<uint32_t*>(smem + static_cast<size_t>(2) * XS_BUF * sizeof(u32x4));   // [64 rows][PER][mstride] (zero | scale << 16)
  constexpr int XP = (MT * 8 + SK_WAVES - 1) / SK_WAVES;   // 16-byte pieces of x per thread and chunk (8 MT fragments of 64 lanes / waves)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, c = lane >> 4;
  const int rg = wave % SK_RG, hf = wave / SK_RG;   // row group, and which blocks of a chunk this wave dequantises
  const int K = a.K, G = a.G, M = a.M;
  // XCD-aware placement: workgroup b runs on XCD b % 8 (observed; a speed assumption only), and the KS workgroups of one panel all
  // read the same 128-byte lines of zero / scale (a row's 64 groups).  Spread over eight L2s each of those lines was fetched up to
  // KS times and the group constants cost as much HBM traffic as the weights; on one XCD they are fetched once.
  const int panel = blockIdx.y * 8 + blockIdx.x, ks = blockIdx.z;   // grid (8, panels / 8, KS): blockIdx.x is the XCD
  if (panel >= a.total_panels) return;
  const int nchunks = K / SK_KC;
  const int cps = a.cps;                                       // chunks per split (<= SK_MAX_CPS)
  const int c0 = ks * cps, c1 = (c0 + cps < nchunks) ? c0 + cps : nchunks;
  const int mstride = cps * SK_BLK + 1;   // dwords per (row, slab) of group constants in LDS; odd: rows fall on different banks
  const SkLayer ly = sk_select(a, panel);
  const int rows_per_slab = ly.N / PER;
  int p = (panel - ly.panel0) * SK_ROWS + rg * 16 + r;         // packed row inside the layer
  p = p < rows_per_slab ? p : rows_per_slab - 1;               // ragged last panel: duplicate the last row (masked at the store)
  const uint8_t* wrow = ly.Wq + static_cast<int64_t>(p) * (K / 16 * LB) + c * LB;
  // (3-bit: 12-byte loads through a buffer descriptor over the layer — offsets below 4 GiB, checked on the host)
  const __amdgpu_buffer_rsrc_t wrs = buffer_rsrc(ly.Wq);
  const uint32_t wvoff = static_cast<uint32_t>(p) * static_cast<uint32_t>(K / 16 * LB) + static_cast<uint32_t>(c * LB);
  // Whole-line loads (SK_LINE_LOADS, two blocks per wave): a wave instruction that reads 16 rows x 64 B touches HALF of sixteen 128-byte
  // lines, and the other halves come with the next instruction — measured on pure loads (tools/floor_probe.hip, profiles/r03_floor_probe5.txt)
  // that pattern streams 26 % slower than contiguous KiBs, 8 rows x 128 B only 4.5 % slower.  So load L0 = rows 0-7 and L1 = rows 8-15 of
  // the row group, both blocks each — lane (r8 = lane & 7, h = (lane >> 3) & 1, c): 16 bytes of block h — and put the MFMA layout (lane =
  // row + 16 c, one block per register set) back with one DPP move per dword at consume time: block 0 = L0 in lanes 0-7 | L1 rotated by 8
  // in lanes 8-15 of every 16-lane row, block 1 the other way round.
  // Measured in THIS kernel (tools/r3_lab_skinny.sh, bit-identical results): no difference (10.4 / 13.1 / 14.4 / 17.7 us either way) — the
  // skinny GEMM is bound by its per-chunk barrier and the bytes it keeps in flight, not by the request pattern.  Lab switch, off.
#ifndef SK_LINE_LOADS
#define SK_LINE_LOADS 0
#endif
  constexpr bool LINES = SK_LINE_LOADS && SK_BPW == 2;
  const uint8_t* wline[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    int pl = (panel - ly.panel0) * SK_ROWS + rg * 16 + 8 * u + (lane & 7);
    pl = pl < rows_per_slab ? pl : rows_per_slab - 1;
    wline[u] = ly.Wq + static_cast<int64_t>(pl) * K + ((lane >> 3) & 1) * 64 + c * 16;
  }

  // x: the 256 threads fill the chunk's 8 MT fragments in LDS order — piece q = tid + 256 i is lane (q & 63) of fragment
  // f = q >> 6 = (m-tile t, block j, half h), i.e. the k-octet 64 j + 16 c + 8 h of activation row 16 t + r.  Rows >= M repeat
  // row 0 (finite values; their columns of the result are never stored) — zeroing them would put the load under a branch:
  // a wave writes 1 KiB of consecutive LDS (no bank conflicts; a row-major assignment of the pieces put 32 lanes on one bank)
  u32x4 xr[2][XP];           // two register sets: a chunk's x is requested a whole half-iteration before the weights requested in
  uint32_t xkeep[2] = {~0u, ~0u};   // that half, so that its arrival (loads return in order) never waits behind an HBM round trip
  auto xload = [&](int chunk, int set) {
    // past the range (odd number of chunks): zeros, so that the ring's second unit can be consumed unconditionally (it then holds
    // finite dummy weights and adds exactly 0) — a consume under a branch gets its first instructions hoisted above the branch,
    // in front of the next request, and the wave ends up with one unit in flight instead of two
    xkeep[set] = chunk < c1 ? ~0u : 0u;   // wave-uniform; applied when the registers are written to LDS (not here: an instruction on
    chunk = chunk < c1 ? chunk : c1 - 1;  // the loaded value would wait for the load on the spot)
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      int f = wave + SK_WAVES * i;
      f = f < 8 * MT ? f : 8 * MT - 1;   // (8 MT not a multiple of the wave count: the spare pieces repeat the last fragment)
      const int t = f >> 3, j = (f & 7) >> 1, h = f & 1;
      const int m = 16 * t + r;
      xr[set][i] = *reinterpret_cast<const u32x4*>(a.x + static_cast<int64_t>(m < M ? m : 0) * K + chunk * SK_KC + 64 * j + 16 * c + 8 * h);
    }
  };
  auto xstore = [&](int buf, int set) {
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      int f = wave + SK_WAVES * i;
      f = f < 8 * MT ? f : 8 * MT - 1;
      const u32x4 v = W3 ? xr[set][i] : sk_permute_x8(xr[set][i]);   // (3-bit stream layout: natural k order)
      xs[buf * XS_BUF + f * 64 + lane] = u32x4{v.x & xkeep[set], v.y & xkeep[set], v.z & xkeep[set], v.w & xkeep[set]};
    }
  };
  // Every issue() emits exactly SK_BLK weight loads, in range or not, so that the waits the compiler derives
  // are exact vmcnt counts and never vmcnt(0).
  auto issue = [&](SkUnit& un, int chunk) {
    const bool live = chunk < c1;          // past the range: every lane reads the first bytes of x instead (one cached line,
    chunk = live ? chunk : c1 - 1;         // no HBM traffic); the unit is never consumed
#pragma unroll
    for (int jl = 0; jl < SK_BPW; ++jl) {
      const int j = hf * SK_BPW + jl;
      if constexpr (W3) {
        const u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(wrs, live ? wvoff : 0u, live ? (chunk * SK_BLK + j) * (4 * LB) : 0, 2 /* nt */);
        un.w[jl] = u32x4{v.x, v.y, v.z, 0u};
      } else {
      const uint8_t* at = LINES ? wline[jl] + static_cast<int64_t>(chunk) * SK_KC + hf * 128 : wrow + static_cast<int64_t>(chunk) * SK_KC + j * 64;
      const u32x4* src = live ? reinterpret_cast<const u32x4*>(at) : reinterpret_cast<const u32x4*>(a.x);
      un.w[jl] = __builtin_nontemporal_load(src);
      }
    }
  };

  uint32_t magic;
  asm volatile("v_mov_b32 %0, 0x64006400" : "=v"(magic));   // opaque to the optimiser: stays in a VGPR
  f32x4 acc[PER][MT];
#pragma unroll
  for (int s = 0; s < PER; ++s)
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[s][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto consume = [&](const SkUnit& raw, int buf, int chunk) {
    SkUnit cur = raw;
    if constexpr (LINES) {
#pragma unroll
      for (int d = 0; d < 4; ++d) {   // row_ror:8 (0x128), banks 2-3 = lanes 8-15 of a row, banks 0-1 = lanes 0-7
        cur.w[0][d] = __builtin_amdgcn_update_dpp(raw.w[0][d], raw.w[1][d], 0x128, 0xF, 0xC, false);
        cur.w[1][d] = __builtin_amdgcn_update_dpp(raw.w[1][d], raw.w[0][d], 0x128, 0xF, 0x3, false);
      }
    }
#pragma unroll
    for (int jl = 0; jl < SK_BPW; ++jl) {
      const int j = hf * SK_BPW + jl;
      uint32_t zs[PER];
#pragma unroll
      for (int s = 0; s < PER; ++s) zs[s] = mz[((rg * 16 + r) * PER + s) * mstride + (chunk - c0) * SK_BLK + j];   // one group per block
      using frag_t = std::conditional_t<BF16, sk_bf8_t, sk_h8_t>;
      frag_t b0[MT], b1[MT];
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        b0[t] = __builtin_bit_cast(frag_t, xs[buf * XS_BUF + ((t * SK_BLK + j) * 2 + 0) * 64 + lane]);
        b1[t] = __builtin_bit_cast(frag_t, xs[buf * XS_BUF + ((t * SK_BLK + j) * 2 + 1) * 64 + lane]);
      }
      if constexpr (W3) SkSlabW3s<MT, BF16, SUB>::run(cur.w[jl], zs, b0, b1, acc, magic);
      else if constexpr (BF16) SkSlabBF16<NBITS, MT, 0, PER>::run(cur.w[jl], zs, b0, b1, acc, magic);
      else
      SkSlab<NBITS, MT, 0, PER, SUB>::run(cur.w[jl], zs, b0, b1, acc, magic);
    }
  };

  // ---- prologue: x of the first chunk, both units of the ring, then the group constants of the workgroup's whole K range:
  //      thread (row = tid >> 2, q = tid & 3) copies [zero | scale] x [slab] (PER = 2: one each; PER = 4: two passes) of its row,
  //      8 bytes (one chunk's four groups) per load — every line of the two tensors is touched once per workgroup, not once per
  //      chunk as with per-chunk 2-byte loads, which cost more address-path time than the weights themselves ----
  SkUnit un[SK_RING];   // the wave's ring of units in flight (indexed by constants only: registers)
  // group constants FIRST (loads return in order; behind 8 KiB of weights per wave they were the last thing to arrive), and as
  // coalesced as the layout allows: four lanes fetch four consecutive chunks (32 bytes) of one (row, slab, zero | scale) line, so a
  // wave instruction touches 16 lines and every line of the two tensors is requested once per workgroup.  (One lane per line, 8
  // bytes per instruction, was 64 line requests per instruction — several times the L1 -> L2 requests of the weights.)
  constexpr int NRC = SK_ROWS * 2 * PER;        // (row, slab, zero | scale) lines per panel
  constexpr int NPASS = NRC / (SK_T / 4);       // 2 PER
  constexpr int NROUND = SK_MAX_CPS / 4;
  u32x2 mv[NROUND][NPASS];
  // (a round covers four chunks: the rounds past the split's chunks are skipped — a workgroup-uniform branch; what counts for the waits below is the
  //  number of loads issued AFTER a round, which is the same on both paths.  With 2-8 chunks per split they were 4-12 of the 16 + 6 vector-memory
  //  instructions a wave issues before its first consume: -1 % per step at 32 rows on two boxes.  2-bit layers — four slabs, twice the passes — lost 0.8 % with it
  //  on both and keep every round)
#pragma unroll
  for (int rd = 0; rd < NROUND; ++rd)
    if (NBITS == 2 || rd * 4 < c1 - c0)
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
      const int id = pass * (SK_T / 4) + (tid >> 2), row = id / (2 * PER), q = id % (2 * PER), s = q >> 1, hi = q & 1;
      const int cc = rd * 4 + (tid & 3);
      int pm = (panel - ly.panel0) * SK_ROWS + row;
      pm = pm < rows_per_slab ? pm : rows_per_slab - 1;
      const half_t* src = (hi ? ly.scale : ly.zero) + static_cast<int64_t>(pm + s * rows_per_slab) * G + (c0 + (cc < c1 - c0 ? cc : 0)) * SK_BLK;
      mv[rd][pass] = *reinterpret_cast<const u32x2*>(src);
    }
  xload(c0, 0);
  issue(un[0], c0);
  xload(c0 + 1, 1);
#pragma unroll
  for (int k = 1; k < SK_RING; ++k) issue(un[k], c0 + k);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int rd = 0; rd < NROUND; ++rd)
    if (NBITS == 2 || rd * 4 < c1 - c0)
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
      const int id = pass * (SK_T / 4) + (tid >> 2), row = id / (2 * PER), q = id % (2 * PER), s = q >> 1, hi = q & 1;
      const int cc = rd * 4 + (tid & 3);
      if (cc < c1 - c0) {
        uint16_t* dst = reinterpret_cast<uint16_t*>(mz + (row * PER + s) * mstride + cc * SK_BLK) + hi;
        u32x2 v = mv[rd][pass];
        if constexpr (SUB && !W3) {   // (z, s) -> (z 2^-J, s 2^J), J = 9 - shift of the slab: exact for every group (hqq_hip_meta_check); 3-bit: per field offset, by the consumer
          const int J = 9 - NBITS * (PER - 1 - s);
          const uint16_t fb = static_cast<uint16_t>((hi ? 15 + J : 15 - J) << 10);
          const half2_t f = {__builtin_bit_cast(half_t, fb), __builtin_bit_cast(half_t, fb)};
          v.x = sk_u32(sk_h2(v.x) * f);
          v.y = sk_u32(sk_h2(v.y) * f);
        }
        dst[0] = static_cast<uint16_t>(v.x);
        dst[2] = static_cast<uint16_t>(v.x >> 16);
        dst[4] = static_cast<uint16_t>(v.y);
        dst[6] = static_cast<uint16_t>(v.y >> 16);
      }
    }
  xstore(0, 0);
  __syncthreads();

  // ---- SK_RING chunks per iteration (register ring, no copies).  Per half: consume a unit, request x two chunks and then the
  //      weights SK_RING chunks ahead (x first: in-order return must not park it behind an HBM round trip — with x requested after the previous
  //      half's weights every half-iteration lasted one memory latency, whatever it computed), write the x requested one half
  //      earlier to LDS, one barrier. ----
  // (sched_barrier: the machine scheduler knows nothing about what a wait costs — left alone it lifts the first instructions of
  //  the NEXT consume, which read the unit requested last, to the front of the block, and the wait they drag along serialises the ring)
  for (int i = c0; i < c1; i += SK_RING) {
#pragma unroll
    for (int k = 0; k < SK_RING; ++k) {
      // chunks past the range (the ring is deeper than the remainder) meet zero x: finite dummy weights, exactly 0 added
#if SK_X_EARLY   // x of chunk i + k + 2 requested BEFORE this half's consume: almost two half-iterations of lead instead of one
      xload(i + k + 2, k & 1);
      __builtin_amdgcn_sched_barrier(0);
#endif
      consume(un[k], k & 1, i + k < c1 ? i + k : c1 - 1);
      __builtin_amdgcn_sched_barrier(0);
#if !SK_X_EARLY
      xload(i + k + 2, k & 1);
#endif
      issue(un[k], i + k + SK_RING);
      __builtin_amdgcn_sched_barrier(0);
      xstore((k + 1) & 1, (k + 1) & 1);   // x of chunk i + k + 1, requested one half-iteration ago
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- the SK_SPLIT waves of a row group each hold a partial tile: the upper ones hand theirs over through LDS (the x buffers
  //      are free now), wave hf = 0 adds them in a fixed order and stores ----
  if constexpr (SK_SPLIT > 1) {
    f32x4* red = reinterpret_cast<f32x4*>(smem);   // [SK_SPLIT - 1][SK_RG][PER][MT][64 lanes]
    if (hf > 0) {
#pragma unroll
      for (int s = 0; s < PER; ++s)
#pragma unroll
        for (int t = 0; t < MT; ++t) red[((((hf - 1) * SK_RG + rg) * PER + s) * MT + t) * 64 + lane] = acc[s][t];
    }
    __syncthreads();
    if (hf > 0) return;
#pragma unroll
    for (int h = 1; h < SK_SPLIT; ++h)
#pragma unroll
      for (int s = 0; s < PER; ++s)
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          const f32x4 o = red[((((h - 1) * SK_RG + rg) * PER + s) * MT + t) * 64 + lane];
          acc[s][t][0] += o[0]; acc[s][t][1] += o[1]; acc[s][t][2] += o[2]; acc[s][t][3] += o[3];
        }
  }
  // ---- K splits: every split parks its tile in the scratch (accumulator order: a wave stores 256 consecutive bytes per
  //      instruction), the LAST split of the row group to arrive — a ticket from an atomic counter, no waiting — adds all KS tiles
  //      in split order and goes on to the store below.  Fixed order, so the bits do not depend on which split finished last;
  //      no second launch (the finishing kernel this replaces cost one graph-node gap + ~1 us per call). ----
  if (a.KS > 1) {
    constexpr int TILE = PER * MT * 4 * 64;   // floats per (split, panel, row group)
    const int64_t slot = static_cast<int64_t>(panel) * SK_RG + rg;
    const int64_t kstride = static_cast<int64_t>(a.total_panels) * SK_RG * TILE;
    float* mine = a.part + ks * kstride + slot * TILE + lane;
#pragma unroll
    for (int s = 0; s < PER; ++s)
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) __hip_atomic_store(mine + ((s * MT + t) * 4 + i) * 64, acc[s][t][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // Device-scope (sc1) stores and loads for the tiles, a wait for the stores' acknowledgement before the ticket: the tile is
    // at the device's coherence point before the counter moves, and the finisher's loads go there too.  (A full __threadfence()
    // on either side writes back / invalidates the whole L2 of the XCD: measured +17 us per launch.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int ticket = 0;
    if (lane == 0) ticket = __hip_atomic_fetch_add(a.cnt + slot, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    if (ticket != a.KS - 1) return;
    const float* all = a.part + slot * TILE + lane;
#pragma unroll
    for (int s = 0; s < PER; ++s)
#pragma unroll
      for (int t = 0; t < MT; ++t) acc[s][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // KB tiles per round trip (these loads go to the device's coherence point, ~0.6 us each way): all of a batch's loads are in
    // flight before the first add.  One tile per trip made the finisher the last wave of the launch by 5 us (KS = 8).
    constexpr int V = PER * MT * 4;
    constexpr int KB = V >= 64 ? 1 : 64 / V;
    for (int k0 = 0; k0 < a.KS; k0 += KB) {
      float tmp[KB][V];
#pragma unroll
      for (int kk = 0; kk < KB; ++kk) {
        const int kc = k0 + kk < a.KS ? k0 + kk : a.KS - 1;   // (past the last split: the last tile again, dropped below)
#pragma unroll
        for (int v = 0; v < V; ++v) tmp[kk][v] = __hip_atomic_load(all + kc * kstride + v * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int kk = 0; kk < KB; ++kk) {
        const bool keep = k0 + kk < a.KS;
#pragma unroll
        for (int s = 0; s < PER; ++s)
#pragma unroll
          for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[s][t][i] += keep ? tmp[kk][(s * MT + t) * 4 + i] : 0.f;
      }
    }
    if (lane == 0) __hip_atomic_store(a.cnt + slot, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch (stream order)
  }
  // ---- D layout: lane (column r = activation row inside the m-tile, rows 4c + i = packed row inside the wave's 16) ----
  const int p_base = (panel - ly.panel0) * SK_ROWS + rg * 16 + c * 4;
#pragma unroll
  for (int s = 0; s < PER; ++s)
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int m = t * 16 + r;
      if (m >= M) continue;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int pp = p_base + i;
        if (pp < rows_per_slab) {
          const int n = pp + s * rows_per_slab;
          if constexpr (BF16) {
            uint16_t o = f32_to_bf16(acc[s][t][i]);
            if (ly.bias) o = f32_to_bf16(bf16_to_f32(o) + bf16_to_f32(reinterpret_cast<const uint16_t*>(ly.bias)[n]));
            reinterpret_cast<uint16_t*>(ly.y)[static_cast<int64_t>(m) * ly.N + n] = o;
          } else {
            half_t o = static_cast<half_t>(acc[s][t][i]);
            if (ly.bias) o = o + ly.bias[n];   // `out += bias` on the rounded matmul result (quantize.py:896-897)
            ly.y[static_cast<int64_t>(m) * ly.N + n] = o;
          }
        }
      }
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------------
static int sk_num_cus() {
  static int n_cus = 0;
  if (n_cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) n_cus = n;
    else n_cus = 256;
  }
  return n_cus;
}

// Workspace of the split-K launches (caller-owned, hqq_hip_gemv_workspace_bytes): [arrival counters | fp32 partial tiles].
// The counters must read zero when a call starts; every call leaves them zero again (the finishing split resets its counter),
// so the caller clears the workspace once, when it allocates it.
constexpr size_t SK_CNT_BYTES = WS_COUNTER_BYTES;   // head of the workspace: one int per (panel, row group), 64 Ki counters

// K splits (shape-dependent only, never M: a row's result must not depend on the batch it is computed in).  Measured on 7B / 70B
// shapes: about one workgroup per CU and >= 8 chunks per workgroup wins — every split pays the prologue (first data ~5 us after
// launch) and writes a partial tile; only layers with few panels are worth cutting finer.  forced: HQQ_OPT_SKINNY_KS (tuning).
static void sk_choose(int total_panels, int nchunks, int forced, int& ks, int& cps) {
  const int cus = sk_num_cus();
  ks = (total_panels * 16 >= cus * 9 && nchunks <= SK_MAX_CPS) ? 1   // >= 0.56 workgroups per CU: one pass, no partials
                                                               : (cus + total_panels - 1) / total_panels;
#ifndef SK_NARROW
  if (total_panels >= 48) { const int lim = nchunks / 8 > 1 ? nchunks / 8 : 1; ks = ks > lim ? lim : ks; }
#else   // narrow tile (measured, profiles/r03_skinny_tiles_ks.txt): as many splits as fill the chip without a second round, >= 4 chunks each
  if (ks > 1) { ks = cus / total_panels; const int lim = nchunks / 4 > 1 ? nchunks / 4 : 1; ks = ks > lim ? lim : ks; }
#endif
  ks = ks > nchunks / 2 ? nchunks / 2 : ks;   // at least two chunks per workgroup
  ks = ks > 16 ? 16 : (ks < 1 ? 1 : ks);
  if (forced >= 1 && forced <= nchunks) ks = forced;
  cps = (nchunks + ks - 1) / ks;
  cps = cps > SK_MAX_CPS ? SK_MAX_CPS : cps;  // (long K: more splits than the occupancy rule asks for)
  ks = (nchunks + cps - 1) / cps;             // drop empty splits
}
static size_t sk_part_bytes(int nbits, int ks, int total_panels, int mt) {
  return ks > 1 ? static_cast<size_t>(ks) * total_panels * SK_ROWS * sk_per(nbits) * 16 * mt * sizeof(float) : 0;
}

template <int NBITS, bool BF16, bool SUB = false>
static int sk_launch(SkArgs& a, uint32_t opts, void* ws, size_t ws_bytes, hipStream_t st) {
  const int mt = (a.M + 15) / 16;
  const int nchunks = a.K / SK_KC;
  int ks, cps;
  sk_choose(a.total_panels, nchunks, static_cast<int>(opts >> 24), ks, cps);
  a.KS = ks;
  a.cps = cps;
  a.part = nullptr;
  if (ks > 1) {
    if (static_cast<size_t>(a.total_panels) * SK_RG * sizeof(int) > SK_CNT_BYTES) { set_error("hqq_hip_gemv: too many row panels for the split-K counters"); return HQQ_ERR_UNSUPPORTED; }
    const size_t need = SK_CNT_BYTES + sk_part_bytes(NBITS, ks, a.total_panels, mt);
    if (!ws || ws_bytes < need) { set_error("hqq_hip_gemv: this launch splits K and needs %zu bytes of workspace (hqq_hip_gemv_workspace_bytes), got %zu", need, ws ? ws_bytes : size_t(0)); return HQQ_ERR_WORKSPACE; }
    if (!aligned16(ws)) { set_error("hqq_hip_gemv: workspace must be 16-byte aligned"); return HQQ_ERR_ALIGN; }
    a.cnt = static_cast<int*>(ws);
    a.part = reinterpret_cast<float*>(static_cast<char*>(ws) + SK_CNT_BYTES);
  }
  size_t lds = static_cast<size_t>(2) * mt * SK_BLK * 2 * 64 * sizeof(u32x4) + static_cast<size_t>(SK_ROWS) * sk_per(NBITS) * (cps * SK_BLK + 1) * sizeof(uint32_t);
  // the partial tiles of a row group's upper waves meet in the same LDS once the loop is done: (SK_SPLIT - 1) tiles per row group and slab
  const size_t red = static_cast<size_t>(SK_SPLIT - 1) * SK_RG * sk_per(NBITS) * mt * 64 * sizeof(f32x4);
  lds = lds > red ? lds : red;
  const dim3 grid(8, static_cast<unsigned>((a.total_panels + 7) / 8), static_cast<unsigned>(ks)), block(SK_T);   // see the kernel
#define HQQ_SK_CASE(MT)                                                                                       \
  case MT: {                                                                                                  \
    auto kern = skinny_f16_kernel<NBITS, MT, BF16, SUB>;                                                          \
    if (lds > 64 * 1024) {                                                                                    \
      static LdsRaised raised;                                                                                \
      if (const int rc = raise_lds_limit(raised, reinterpret_cast<const void*>(kern), 160 * 1024, "hqq_hip_gemv")) return rc; \
    }                                                                                                         \
    hipLaunchKernelGGL(kern, grid, block, lds, st, a);                                                        \
    break;                                                                                                    \
  }
  switch (mt) {
    HQQ_SK_CASE(1) HQQ_SK_CASE(2) HQQ_SK_CASE(3) HQQ_SK_CASE(4)
    default: set_error("hqq_hip_gemv: M=%d outside the skinny kernel's range", a.M); return HQQ_ERR_SHAPE;
  }
#undef HQQ_SK_CASE
  return check_launch("hqq_hip_gemv");
}

// shapes this kernel covers; everything else stays on the tile kernel of gemv_mfma.hip / the library composition
bool skinny_covers(int nbits, int64_t M, int64_t K, int64_t group_size, const int64_t* N, int n_layers) {
  if ((nbits != 8 && nbits != 4 && nbits != 3 && nbits != 2) || group_size != 64 || M < 5 || M > 64 || K % SK_KC != 0 || K < 2 * SK_KC) return false;
  const int per = sk_per(nbits);
  for (int i = 0; i < n_layers; ++i)
    if (N[i] % per != 0 || N[i] / per < 1) return false;
  return true;
}

// bytes of workspace a skinny launch of this shape needs (0: it does not split K)
size_t skinny_workspace_bytes(int nbits, int n_layers, const int64_t* N, int64_t M, int64_t K, uint32_t opts) {
  const int per = sk_per(nbits);
  int64_t panels = 0;
  for (int i = 0; i < n_layers; ++i) panels += (N[i] / per + SK_ROWS - 1) / SK_ROWS;
  int ks, cps;
  sk_choose(static_cast<int>(panels), static_cast<int>(K / SK_KC), static_cast<int>(opts >> 24), ks, cps);
  return ks > 1 ? SK_CNT_BYTES + sk_part_bytes(nbits, ks, static_cast<int>(panels), static_cast<int>((M + 15) / 16)) : 0;
}

int skinny_run(int nbits, int n_layers, const void* x, const void* const* Wq, const void* const* scale, const void* const* zero,
               const void* const* bias, void* const* y, const int64_t* N, int64_t M, int64_t K, int dtype, uint32_t opts, void* ws, size_t ws_bytes,
               hipStream_t st) {
  const int per = sk_per(nbits);
  SkArgs a;
  int64_t panels = 0, ntot = 0;
  for (int i = 0; i < n_layers; ++i) {
    a.n_off[i] = static_cast<int>(ntot);
    panels += (N[i] / per + SK_ROWS - 1) / SK_ROWS;
    ntot += N[i];
    if (panels > INT32_MAX / 32 || ntot > INT32_MAX / 2) { set_error("hqq_hip_gemv: size overflow"); return HQQ_ERR_SHAPE; }
    if (nbits == 3 && (N[i] / 2) * (K / 4) * 3 > static_cast<int64_t>(UINT32_MAX)) { set_error("hqq_hip_gemv: size overflow"); return HQQ_ERR_SHAPE; }
    a.Wq[i] = static_cast<const uint8_t*>(Wq[i]);
    a.scale[i] = static_cast<const half_t*>(scale[i]);
    a.zero[i] = static_cast<const half_t*>(zero[i]);
    a.bias[i] = bias ? static_cast<const half_t*>(bias[i]) : nullptr;
    a.y[i] = static_cast<half_t*>(y[i]);
    a.N[i] = static_cast<int>(N[i]);
    a.panel_end[i] = static_cast<int>(panels);
  }
  for (int i = n_layers; i < SK_MAXL; ++i) {
    a.Wq[i] = a.Wq[n_layers - 1]; a.scale[i] = a.scale[n_layers - 1]; a.zero[i] = a.zero[n_layers - 1]; a.bias[i] = a.bias[n_layers - 1];
    a.y[i] = a.y[n_layers - 1]; a.N[i] = a.N[n_layers - 1]; a.panel_end[i] = a.panel_end[n_layers - 1]; a.n_off[i] = a.n_off[n_layers - 1];
  }
  a.K = static_cast<int>(K);
  a.G = static_cast<int>(K / 64);
  a.total_panels = static_cast<int>(panels);
  a.n_total = static_cast<int>(ntot);
  a.M = static_cast<int>(M);
  a.x = static_cast<const half_t*>(x);
  if (nbits == 3) return dtype == HQQ_BF16 ? sk_launch<3, true>(a, opts, ws, ws_bytes, st) : (opts & HQQ_OPT_META_SCALABLE) ? sk_launch<3, false, true>(a, opts, ws, ws_bytes, st) : sk_launch<3, false>(a, opts, ws, ws_bytes, st);
  if (dtype == HQQ_BF16) return nbits == 4 ? sk_launch<4, true>(a, opts, ws, ws_bytes, st) : nbits == 2 ? sk_launch<2, true>(a, opts, ws, ws_bytes, st) : sk_launch<8, true>(a, opts, ws, ws_bytes, st);
  if (opts & HQQ_OPT_META_SCALABLE)
    return nbits == 4 ? sk_launch<4, false, true>(a, opts, ws, ws_bytes, st) : nbits == 2 ? sk_launch<2, false, true>(a, opts, ws, ws_bytes, st) : sk_launch<8, false, true>(a, opts, ws, ws_bytes, st);
  return nbits == 4 ? sk_launch<4, false>(a, opts, ws, ws_bytes, st) : nbits == 2 ? sk_launch<2, false>(a, opts, ws, ws_bytes, st) : sk_launch<8, false>(a, opts, ws, ws_bytes, st);
}

}  // namespace SK_NS

#ifndef SK_NARROW
namespace sk_narrow {
bool skinny_covers(int nbits, int64_t M, int64_t K, int64_t group_size, const int64_t* N, int n_layers);
size_t skinny_workspace_bytes(int nbits, int n_layers, const int64_t* N, int64_t M, int64_t K, uint32_t opts);
int skinny_run(int nbits, int n_layers, const void* x, const void* const* Wq, const void* const* scale, const void* const* zero,
               const void* const* bias, void* const* y, const int64_t* N, int64_t M, int64_t K, int dtype, uint32_t opts, void* ws, size_t ws_bytes,
               hipStream_t st);
}  // namespace sk_narrow

// which tile serves a launch: shapes only (see the head of the file)
#ifndef SK_NARROW_MAX_PROWS
#define SK_NARROW_MAX_PROWS 2048
#endif
static bool sk_takes_narrow(int nbits, int n_layers, const int64_t* N, uint32_t opts) {
  if ((opts & HQQ_OPT_SKINNY_WIDE) || nbits == 8) return false;   // (8-bit: one slab per byte — the narrow tile's constant staging has fewer lines than threads; not built)
  const int per = sk_wide::sk_per(nbits);
  int64_t prows = 0;
  for (int i = 0; i < n_layers; ++i) prows += N[i] / per;
  return prows <= (nbits == 2 ? 4 : 1) * static_cast<int64_t>(SK_NARROW_MAX_PROWS);
}
bool skinny_covers(int nbits, int64_t M, int64_t K, int64_t group_size, const int64_t* N, int n_layers) {
  return sk_wide::skinny_covers(nbits, M, K, group_size, N, n_layers);   // (the same conditions for both tiles)
}
size_t skinny_workspace_bytes(int nbits, int n_layers, const int64_t* N, int64_t M, int64_t K, uint32_t opts) {
  return sk_takes_narrow(nbits, n_layers, N, opts) ? sk_narrow::skinny_workspace_bytes(nbits, n_layers, N, M, K, opts)
                                                   : sk_wide::skinny_workspace_bytes(nbits, n_layers, N, M, K, opts);
}
int skinny_run(int nbits, int n_layers, const void* x, const void* const* Wq, const void* const* scale, const void* const* zero,
               const void* const* bias, void* const* y, const int64_t* N, int64_t M, int64_t K, int dtype, uint32_t opts, void* ws, size_t ws_bytes,
               hipStream_t st) {
  return sk_takes_narrow(nbits, n_layers, N, opts) ? sk_narrow::skinny_run(nbits, n_layers, x, Wq, scale, zero, bias, y, N, M, K, dtype, opts, ws, ws_bytes, st)
                                                   : sk_wide::skinny_run(nbits, n_layers, x, Wq, scale, zero, bias, y, N, M, K, dtype, opts, ws, ws_bytes, st);
}
#endif

}  // namespace hqq
