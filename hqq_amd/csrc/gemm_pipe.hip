// gemm_pipe.hip — fused unpack -> dequantize -> MFMA GEMM for the rows between decode and long prefill (65 <= M <= ~1024: batched
// decode, speculative verification, short prompts) and beyond, gfx950, fp16 / bf16, 8-/4-/2-bit.
//
// Reference chain replaced (axis=1): BitPack.unpack_* -> (W_r - zero) * scale -> torch.matmul(x, W.t()) (+ bias)
//   hqq/core/bitpack.py:31-64, hqq/core/quantize.py:183-199, :880-898.
// Why a kernel of its own: in this range the layer is neither weight-streaming (skinny.hip: one pass over the packed bytes, M <= 64)
// nor big enough for a plain output-tile grid — a 4096 x 4096 layer at M = 128 has 32 tiles of 128 x 128 for 256 CUs, and the
// composition "dequantise kernel + library GEMM" pays 11-36 us for writing and re-reading the fp16 matrix around a GEMM of 20-50 us
// (tools/sweep_prefill.py).  Here the packed weights are the only weight bytes that leave HBM, and K is split across workgroups
// until the chip is full.
//
// Design (what the measurements of the first, LDS-staged version asked for: there the ds_write_b128 of the rebuilt weight tile and
// of x — ~13 cycles per wave instruction through a path two SIMDs share — were 40 % of a step, the MFMAs were free):
//   no LDS writes by the waves at all.  Every byte comes in by LDS-DMA (global_load_lds, 16 B per lane): the activation tile
//             [128 tokens x 64 k] into a ring of 4 stages, each wave's 1 KiB of packed weights (16 packed rows x 64 B) into a ring
//             of 8 slots, the (zero, scale) pairs of two steps as one dword per (row, slab) into a ring of 4.
//   weights   wave w owns packed rows 16 w .. 16 w + 15 of the tile's 64 (-> 16 PER features each); lane (r = lane & 15,
//             c = lane >> 4) rebuilds the 16 k-values 16 c .. 16 c + 15 of row r exactly (three-op form of decode_common.h where
//             the layer allows it, else four-op: the same two fp16 roundings as Quantizer.dequantize) straight into MFMA A
//             fragments, one step ahead of their use.  The packed dword's middle bytes are swapped first (one v_perm per 4 bytes), so the
//             masked pairs come out in natural k order and x needs no permutation: the DMA can bring it.
//   x         B fragment of token tile j for the k-octets (2c, 2c + 1): ds_read_b128 of chunk 2c (+1) of row 16 j + r; the chunk
//             position inside a row is XOR-ed with a function of the row (gd_swz) that makes exactly this access pattern
//             conflict-free for the 16-lane groups of ds_read_b128 — applied on the SOURCE address of the DMA, which writes linearly.
//   sync      one workgroup barrier per step; before it `s_waitcnt vmcnt(N)` with N = the DMA instructions issued after the ones
//             the next step needs (never 0: three steps of x and five of weights stay in flight across the barrier).
//   split-K   grid = tiles x KS; every split parks its fp32 tile in the caller's workspace and a second small launch adds the KS
//             tiles in split order, rounds, adds the bias and stores: fixed order, reproducible bits.
#include "decode_common.h"
#include "w3s.h"

namespace hqq {

// nbits = 3 is the 3-bit STREAM layout (w3s.h): two row slabs per packed row like the 4-bit container, 12 bytes per lane and step
constexpr int gd_per(int nbits) { return nbits == 3 ? 2 : 8 / nbits; }
constexpr int GD_K = 64;   // k per step.  Waves per workgroup NW (4 or 8: 16 NW packed rows per tile) and tokens per tile BM (128 or 256) are template parameters
constexpr int GD_MAX_KS = 16;
#ifndef GD_DX_NARROW
#define GD_DX_NARROW 4   // x stages of the 4-wave x 128-token tile
#endif
template <int NW, int BM> struct GdCfg {   // LDS rings: x stages / steps ahead, packed-weight slots / steps ahead (odd), (zero, scale) slots of two steps
  // (Round 6: the x ring of the 4-wave x 128-token tile 6 and 7 stages deep — as deep as the LDS allows, 5-6 steps ahead — changed nothing at 128..1024 rows on any
  //  7B launch: 130.9 / 136.0 / 131.9 us per block at 128 rows with 4 / 6 / 7 stages, 465 / 461 / 462 at 1024 (tools/r6/bs128.py).  A step of that tile is not bound by how
  //  much x it has in flight; at 128 rows the launch is bound by the 32 MB of fp32 partial tiles its 8 K splits park and re-read around 8.4 MB of weights.)
  static constexpr int DX = BM == 128 ? (NW == 4 ? GD_DX_NARROW : 4) : 3, PX = DX - 1;
  static constexpr int DW = BM == 128 ? 8 : 4, PW = BM == 128 ? 5 : 3;
  static constexpr int DM = BM == 128 ? 4 : 2;
  static constexpr int XSTAGE = BM * GD_K * 2;
};

// A launch serves a GROUP of up to GD_MAXL layers that read the same x (q | k | v, gate | up, or one layer; round 6): their feature tiles form one
// concatenated tile space — a workgroup looks its layer up once, before anything else — so a decoder block at 65..2560 rows is 4 launches (+ 4 split-K
// reduces) instead of 7 (+ 7).  Entries past the last layer repeat it.
constexpr int GD_MAXL = HQQ_GEMV_MAX_GROUP;
struct GdArgs {
  const half_t* x;
  const uint8_t* Wq[GD_MAXL];
  const half_t* scale[GD_MAXL];
  const half_t* zero[GD_MAXL];
  const half_t* bias[GD_MAXL];
  half_t* y[GD_MAXL];
  int N[GD_MAXL];
  int tile_end[GD_MAXL];   // end (exclusive) of layer i's feature tiles in the group's concatenated tile space
  float* part;     // [KS][split tiles][PER * BM / 16 accumulator quads][threads] x 4 fp32 (KS > 1 only)
  int M, K, G, n_tiles, m_tiles, KS, kps;
  int full;        // the first `full` tiles run whole (no split); the remaining tiles x KS splits follow (full = 0: every tile is split, or KS = 1)
};
// DMA instructions of the `back` iterations before iteration parity `par`: 1 weight piece + xp x pieces each, + ni constant pieces in the iterations that fetch them
constexpr int gd_n_out(int back, int par, int pw, int xp, int ni) {
  int n = 0;
  for (int d = 1; d <= back; ++d) n += 1 + xp + ((((par ^ (d & 1)) + pw) & 1) == 0 ? ni : 0);
  return n;
}
// DMA instructions a wave issues in the prologue AFTER the last one of step 0 (the prologue's issue order: per v = -max(pw, px) .. -1: weights of step v + pw
// [+ ni constant pieces on even steps], then xp x pieces of step v + px): what may still be in flight when step 0 is rebuilt and contracted
constexpr int gd_prologue_after_step0(int pw, int px, int xp, int ni) {
  const int pmax = pw > px ? pw : px;
  int total = 0, last0 = 0;
  for (int v = -pmax; v < 0; ++v) {
    if (v + pw >= 0) {
      total += 1;
      if (((v + pw) & 1) == 0) total += ni;
      if (v + pw == 0) last0 = total;
    }
    if (v + px >= 0) {
      total += xp;
      if (v + px == 0) last0 = total;
    }
  }
  return total - last0;
}
struct GdLayer { const uint8_t* Wq; const half_t* scale; const half_t* zero; const half_t* bias; half_t* y; int N, nt; };
// the layer of feature tile `nt` and the tile's index inside it (wave-uniform: scalar selects over the argument arrays)
__device__ __forceinline__ GdLayer gd_layer(const GdArgs& a, int nt) {
  GdLayer L{a.Wq[0], a.scale[0], a.zero[0], a.bias[0], a.y[0], a.N[0], nt};
#pragma unroll
  for (int i = 1; i < GD_MAXL; ++i) {
    const bool in = nt >= a.tile_end[i - 1];   // (entries past the last layer repeat its tile_end: never true for a valid tile)
    L.Wq = pick(in, a.Wq[i], L.Wq); L.scale = pick(in, a.scale[i], L.scale); L.zero = pick(in, a.zero[i], L.zero); L.bias = pick(in, a.bias[i], L.bias);
    L.y = pick(in, a.y[i], L.y); L.N = pick(in, a.N[i], L.N); L.nt = pick(in, nt - a.tile_end[i - 1], L.nt);
  }
  return L;
}

// chunk position (16 B units) inside a 128-byte row of the x stage: chunk ^ gd_swz(row).  Found by search over the GF(2)-linear maps
// row -> 3 bits: with it the four 16-lane groups of a ds_read_b128 whose lane (r, c) reads chunk 2 c + h of row 16 j + r touch 16
// distinct 16-byte slots of the 256-byte bank row, for both h.
__device__ __forceinline__ int gd_swz(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 2); }

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
__device__ __forceinline__ void gd_dma16(const void* src, uint8_t* lds_wave_base) {   // lane l: 16 bytes from src -> lds_wave_base + 16 l
  __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ void gd_dma12(const void* src, uint8_t* lds_wave_base) {   // lane l: 12 bytes -> lds_wave_base + 16 l (the 12-byte form keeps the 16-byte lane stride: measured, tools/dma12_probe.hip)
  __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)lds_wave_base, 12, 0, 0);
}
__device__ __forceinline__ void gd_dma4(const void* src, uint8_t* lds_wave_base) {    // lane l: 4 bytes -> lds_wave_base + 4 l
  __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)lds_wave_base, 4, 0, 0);
}

template <int NBITS, int S, int PER, bool SUB>
struct GdSlab {   // the lane's 16 k-values of slab S (dwords with swapped middle bytes) -> A fragments a0 (k 16c .. +7), a1 (k 16c+8 .. +15)
  static __device__ __forceinline__ void run(const u32x4& w, const half_t (&z)[PER], const half_t (&s)[PER], h8_t (&a0)[PER], h8_t (&a1)[PER]) {
    constexpr int sh = NBITS * (PER - 1 - S);
    constexpr uint32_t m1 = ((NBITS == 8) ? 0xFFu : ((1u << NBITS) - 1u)) << sh;
    constexpr uint32_t m = m1 | (m1 << 16);
    half2_t q[8];
    uint32_t o[8];
    if constexpr (SUB) {
      constexpr int J = 9 - sh;
      const half_t zj = z[S] * static_cast<half_t>(1.0f / static_cast<float>(1 << J));   // exact (hqq_hip_meta_check)
      const half_t sj = s[S] * static_cast<half_t>(static_cast<float>(1 << J));
      const half2_t nz = {-zj, -zj}, ss = {sj, sj};
      const half2_t lift = {static_cast<half_t>(32768.0f), static_cast<half_t>(32768.0f)};
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        q[2 * d] = as_h2(w[d] & m);              // (k 4d, 4d+1): q * 2^(sh-24), a subnormal pair
        q[2 * d + 1] = as_h2((w[d] >> 8) & m);   // (k 4d+2, 4d+3)
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) q[i] = __builtin_elementwise_fma(q[i], lift, nz);   // rounding 1
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = __builtin_bit_cast(uint32_t, q[i] * ss);      // rounding 2
    } else {
      constexpr float inv = 1.0f / static_cast<float>(1 << sh);
      const half2_t k1 = {static_cast<half_t>(inv), static_cast<half_t>(inv)};
      const half2_t k2 = {static_cast<half_t>(-1024.0f * inv), static_cast<half_t>(-1024.0f * inv)};
      const half2_t zz = {z[S], z[S]}, ss = {s[S], s[S]};
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        q[2 * d] = as_h2((w[d] & m) | 0x64006400u);
        q[2 * d + 1] = as_h2(((w[d] >> 8) & m) | 0x64006400u);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) q[i] = __builtin_elementwise_fma(q[i], k1, k2);   // exact integer level
#pragma unroll
      for (int i = 0; i < 8; ++i) q[i] = q[i] - zz;                                  // rounding 1
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = __builtin_bit_cast(uint32_t, q[i] * ss);    // rounding 2
    }
    a0[S] = __builtin_bit_cast(h8_t, u32x4{o[0], o[1], o[2], o[3]});
    a1[S] = __builtin_bit_cast(h8_t, u32x4{o[4], o[5], o[6], o[7]});
    if constexpr (S + 1 < PER) GdSlab<NBITS, S + 1, PER, SUB>::run(w, z, s, a0, a1);
  }
};

// bf16 compute dtype: the reference's two roundings are to bf16 (quantize.py:198 on bf16 tensors).  gfx950 has no packed bf16
// arithmetic, so a weight goes through fp32 (as in skinny.hip): v_cvt_f32_ubyteN lifts the masked byte F q, one fma forms q - z exactly,
// v_cvt_pk_bf16_f32 rounds it (RNE), v_dot2_f32_bf16 against (s, 0) / (0, s) forms the exact product with s, a second v_cvt_pk rounds
// again — 8 VALU ops per weight pair against 3.  Bytes are taken in natural k order, so the dwords are NOT byte-swapped on this path.
typedef __bf16 gd_bf2_t __attribute__((ext_vector_type(2)));
typedef __bf16 gd_bf8_t __attribute__((ext_vector_type(8)));
typedef float gd_f2_t __attribute__((ext_vector_type(2)));
template <int B> __device__ __forceinline__ float gd_ubyte(uint32_t v) { return static_cast<float>((v >> (8 * B)) & 0xFFu); }   // v_cvt_f32_ubyteB
template <int NBITS, int S, int PER>
struct GdSlabBF {
  static __device__ __forceinline__ void run(const u32x4& w, const uint16_t (&z)[PER], const uint16_t (&s)[PER], u32x4 (&a0)[PER], u32x4 (&a1)[PER]) {
    constexpr int sh = NBITS * (PER - 1 - S);
    constexpr uint32_t m1 = ((NBITS == 8) ? 0xFFu : ((1u << NBITS) - 1u)) << sh;
    constexpr float inv = 1.0f / static_cast<float>(1 << sh);
    const float zf = __uint_as_float(static_cast<uint32_t>(z[S]) << 16);
    const gd_bf2_t s_lo = __builtin_bit_cast(gd_bf2_t, static_cast<uint32_t>(s[S]));          // (s, 0)
    const gd_bf2_t s_hi = __builtin_bit_cast(gd_bf2_t, static_cast<uint32_t>(s[S]) << 16);    // (0, s)
    uint32_t o[8];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const uint32_t fq = (NBITS == 8) ? w[d] : (w[d] & (m1 * 0x01010101u));
      const gd_f2_t dq[2] = {{__builtin_fmaf(gd_ubyte<0>(fq), inv, -zf), __builtin_fmaf(gd_ubyte<1>(fq), inv, -zf)},    // k 4d, 4d+1
                             {__builtin_fmaf(gd_ubyte<2>(fq), inv, -zf), __builtin_fmaf(gd_ubyte<3>(fq), inv, -zf)}};   // k 4d+2, 4d+3
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const gd_bf2_t dr = __builtin_convertvector(dq[h], gd_bf2_t);                 // rounding 1
        const gd_f2_t pw = {__builtin_amdgcn_fdot2_f32_bf16(dr, s_lo, 0.f, false), __builtin_amdgcn_fdot2_f32_bf16(dr, s_hi, 0.f, false)};
        o[2 * d + h] = __builtin_bit_cast(uint32_t, __builtin_convertvector(pw, gd_bf2_t));   // rounding 2
      }
    }
    a0[S] = u32x4{o[0], o[1], o[2], o[3]};
    a1[S] = u32x4{o[4], o[5], o[6], o[7]};
    if constexpr (S + 1 < PER) GdSlabBF<NBITS, S + 1, PER>::run(w, z, s, a0, a1);
  }
};
// one fp32 accumulator -> the compute dtype, + bias in the compute dtype (`out += bias` on the rounded matmul result, quantize.py:896-897)
template <bool BF> __device__ __forceinline__ uint16_t gd_out(float v, const half_t* bias, int n) {
  if constexpr (BF) {
    uint16_t o = f32_to_bf16(v);
    if (bias) o = f32_to_bf16(bf16_to_f32(o) + bf16_to_f32(reinterpret_cast<const uint16_t*>(bias)[n]));
    return o;
  } else {
    half_t o = static_cast<half_t>(v);
    if (bias) o = o + bias[n];
    return __builtin_bit_cast(uint16_t, o);
  }
}

template <int NBITS> struct GdMeta {   // (zero, scale) DMA: one dword = the two steps' values of one (row, slab, zero | scale)
  static constexpr int PER = gd_per(NBITS);
  static constexpr int NI = (PER * 2 * 16 + 63) / 64;         // DMA instructions per wave and pair of steps
  static constexpr int SLOT = NI * 256;                        // bytes per wave and pair of steps
};

// 2-bit, 4-wave tile: four slabs x eight token tiles of accumulators (128 registers) + the rebuilt fragments do not fit 256 registers — round 4's build
// spilled 14-80 dwords into scratch INSIDE the loop.  With one wave per SIMD (a 512-register budget) the accumulators live in AGPRs and nothing spills:
// 20-30 % faster on every 7B shape at 128..2048 rows (tools/r5_pipe2bit.py: 4096 x 4096 at 128 / 1024 rows 37.3 -> 29.9 / 78.8 -> 63.3 us, 11008 x 4096
// 43.7 -> 35.3 / 202 -> 141), and ahead of the 8-wave tile (which cannot have that budget) everywhere: 2-bit layers always take the 4-wave tile (gp_plan).
#ifndef GD_2BIT_ONE_WAVE_PER_SIMD
#define GD_2BIT_ONE_WAVE_PER_SIMD 1
#endif
template <int NBITS, bool SUB, int NW, int BM, bool BF>
__global__ __launch_bounds__(64 * NW, (NW == 4 && !(NBITS == 2 && GD_2BIT_ONE_WAVE_PER_SIMD)) ? 2 : 1) void gemm_pipe_f16_kernel(const GdArgs a) {   // ("2": a 256-register budget keeps the accumulators in VGPRs; with 512 hipcc parks them in AGPRs and copies)
  constexpr bool W3 = NBITS == 3;
  constexpr int PER = gd_per(NBITS);
  constexpr int LB = W3 ? 12 : 16;   // bytes per lane and step (16 k of PER rows)
  using CF = GdCfg<NW, BM>;
  constexpr int GD_BM = BM, GD_MT = BM / 16, GD_DX = CF::DX, GD_PX = CF::PX, GD_DW = CF::DW, GD_PW = CF::PW, GD_DM = CF::DM, GD_XSTAGE = CF::XSTAGE;
  constexpr int GD_WAVES = NW, GD_T = 64 * NW, GD_PROWS = 16 * NW, XP = BM / 8 / NW;   // XP: x DMA pieces (1 KiB = 8 token rows) per wave and step
  using MD = GdMeta<NBITS>;
  constexpr int X_BYTES = GD_DX * GD_XSTAGE;                   // 64 KiB
  constexpr int W_BYTES = GD_DW * GD_WAVES * 1024;             // 32 / 64 KiB
  extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];   // [x ring | weight ring | (zero, scale) ring]
  uint8_t* const xring = lds;
  uint8_t* const wring = lds + X_BYTES;
  uint8_t* const mring = lds + X_BYTES + W_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, c = lane >> 4;
  // XCD-aware order (workgroup b runs on XCD b % 8 — observed; a speed assumption only): each XCD gets a contiguous run of logical
  // tiles, i.e. a band of token tiles x all feature tiles — its L2 then holds a few x tiles and one pass over the packed weights
  // instead of every x tile of the round.  Bijective for any grid size.
  // A plan with more tiles than CUs whose last round would be partly empty runs the full rounds whole and splits only the tiles of
  // the last round (they are the highest workgroup ids: dispatched last, finishing together with K / KS steps each).
  const int tiles_all = a.n_tiles * a.m_tiles;
  const bool whole = static_cast<int>(blockIdx.x) < a.full || a.KS == 1;
  int tile, ks;
  {
    const int base = whole ? 0 : a.full;
    const int nwg = whole ? (a.KS == 1 ? static_cast<int>(gridDim.x) : a.full) : static_cast<int>(gridDim.x) - a.full;
    const int h = static_cast<int>(blockIdx.x) - base, xcd = h & 7, idx = h >> 3;
    const int q8 = nwg >> 3, r8 = nwg & 7;
    const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    const int split_tiles = tiles_all - a.full;
    tile = whole ? L : a.full + L % split_tiles;
    ks = whole ? 0 : L / split_tiles;
  }
  const int mt = tile / a.n_tiles;
  const GdLayer L = gd_layer(a, tile % a.n_tiles);
  const int nt = L.nt;
  const int N = L.N, K = a.K, M = a.M, G = a.G;
  const int rows_per_slab = N / PER;
  const int p0 = nt * GD_PROWS + wave * 16, m0 = mt * GD_BM;
  const int nk = K / GD_K;
  const int kt0 = ks * a.kps;                                   // even (gp_plan)
  const int nsteps = whole ? nk : (kt0 + a.kps < nk ? kt0 + a.kps : nk) - kt0;

  // ---- per-lane DMA sources (rows past the end of the slab read the last row and are masked by a zero scale below; token rows past
  //      M read row 0: their accumulator columns are never stored, and a column depends on its own x row only) ----
  const bool w_active = (p0 + r) < rows_per_slab;
  const int wrow = w_active ? p0 + r : rows_per_slab - 1;
  const uint8_t* wsrc = L.Wq + static_cast<int64_t>(wrow) * (K / 16 * LB) + c * LB + static_cast<int64_t>(kt0) * (4 * LB);
  const half_t* xsrc[XP];
#pragma unroll
  for (int q = 0; q < XP; ++q) {   // piece XP w + q fills rows 8 (XP w + q) .. + 7 of the stage: lane -> (row, position)
    const int row = 8 * (XP * wave + q) + (lane >> 3), pos = lane & 7;
    const int chunk = pos ^ gd_swz(row);
    xsrc[q] = a.x + static_cast<int64_t>(m0 + row < M ? m0 + row : 0) * K + chunk * 8 + static_cast<int64_t>(kt0) * GD_K;
  }
  const half_t* msrc[MD::NI];
#pragma unroll
  for (int t = 0; t < MD::NI; ++t) {   // element e = 64 t + lane: row e & 15, slab (e >> 4) % PER, zero | scale (e >> 4) / PER
    const int e = 64 * t + lane, er = e & 15, es = (e >> 4) % PER, which = ((e >> 4) / PER) & 1;
    const int prow = (p0 + er) < rows_per_slab ? p0 + er : rows_per_slab - 1;
    msrc[t] = (which ? L.scale : L.zero) + (static_cast<int64_t>(es) * rows_per_slab + prow) * G + kt0;
  }
  const uint16_t smask = w_active ? 0xFFFFu : 0u;   // rows past the end of the slab: scale 0 -> exact zero weights

  auto issue_w = [&](int step) {   // packed weights of `step`
    const int sc = step < nsteps ? step : nsteps - 1;   // past the range: the last step again (cached; lands in a slot nobody reads)
    if constexpr (W3) gd_dma12(wsrc + static_cast<int64_t>(sc) * (4 * LB), wring + ((step % GD_DW) * GD_WAVES + wave) * 1024);
    else gd_dma16(wsrc + static_cast<int64_t>(sc) * GD_K, wring + ((step % GD_DW) * GD_WAVES + wave) * 1024);
  };
  auto issue_m = [&](int step) {   // step even: the (zero, scale) pairs of steps (step, step + 1)
    const int sc = step < nsteps ? step : ((nsteps - 1) & ~1);
#pragma unroll
    for (int t = 0; t < MD::NI; ++t) gd_dma4(msrc[t] + sc, mring + (((step >> 1) % GD_DM) * GD_WAVES + wave) * MD::SLOT + t * 256);
  };
  auto issue_x = [&](int step) {
    const int sc = step < nsteps ? step : nsteps - 1;
#pragma unroll
    for (int q = 0; q < XP; ++q) gd_dma16(xsrc[q] + static_cast<int64_t>(sc) * GD_K, xring + (step % GD_DX) * GD_XSTAGE + (XP * wave + q) * 1024);
  };

  // ---- the lane's packed bytes and group constants of a step, out of the rings; rebuilt into A fragments ----
  uint16_t zc[2][PER], sc_[2][PER];   // (zero, scale) bits of the current pair of steps, per slab: [parity][slab]
  auto fetch_meta = [&](int step) {   // step even: both steps' constants in one read per (slab, zero | scale)
    const uint8_t* slot = mring + (((step >> 1) % GD_DM) * GD_WAVES + wave) * MD::SLOT;
#pragma unroll
    for (int s = 0; s < PER; ++s) {
      const uint32_t zd = *reinterpret_cast<const uint32_t*>(slot + ((0 * PER + s) * 16 + r) * 4);
      const uint32_t sd = *reinterpret_cast<const uint32_t*>(slot + ((1 * PER + s) * 16 + r) * 4);
      zc[0][s] = static_cast<uint16_t>(zd); zc[1][s] = static_cast<uint16_t>(zd >> 16);
      sc_[0][s] = static_cast<uint16_t>(sd) & smask; sc_[1][s] = static_cast<uint16_t>(sd >> 16) & smask;
    }
  };
  auto read_w = [&](int step) {
    const uint8_t* slot = wring + ((step % GD_DW) * GD_WAVES + wave) * 1024;
    return *reinterpret_cast<const u32x4*>(slot + lane * 16);   // (3-bit stream layout: 12 bytes landed at the same 16-byte lane stride; the fourth dword is not read)
  };
  auto rebuild = [&](const u32x4& raw, int step, u32x4 (&a0)[PER], u32x4 (&a1)[PER]) {
    if constexpr (W3) {   // the stream layout is in natural k order already
      const uint32_t zs[2] = {static_cast<uint32_t>(zc[step & 1][0]) | (static_cast<uint32_t>(sc_[step & 1][0]) << 16),
                              static_cast<uint32_t>(zc[step & 1][1]) | (static_cast<uint32_t>(sc_[step & 1][1]) << 16)};
      if constexpr (BF) {
        w3s_bf8_t h0[2], h1[2];
        w3s_rebuild_bf16(raw.x, raw.y, raw.z, zs, h0, h1);
#pragma unroll
        for (int s = 0; s < 2; ++s) { a0[s] = __builtin_bit_cast(u32x4, h0[s]); a1[s] = __builtin_bit_cast(u32x4, h1[s]); }
      } else {
        h8_t h0[2], h1[2];
        w3s_rebuild_f16<SUB>(raw.x, raw.y, raw.z, zs, 0x64006400u, h0, h1);
#pragma unroll
        for (int s = 0; s < 2; ++s) { a0[s] = __builtin_bit_cast(u32x4, h0[s]); a1[s] = __builtin_bit_cast(u32x4, h1[s]); }
      }
    } else if constexpr (BF) {
      GdSlabBF<NBITS, 0, PER>::run(raw, zc[step & 1], sc_[step & 1], a0, a1);
    } else {
      u32x4 w;
#pragma unroll
      for (int d = 0; d < 4; ++d) w[d] = __builtin_amdgcn_perm(raw[d], raw[d], 0x03010200u);   // bytes (b0,b1,b2,b3) -> (b0,b2,b1,b3)
      half_t zh[PER], sh_[PER];
      h8_t h0[PER], h1[PER];
#pragma unroll
      for (int s = 0; s < PER; ++s) { zh[s] = __builtin_bit_cast(half_t, zc[step & 1][s]); sh_[s] = __builtin_bit_cast(half_t, sc_[step & 1][s]); }
      GdSlab<NBITS, 0, PER, SUB>::run(w, zh, sh_, h0, h1);
#pragma unroll
      for (int s = 0; s < PER; ++s) { a0[s] = __builtin_bit_cast(u32x4, h0[s]); a1[s] = __builtin_bit_cast(u32x4, h1[s]); }
    }
  };

  f32x4 acc[PER][GD_MT];
#pragma unroll
  for (int s = 0; s < PER; ++s)
#pragma unroll
    for (int j = 0; j < GD_MT; ++j) acc[s][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- prologue: the issue order of the steady state (weights PW steps ahead, then x PX steps ahead); step 0 waited for ----
  constexpr int GD_PMAX = GD_PW > GD_PX ? GD_PW : GD_PX;
#pragma unroll
  for (int v = -GD_PMAX; v < 0; ++v) {
    if (v + GD_PW >= 0) {
      issue_w(v + GD_PW);
      if (((v + GD_PW) & 1) == 0) issue_m(v + GD_PW);
    }
    if (v + GD_PX >= 0) issue_x(v + GD_PX);
  }
  {
    // only step 0 has to have landed: the steps behind it stay in flight across the barrier and the loop's own counted waits take over from iteration 0
    // (until round 6 everything was drained here: at 128 rows — 8-16 steps per workgroup — that was 4 % of a launch; 2 % at 256, nothing from 1024 on)
    constexpr int K0 = gd_prologue_after_step0(GD_PW, GD_PX, XP, MD::NI);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K0) : "memory");
  }
  __builtin_amdgcn_s_barrier();
  u32x4 a0[2][PER], a1[2][PER];   // A fragments of the current / next step (8 fp16 / bf16 values each)
  fetch_meta(0);
  rebuild(read_w(0), 0, a0[0], a1[0]);

  // ---- main loop, one step per iteration (unrolled by two: the (zero, scale) ring moves every other step).  With one wave per SIMD
  //      nothing hides an LDS round trip but the wave's own MFMAs, so the step is cut in parts of 64 tokens and every fragment read is
  //      issued one part ahead of the MFMAs that use it — the workgroup barrier sits INSIDE a step's MFMA work.  Iteration i (two parts):
  //        LDS    B fragments of (step i, tokens 64..127)
  //        MFMA   (step i, tokens 0..63)            [fragments read during iteration i - 1]
  //        wait   all but the DMA instructions of iteration i - 1: x of step i + 1 and the weights of step i + 1 have landed; barrier
  //        DMA    weights of step i + 5 (+ constants), x of step i + 3   [the slots they overwrite were last read before this barrier]
  //        LDS    B fragments of (step i + 1, tokens 0..63); packed bytes (+ constants) of step i + 1
  //        MFMA   (step i, tokens 64..127), and under them the VALU rebuild of step i + 1 into the other A fragment set ----
  constexpr int HT = 4, NQ = GD_MT / HT;   // the step's tokens in NQ parts of 64 (2 at 128 tokens per tile, 4 at 256)
  u32x4 bA0[HT], bA1[HT], bB0[HT], bB1[HT];   // B fragments of the even / odd parts
  auto read_b = [&](int step, int part, u32x4 (&f0)[HT], u32x4 (&f1)[HT]) {
    const uint8_t* xs = xring + (step % GD_DX) * GD_XSTAGE;
#pragma unroll
    for (int j = 0; j < HT; ++j) {
      const int row = (part * HT + j) * 16 + r;
      f0[j] = *reinterpret_cast<const u32x4*>(xs + row * 128 + (((2 * c) ^ gd_swz(row)) << 4));
      f1[j] = *reinterpret_cast<const u32x4*>(xs + row * 128 + (((2 * c + 1) ^ gd_swz(row)) << 4));
    }
  };
  auto mfma = [&](const u32x4& A, const u32x4& B, f32x4 C) {
    if constexpr (BF) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(gd_bf8_t, A), __builtin_bit_cast(gd_bf8_t, B), C, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_t, A), __builtin_bit_cast(h8_t, B), C, 0, 0, 0);
  };
  auto mma = [&](int part, const u32x4 (&ca0)[PER], const u32x4 (&ca1)[PER], const u32x4 (&f0)[HT], const u32x4 (&f1)[HT]) {
#pragma unroll
    for (int s = 0; s < PER; ++s)
#pragma unroll
      for (int j = 0; j < HT; ++j) acc[s][part * HT + j] = mfma(ca0[s], f0[j], acc[s][part * HT + j]);
#pragma unroll
    for (int s = 0; s < PER; ++s)
#pragma unroll
      for (int j = 0; j < HT; ++j) acc[s][part * HT + j] = mfma(ca1[s], f1[j], acc[s][part * HT + j]);
  };
  read_b(0, 0, bA0, bA1);
  // DMA instructions issued after the ones the next step needs: the groups of the PX - 2 iterations in between
  auto iter = [&](int i, auto parity, u32x4 (&ca0)[PER], u32x4 (&ca1)[PER], u32x4 (&na0)[PER], u32x4 (&na1)[PER]) {
    constexpr int par = decltype(parity)::value;   // i & 1
    // what may stay in flight across the barrier: x and the weights of step i + 1 were issued min(PX, PW) - 1 iterations ago, so the groups of the
    // min(PX, PW) - 2 iterations since (one weight piece, XP x pieces, and the constants every other iteration) need not have landed
    constexpr int N_BACK = (GD_PX < GD_PW ? GD_PX : GD_PW) - 2;
    constexpr int N_OUT = gd_n_out(N_BACK, par, GD_PW, XP, MD::NI);
#pragma unroll
    for (int q = 1; q < NQ; ++q) {
      if (q & 1) read_b(i, q, bB0, bB1); else read_b(i, q, bA0, bA1);
      __builtin_amdgcn_sched_barrier(0);   // reads first: left alone the scheduler sinks them below the MFMAs they were meant to hide under
      if (q & 1) mma(q - 1, ca0, ca1, bA0, bA1); else mma(q - 1, ca0, ca1, bB0, bB1);
    }
    // (lgkmcnt(0): this wave's fragment reads have left the LDS before another wave's DMA may overwrite the stage)
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N_OUT) : "memory");
    __builtin_amdgcn_s_barrier();
    constexpr bool SPREAD = NW == 8;   // measured: +5 % with two waves per SIMD, -3..9 % with one (there the earlier issue matters more)
    auto issue_all = [&]() {
      issue_w(i + GD_PW);
      if constexpr (((par + GD_PW) & 1) == 0) issue_m(i + GD_PW);
      issue_x(i + GD_PX);
    };
    if constexpr (!SPREAD) issue_all();
    read_b(i + 1, 0, bA0, bA1);
    if constexpr (par == 1) fetch_meta(i + 1);
    const u32x4 raw = read_w(i + 1);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (SPREAD) issue_all();   // the DMA issue (60-185 cycles a piece in a burst) goes UNDER the last part's MFMAs, like the rebuild's VALU work
    rebuild(raw, i + 1, na0, na1);
    mma(NQ - 1, ca0, ca1, bB0, bB1);
    if constexpr (SPREAD) {
      constexpr int NMF = PER * HT * 2, NDMA = 1 + XP + MD::NI;
#pragma unroll
      for (int t = 0; t < NMF; ++t) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, (100 + NMF - 1) / NMF, 0);
        if (t * NDMA / NMF != (t + 1) * NDMA / NMF) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);   // one DMA piece every NMF / NDMA MFMAs
      }
    }
  };
  // (A staggered order for the second wave of each SIMD — its pure-MFMA parts beside the other's rebuild — was built and measured 5 % SLOWER than
  //  lockstep, bit-identical: profiles/r03_pipe8k_ablation.txt.  The matrix pipe is not what the two waves of a SIMD contend for.)
  for (int i = 0; i < nsteps; i += 2) {
    iter(i, std::integral_constant<int, 0>{}, a0[0], a1[0], a0[1], a1[1]);
    if (i + 1 < nsteps) iter(i + 1, std::integral_constant<int, 1>{}, a0[1], a1[1], a0[0], a1[0]);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the clamped DMAs past the last step: nothing may land in LDS after the workgroup is gone)

  // ---- D layout: lane (column r = token inside tile j, rows 4 c + i = packed row inside the wave's 16) ----
  if (!whole) {   // park the split's fp32 tile in accumulator order (a wave writes 1 KiB of consecutive bytes per instruction)
    const int64_t tiles = tiles_all - a.full;
    f32x4* mine = reinterpret_cast<f32x4*>(a.part) + (static_cast<int64_t>(ks) * tiles + (tile - a.full)) * (PER * GD_MT * GD_T) + tid;
#pragma unroll
    for (int s = 0; s < PER; ++s)
#pragma unroll
      for (int j = 0; j < GD_MT; ++j) __builtin_nontemporal_store(acc[s][j], mine + (s * GD_MT + j) * GD_T);
    return;
  }
  const int pb = p0 + 4 * c;
  if (pb >= rows_per_slab) return;
#pragma unroll
  for (int s = 0; s < PER; ++s) {
    const int n = s * rows_per_slab + pb;
#pragma unroll
    for (int j = 0; j < GD_MT; ++j) {
      const int m = m0 + 16 * j + r;
      if (m >= M) continue;
      uint16_t o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = gd_out<BF>(acc[s][j][i], (L.bias && pb + i < rows_per_slab) ? L.bias : nullptr, n + i);
      uint16_t* dst = reinterpret_cast<uint16_t*>(L.y) + static_cast<int64_t>(m) * N + n;
      if (pb + 3 < rows_per_slab) {
        *reinterpret_cast<u32x2*>(dst) = *reinterpret_cast<u32x2*>(o);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (pb + i < rows_per_slab) dst[i] = o[i];
      }
    }
  }
}

// Second launch of a split-K call: output quad (tile, slab s, token tile j, thread) = the sum of the KS parked tiles in split order
// (four tiles' loads of a thread in flight at once, every CU takes part), rounded once, + bias.  One finishing workgroup per tile
// inside the first kernel (ticket scheme) read its KS x 64 KiB alone and cost 2-3 us per split.
template <int NBITS, int NW, int BM, bool BF>
__global__ __launch_bounds__(64 * NW) void gemm_pipe_reduce_kernel(const GdArgs a) {
  constexpr int PER = gd_per(NBITS), GD_T = 64 * NW, GD_PROWS = 16 * NW, GD_BM = BM, GD_MT = BM / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, c = lane >> 4;
  const int sj = blockIdx.x % (PER * GD_MT), stile = blockIdx.x / (PER * GD_MT), tile = a.full + stile;   // (the split tiles only)
  const int s = sj / GD_MT, j = sj % GD_MT;
  const int mt = tile / a.n_tiles;
  const GdLayer L = gd_layer(a, tile % a.n_tiles);
  const int nt = L.nt;
  const int64_t tiles = static_cast<int64_t>(a.n_tiles) * a.m_tiles - a.full;
  const f32x4* src = reinterpret_cast<const f32x4*>(a.part) + (static_cast<int64_t>(stile) * (PER * GD_MT) + sj) * GD_T + tid;
  const int64_t kstride = tiles * (PER * GD_MT * GD_T);
  f32x4 sum = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < a.KS; k0 += 4) {
    f32x4 t[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) t[kk] = __builtin_nontemporal_load(src + (k0 + kk < a.KS ? k0 + kk : a.KS - 1) * kstride);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
      if (k0 + kk < a.KS) { sum[0] += t[kk][0]; sum[1] += t[kk][1]; sum[2] += t[kk][2]; sum[3] += t[kk][3]; }
  }
  const int rows_per_slab = L.N / PER;
  const int pb = nt * GD_PROWS + wave * 16 + 4 * c;
  const int m = mt * GD_BM + 16 * j + r;
  if (pb >= rows_per_slab || m >= a.M) return;
  const int n = s * rows_per_slab + pb;
  uint16_t o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = gd_out<BF>(sum[i], (L.bias && pb + i < rows_per_slab) ? L.bias : nullptr, n + i);
  uint16_t* dst = reinterpret_cast<uint16_t*>(L.y) + static_cast<int64_t>(m) * L.N + n;
  if (pb + 3 < rows_per_slab) {
    *reinterpret_cast<u32x2*>(dst) = *reinterpret_cast<u32x2*>(o);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (pb + i < rows_per_slab) dst[i] = o[i];
  }
}

// ---- host side --------------------------------------------------------------------------------------------------------------------
struct GpPlan { int NW, BM, n_tiles, m_tiles, KS, kps, full; };   // full: tiles that run whole before the split ones (0 unless the plan is a hybrid)

// the layers of a launch (one layer: nl = 1).  A plan looks at the group as ONE layer of sum(N) features whose feature tiles never straddle two layers.
struct GpGroup { int nl; int64_t N[GD_MAXL]; int64_t Ntot; };
static GpGroup gp_group(const int64_t* N, int nl) {
  GpGroup g;
  g.nl = nl; g.Ntot = 0;
  for (int i = 0; i < GD_MAXL; ++i) { g.N[i] = N[i < nl ? i : nl - 1]; if (i < nl) g.Ntot += N[i]; }
  return g;
}
static int gp_feature_tiles(const GpGroup& g, int nbits, int nw) {
  int64_t t = 0;
  for (int i = 0; i < g.nl; ++i) t += (g.N[i] / gd_per(nbits) + 16 * nw - 1) / (16 * nw);
  return static_cast<int>(t);
}

static GpPlan gp_make(int nbits, int64_t M, const GpGroup& g, int64_t K, int nw, int bm, int ks) {
  GpPlan p;
  const int nk = static_cast<int>(K / GD_K);
  p.NW = nw;
  p.BM = bm;
  p.full = 0;
  p.m_tiles = static_cast<int>((M + bm - 1) / bm);
  p.n_tiles = gp_feature_tiles(g, nbits, nw);
  if (ks > GD_MAX_KS) ks = GD_MAX_KS;
  if (ks < 1) ks = 1;
  p.kps = (nk + ks - 1) / ks;
  p.kps += p.kps & 1;                                   // even: the (zero, scale) DMA fetches two steps per dword
  p.KS = (nk + p.kps - 1) / p.kps;                      // no empty split
  return p;
}

// Estimated time of a plan in microseconds: a model of the measurements in profiles/r02_prefill_sweep.md (MI355X, one workgroup per CU):
// rounds of workgroups x (steps x time per step + a fixed 5 us), the time per step growing with the number of CUs that pull the same x
// tiles through L2 at once; a split adds the second launch and 0.4 us per MiB of parked fp32 tiles.  Picks the measured-best
// (waves, splits) for 15 of the 16 Llama-2-7B cases swept and is within 12 % of the measured time everywhere.
static double gp_tstep(const GpPlan& p, double active) {
  // us per 64-k step at low load, and its growth with the number of CUs at work (fitted to tools/lab_pipe_plan.py, 640..2048 rows)
  const double base = p.BM == 256 ? 1.37 : (p.NW == 4 ? 0.50 : 0.85);
  const double slope = p.BM == 256 ? 0.20 : (p.NW == 4 ? 0.40 : 0.28);
  return base * (1.0 + slope * active / 256.0);
}
static double gp_cost(const GpPlan& p, int64_t M, int64_t N, int nk) {
  const double tiles = static_cast<double>(p.n_tiles) * p.m_tiles;
  const double split_tiles = tiles - p.full;
  double t = 0.0;
  if (p.full > 0) t += (p.full / 256) * (nk * gp_tstep(p, 256.0) + 5.0);           // whole rounds of unsplit tiles (full is a multiple of 256)
  const double wgs = p.KS > 1 ? split_tiles * p.KS : tiles;
  const int64_t rounds = (static_cast<int64_t>(wgs) + 255) / 256;
  const double active = wgs < 256.0 ? wgs : 256.0;
  t += static_cast<double>(rounds) * ((p.KS > 1 ? p.kps : nk) * gp_tstep(p, active) + 5.0);
  if (p.KS > 1) t += 1.5 + 0.4 * p.KS * split_tiles * p.BM * (16.0 * p.NW * (static_cast<double>(N) / p.n_tiles / (16.0 * p.NW))) * 4.0 / 1.0e6;
  return t;
}

// Shapes only (never the data): the split depends on (M, N, K), so a row of y can differ in the last bit between batch sizes that
// choose different splits — as with any split-K GEMM — but is reproducible run to run.
static GpPlan gp_plan(int nbits, int64_t M, const GpGroup& grp, int64_t K, uint32_t opts) {
  const int64_t N = grp.Ntot;
  const int nk = static_cast<int>(K / GD_K);
  const int forced_ks = static_cast<int>(opts >> 24);
  const int forced_nw = (opts & HQQ_OPT_GEMM_WIDE) ? 8 : (opts & HQQ_OPT_GEMM_NARROW) ? 4 : 0;
  const bool both = (opts & HQQ_OPT_GEMM_WIDE) && (opts & HQQ_OPT_GEMM_NARROW);   // both bits: the 256-token tile (8 waves)
  static const int KSS[] = {1, 2, 3, 4, 5, 6, 8, 10, 12, 16};
  static const int SHAPES[3][2] = {{4, 128}, {8, 128}, {8, 256}};
  if (nbits == 2) {   // the 4-wave tile only (see GD_2BIT_ONE_WAVE_PER_SIMD): the split is still chosen by the model (or forced)
    GpPlan b2 = gp_make(2, M, grp, K, 4, 128, forced_ks ? forced_ks : 1);
    double c2 = gp_cost(b2, M, N, nk);
    for (int ks : KSS) {
      if (forced_ks) break;
      GpPlan p = gp_make(2, M, grp, K, 4, 128, ks);
      if (p.KS > 1 && (p.kps < 16 || nk / p.KS < 16)) continue;
      const double c = gp_cost(p, M, N, nk);
      if (c < c2) { b2 = p; c2 = c; }
    }
    return b2;
  }
  GpPlan best = gp_make(nbits, M, grp, K, both ? 8 : (forced_nw ? forced_nw : 4), both && nbits != 2 ? 256 : 128, forced_ks ? forced_ks : 1);
  double best_cost = gp_cost(best, M, N, nk);
  for (const auto& sh : SHAPES) {
    if (both ? sh[1] != 256 : (forced_nw && (sh[0] != forced_nw || sh[1] != 128))) continue;
    if (sh[1] == 256 && nbits == 2) continue;   // four slabs x 16 token tiles of accumulators do not fit the register file
    for (int ks : KSS) {
      if (forced_ks && ks != 1) continue;
      GpPlan p = gp_make(nbits, M, grp, K, sh[0], sh[1], forced_ks ? forced_ks : ks);
      // at least sixteen steps (1024 k) per split: below that the prologue and the parked tile cost more than the split saves
      if (!forced_ks && p.KS > 1 && (p.kps < 16 || nk / p.KS < 16)) continue;
      const double c = gp_cost(p, M, N, nk);
      if (c < best_cost) { best = p; best_cost = c; }
      // hybrid: more tiles than CUs and a partly filled last round -> the full rounds whole, the last round's tiles split `ks` ways
      const int64_t tiles = static_cast<int64_t>(p.n_tiles) * p.m_tiles;
      if (!forced_ks && !(opts & HQQ_OPT_GEMM_NOHYBRID) && p.KS > 1 && tiles > 256 && tiles % 256 != 0 && (tiles % 256) * p.KS <= 256) {
        p.full = static_cast<int>(tiles / 256 * 256);
        const double ch = gp_cost(p, M, N, nk);
        if (ch < best_cost) { best = p; best_cost = ch; }
      }
    }
  }
  return best;
}

size_t gemm_pipe_workspace_bytes_grouped(int nbits, int n_layers, const int64_t* N, int64_t M, int64_t K, uint32_t opts) {
  const GpPlan p = gp_plan(nbits, M, gp_group(N, n_layers), K, opts);
  if (p.KS <= 1) return 0;
  const int64_t tiles = static_cast<int64_t>(p.n_tiles) * p.m_tiles - p.full;
  return WS_COUNTER_BYTES + static_cast<size_t>(p.KS) * tiles * p.BM * (16 * p.NW) * gd_per(nbits) * sizeof(float);   // (the head stays zero: the decode kernels' arrival counters)
}

// Where this kernel beats "dequantise kernel + library GEMM" on MI355X (profiles/r02_prefill_sweep.md, Llama-2-7B shapes, int4): up to
// 640 rows everywhere (1.04-2.6x; below 512 the dequantise pass is as long as the GEMM), and up to 1024 rows when the plan fills the
// chip in one round (192..256 workgroups: o, down 1.0-1.15x; the other shapes are within +-6 % there and go to the library).  Beyond,
// the library's tile scheduler and hand-tuned loop are ahead (1.13-1.21 PFLOP/s here at 8192 rows against 1.23-1.45 for the composition).
// Against the other prefill route, hqq_hip_dequantize + hqq_hip_gemm_dense (rebuild the weights once, stream them as fp16): this kernel rebuilds
// every weight once per 256-token tile and is ahead while that is a few times — to ~2000 tokens on the 7B shapes, level at 3072, behind from
// 4096 (profiles/r04_prefill_routes_int4.txt; the dense kernel's 256 x 256 tiles also leave CUs idle below ~2000 tokens)
size_t gemm_pipe_workspace_bytes(int nbits, int64_t M, int64_t N, int64_t K, uint32_t opts) { return gemm_pipe_workspace_bytes_grouped(nbits, 1, &N, M, K, opts); }

bool gemm_pipe_wins(int nbits, int64_t M, int64_t N, int64_t K) {
  (void)nbits; (void)N; (void)K;
  return M <= 2560;
}

void gemm_pipe_describe(int nbits, int64_t M, int64_t N, int64_t K, uint32_t opts, int out[8]) {
  const GpPlan p = gp_plan(nbits, M, gp_group(&N, 1), K, opts);
  out[0] = p.NW; out[1] = p.BM; out[2] = p.n_tiles; out[3] = p.m_tiles; out[4] = p.KS; out[5] = p.kps; out[6] = p.full;
  out[7] = static_cast<int>(p.full + (static_cast<int64_t>(p.n_tiles) * p.m_tiles - p.full) * p.KS);
}

bool gemm_pipe_covers(int nbits, int64_t M, int64_t N, int64_t K, int64_t gs, int dtype) {
  if ((dtype != HQQ_F16 && dtype != HQQ_BF16) || (nbits != 8 && nbits != 4 && nbits != 3 && nbits != 2)) return false;   // (3: the stream layout — the caller checks HQQ_OPT_W3S)
  const int per = gd_per(nbits);
  // group_size 64 = one step: a step's weights share one (zero, scale) per row; K / 64 even: two steps' constants per DMA dword
  return N % per == 0 && (N / per) % 4 == 0 && gs == 64 && K % 128 == 0 && M >= 1;
}

template <int NBITS, bool SUB, int NW, int BM, bool BF>
static int gp_launch(const GdArgs& a, int64_t blocks, hipStream_t st) {
  using CF = GdCfg<NW, BM>;
  constexpr int lds_bytes = CF::DX * CF::XSTAGE + CF::DW * NW * 1024 + CF::DM * NW * GdMeta<NBITS>::SLOT;
  static bool done_on[64] = {};    // per device (the attribute belongs to the function ON a device); idempotent: a race sets it twice
  int devid = 0;
  (void)hipGetDevice(&devid);
  bool& attr_done = done_on[devid & 63];
  if (!attr_done) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pipe_f16_kernel<NBITS, SUB, NW, BM, BF>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) {
      set_error("hqq_hip_gemm: cannot reserve %d bytes of LDS: %s", lds_bytes, hipGetErrorString(e));
      return static_cast<int>(e);   // (positive: a HIP error, as check_launch reports them)
    }
    attr_done = true;
  }
  hipLaunchKernelGGL((gemm_pipe_f16_kernel<NBITS, SUB, NW, BM, BF>), dim3(static_cast<unsigned>(blocks)), dim3(64 * NW), lds_bytes, st, a);
  int rc = check_launch("hqq_hip_gemm(pipelined)");
  if (rc || a.KS <= 1) return rc;
  const int64_t rblocks = (static_cast<int64_t>(a.n_tiles) * a.m_tiles - a.full) * (gd_per(NBITS) * (BM / 16));
  hipLaunchKernelGGL((gemm_pipe_reduce_kernel<NBITS, NW, BM, BF>), dim3(static_cast<unsigned>(rblocks)), dim3(64 * NW), 0, st, a);
  return check_launch("hqq_hip_gemm(split-K reduce)");
}

int gemm_pipe_run_grouped(int nbits, int n_layers, const void* x, const void* const* Wq, const void* const* scale, const void* const* zero, const void* const* bias,
                          void* const* y, const int64_t* N, int64_t M, int64_t K, int64_t gs, int dtype, uint32_t opts, void* workspace, size_t workspace_bytes, hipStream_t st) {
  const GpGroup grp = gp_group(N, n_layers);
  const GpPlan p = gp_plan(nbits, M, grp, K, opts);
  const int64_t tiles = static_cast<int64_t>(p.n_tiles) * p.m_tiles;
  const int64_t blocks = p.full + (tiles - p.full) * p.KS;
  if (blocks * (gd_per(nbits) * (p.BM / 16)) > INT32_MAX) { set_error("hqq_hip_gemm: grid too large"); return HQQ_ERR_SHAPE; }
  GdArgs a;
  a.x = static_cast<const half_t*>(x);
  int64_t t_end = 0;
  for (int i = 0; i < GD_MAXL; ++i) {
    const int j = i < n_layers ? i : n_layers - 1;
    if (!aligned16(scale[j]) || !aligned16(zero[j])) { set_error("hqq_hip_gemm: scale / zero must be 16-byte aligned"); return HQQ_ERR_ALIGN; }
    a.Wq[i] = static_cast<const uint8_t*>(Wq[j]); a.scale[i] = static_cast<const half_t*>(scale[j]); a.zero[i] = static_cast<const half_t*>(zero[j]);
    a.bias[i] = bias ? static_cast<const half_t*>(bias[j]) : nullptr; a.y[i] = static_cast<half_t*>(y[j]); a.N[i] = static_cast<int>(N[j]);
    if (i < n_layers) t_end += (N[i] / gd_per(nbits) + 16 * p.NW - 1) / (16 * p.NW);
    a.tile_end[i] = static_cast<int>(t_end);
  }
  a.part = nullptr;
  a.M = static_cast<int>(M); a.K = static_cast<int>(K); a.G = static_cast<int>(K / gs);
  a.n_tiles = p.n_tiles; a.m_tiles = p.m_tiles; a.KS = p.KS; a.kps = p.kps; a.full = p.full;
  if (p.KS > 1) {
    const size_t need = gemm_pipe_workspace_bytes_grouped(nbits, n_layers, N, M, K, opts);
    if (!workspace || workspace_bytes < need || !aligned16(workspace)) {
      set_error("hqq_hip_gemm: workspace %zu < %zu bytes (hqq_hip_forward_workspace_bytes / hqq_hip_gemm_grouped_workspace_bytes)", workspace_bytes, need);
      return HQQ_ERR_WORKSPACE;
    }
    a.part = reinterpret_cast<float*>(static_cast<char*>(workspace) + WS_COUNTER_BYTES);
  }
  const bool sub = (opts & HQQ_OPT_META_SCALABLE) != 0 && dtype == HQQ_F16;
#define GP_GO3(NB, SB, BF_) (p.BM == 256 ? gp_launch<(NB == 2 ? 4 : NB), SB, 8, 256, BF_>(a, blocks, st) /* (never planned at 2 bits: 256 accumulator registers) */ : p.NW == 8 ? gp_launch<(NB == 2 ? 4 : NB), SB, 8, 128, BF_>(a, blocks, st) /* (never planned at 2 bits either: gp_plan) */ : gp_launch<NB, SB, 4, 128, BF_>(a, blocks, st))
#define GP_GO(NB) (dtype == HQQ_BF16 ? GP_GO3(NB, false, true) : sub ? GP_GO3(NB, true, false) : GP_GO3(NB, false, false))
  if (nbits == 8) return GP_GO(8);
  if (nbits == 4) return GP_GO(4);
  if (nbits == 3) return GP_GO(3);
  return GP_GO(2);
#undef GP_GO3
#undef GP_GO
}

int gemm_pipe_run(int nbits, const void* x, const void* Wq, const void* scale, const void* zero, const void* bias, void* y,
                  int64_t M, int64_t N, int64_t K, int64_t gs, int dtype, uint32_t opts, void* workspace, size_t workspace_bytes, hipStream_t st) {
  return gemm_pipe_run_grouped(nbits, 1, x, &Wq, &scale, &zero, bias ? &bias : nullptr, &y, &N, M, K, gs, dtype, opts, workspace, workspace_bytes, st);
}

}  // namespace hqq
