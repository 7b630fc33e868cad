// batch.hip — decode with a batch (5 <= M <= 64 activation rows): fused unpack -> dequantize -> skinny GEMM with NO K split across
// workgroups, fp16 / bf16, 4- / 2-bit, group_size 64, gfx950.  Round 4's answer to what skinny.hip measured about itself.
//
// Reference chain replaced (axis=1): BitPack.unpack_* -> (W_r - zero) * scale -> torch.matmul(x, W.t()) (+ bias)
//   hqq/core/bitpack.py:31-64, hqq/core/quantize.py:183-199, :880-898.  The weights multiplied are bit-identical to hqq_hip_dequantize /
//   Quantizer.dequantize (two roundings in the compute dtype); the fp32 summation order is fixed and depends on K only.
//
// What skinny.hip paid per launch above its streaming time: ~5 us until the first data (group constants of the
// whole K range staged before the loop), ~4.4 us of split-K finish (three trips to the device's coherence point), 64-row panels that
// leave 7B launches with 1.3 workgroups per CU, and x chunks requested from the same in-order queue as the HBM weight loads.
// This kernel:
//   tiles    a UNIT is 16 / PER packed rows (8 at 4 bits, 4 at 2): the 16 rows of one MFMA A tile are (unit row, slab) pairs — lane
//            i of a 16-lane row holds packed row i % (16 / PER), slab i / (16 / PER), shifts its own slab's field down to bit 0 and rebuilds 16
//            weights.  A workgroup owns HU <= 3 consecutive units over the WHOLE K: a 4096 x 4096 layer is 256 workgroups, q|k|v 256,
//            gate|up 459, down 256 — every launch of a 7B block fills the chip without cutting K, so there is no partial tile in HBM,
//            no arrival counter and no finishing pass.
//   waves    eight COMPUTE waves + one LOADER wave.  Compute wave w takes k-block (64 k = one group) w of every chunk of 8 blocks; it
//            brings its own x block [32 tokens x 64 k] into a wave-private LDS ring by LDS-DMA (full 128-byte lines, XOR-swizzled
//            source addresses as in gemm_pipe.hip) — L2 hits in a queue of their own.  The loader wave alone talks to HBM: the group
//            constants of the tile once, then the packed bytes of chunk c + D - 1 into a shared LDS ring (unique bytes only).  Memory
//            returns a wave's loads in order; with both kinds in one queue every x block waited one HBM latency (skinny.hip's
//            lesson).  One workgroup barrier per chunk hands a landed chunk from the loader to the compute waves.
//   x        every workgroup streams all of x through its CU: M K 2 bytes per tile against (16 / PER) HU K bytes of weights — at 32
//            rows and one unit that is 8x the weight bytes, from L2, at the CU's 64 B/clk; it buys the absence of a cross-CU
//            reduction (10.2 -> ~6 us for a 4096 x 4096 layer).  33..64 rows: the compute waves form two token halves of four
//            blocks per chunk, each half with the same per-wave code.
//   sums     a compute wave adds its blocks in ascending k (two accumulator sets in the two-half mode, one per chunk parity), the
//            eight block classes j % 8 meet in LDS and are added in class order: the association depends on K alone — not on M, HU
//            or what else was in the launch — so a row's bits do not depend on the batch or the group it was computed in.
#include "decode_common.h"

#ifdef BT_LAB_TS
extern unsigned long long* g_bt_lab_ts;
#endif
namespace hqq {
namespace bt {

constexpr int BT_MAXL = HQQ_GEMV_MAX_GROUP;
constexpr int BT_CW = 8;              // compute waves
constexpr int BT_T = (BT_CW + 1) * 64;
constexpr int BT_HUMAX = 3;           // units per workgroup
#ifndef BT_R_DEPTH
#define BT_R_DEPTH 2
#endif
constexpr int BT_R = BT_R_DEPTH;      // x blocks in flight per compute wave
#ifndef BT_WARM_DIV
#define BT_WARM_DIV 0                 // lab: every workgroup touches 1 / BT_WARM_DIV of x's lines at its start (0: no warm-up — measured: no gain, the chunk time is not x latency)
#endif
constexpr int BT_DUMMY = 1024;        // bytes of LDS that the warm-up's loads land in (never read)
constexpr int BT_RED_CLASSES = 8;

struct BtArgs {
  const uint8_t* Wq[BT_MAXL];
  const half_t* scale[BT_MAXL];
  const half_t* zero[BT_MAXL];
  const half_t* bias[BT_MAXL];
  half_t* y[BT_MAXL];
  int N[BT_MAXL];
  int unit_end[BT_MAXL];    // end (exclusive) of layer i's units in the concatenated unit space (unused entries repeat the last)
  const half_t* x;
  int M, K, G, total_units, HU, MH, RS;   // MH: token halves (1: M <= 32, 2: 33..64); RS: bytes per row of the group constants in LDS
#ifdef BT_LAB_TS
  unsigned long long* ts;   // lab only (tools/batch_lab.hip): eight time stamps per wave
#endif
};
#ifdef BT_LAB_TS
#define BT_TS(i) do { if (t_[i] == 0) t_[i] = __builtin_amdgcn_s_memrealtime(); } while (0)   /* 100 MHz, one clock for the whole device; first occurrence */
#else
#define BT_TS(i)
#endif

typedef __attribute__((address_space(3))) void* bt_lds_t;
typedef const __attribute__((address_space(1))) void* bt_glb_t;
__device__ __forceinline__ void bt_dma16(const void* src, uint8_t* lds_wave_base) { __builtin_amdgcn_global_load_lds((bt_glb_t)src, (bt_lds_t)lds_wave_base, 16, 0, 0); }
__device__ __forceinline__ void bt_dma16_nt(const void* src, uint8_t* lds_wave_base) { __builtin_amdgcn_global_load_lds((bt_glb_t)src, (bt_lds_t)lds_wave_base, 16, 0, 2); }   // streamed once
__device__ __forceinline__ void bt_dma4(const void* src, uint8_t* lds_wave_base) { __builtin_amdgcn_global_load_lds((bt_glb_t)src, (bt_lds_t)lds_wave_base, 4, 0, 0); }
// position of a 16-byte chunk inside a 128-byte row of an x block: chunk ^ bt_swz(row) (gemm_pipe.hip's map: conflict-free B-fragment reads)
__device__ __forceinline__ int bt_swz(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 2); }

struct BtUnit {   // wave-uniform: where a unit of the workgroup lives
  const uint8_t* Wq;
  const half_t* scale;
  const half_t* zero;
  const half_t* bias;
  half_t* y;
  int N, prow0;   // out_features of its layer; first packed row of the unit inside the layer
};
template <int PER>
__device__ __forceinline__ BtUnit bt_unit(const BtArgs& a, int U) {
  BtUnit c{a.Wq[0], a.scale[0], a.zero[0], a.bias[0], a.y[0], a.N[0], 0};
  int u0 = 0;
#pragma unroll
  for (int i = 1; i < BT_MAXL; ++i) {
    const bool in = U >= a.unit_end[i - 1];
    c.Wq = pick(in, a.Wq[i], c.Wq);
    c.scale = pick(in, a.scale[i], c.scale);
    c.zero = pick(in, a.zero[i], c.zero);
    c.bias = pick(in, a.bias[i], c.bias);
    c.y = pick(in, a.y[i], c.y);
    c.N = pick(in, a.N[i], c.N);
    u0 = pick(in, a.unit_end[i - 1], u0);
  }
  c.prow0 = (U - u0) * (16 / PER);
  return c;
}

// the lane's 16 weights of ITS slab (already shifted down to bit 0 of every byte), exactly as Quantizer.dequantize rebuilds them, as MFMA A
// operands in natural k order: a0 = k 0..7 of the lane's 16, a1 = k 8..15
template <int NBITS>
__device__ __forceinline__ void bt_rebuild_f16(const u32x4& w, half_t z, half_t s, u32x4& a0, u32x4& a1) {
  constexpr uint32_t m1 = (NBITS == 8) ? 0xFFu : ((1u << NBITS) - 1u);
  constexpr uint32_t m = m1 | (m1 << 16);
  const half2_t k2 = {static_cast<half_t>(-1024.0f), static_cast<half_t>(-1024.0f)};
  const half2_t zz = {z, z}, ss = {s, s};
  half2_t q[8];
  uint32_t o[8];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const uint32_t v = __builtin_amdgcn_perm(w[d], w[d], 0x03010200u);   // bytes (b0,b1,b2,b3) -> (b0,b2,b1,b3): pairs come out in natural k order
    q[2 * d] = as_h2((v & m) | 0x64006400u);              // (k 4d, 4d+1): 1024 + level
    q[2 * d + 1] = as_h2(((v >> 8) & m) | 0x64006400u);   // (k 4d+2, 4d+3)
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) q[i] = q[i] + k2;                                // exact integer level
#pragma unroll
  for (int i = 0; i < 8; ++i) q[i] = q[i] - zz;                                // rounding 1
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = __builtin_bit_cast(uint32_t, q[i] * ss);  // rounding 2
  a0 = u32x4{o[0], o[1], o[2], o[3]};
  a1 = u32x4{o[4], o[5], o[6], o[7]};
}
// bf16 (gemm_pipe.hip's GdSlabBF for a field at bit 0): through fp32 — q - z with ONE rounding, v_cvt_pk_bf16_f32, exact product with s by
// v_dot2_f32_bf16 against (s, 0) / (0, s), a second v_cvt_pk
typedef __bf16 bt_bf2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bt_bf8_t __attribute__((ext_vector_type(8)));
typedef float bt_f2_t __attribute__((ext_vector_type(2)));
template <int B> __device__ __forceinline__ float bt_ubyte(uint32_t v) { return static_cast<float>((v >> (8 * B)) & 0xFFu); }   // v_cvt_f32_ubyteB
template <int NBITS>
__device__ __forceinline__ void bt_rebuild_bf16(const u32x4& w, uint16_t z, uint16_t s, u32x4& a0, u32x4& a1) {
  constexpr uint32_t m1 = (NBITS == 8) ? 0xFFu : ((1u << NBITS) - 1u);
  const float zf = __uint_as_float(static_cast<uint32_t>(z) << 16);
  const bt_bf2_t s_lo = __builtin_bit_cast(bt_bf2_t, static_cast<uint32_t>(s));          // (s, 0)
  const bt_bf2_t s_hi = __builtin_bit_cast(bt_bf2_t, static_cast<uint32_t>(s) << 16);    // (0, s)
  uint32_t o[8];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const uint32_t fq = w[d] & (m1 * 0x01010101u);
    const bt_f2_t dq[2] = {{bt_ubyte<0>(fq) - zf, bt_ubyte<1>(fq) - zf}, {bt_ubyte<2>(fq) - zf, bt_ubyte<3>(fq) - zf}};   // q - z: one fp32 rounding
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const bt_bf2_t dr = __builtin_convertvector(dq[h], bt_bf2_t);                 // rounding 1
      const bt_f2_t pw = {__builtin_amdgcn_fdot2_f32_bf16(dr, s_lo, 0.f, false), __builtin_amdgcn_fdot2_f32_bf16(dr, s_hi, 0.f, false)};
      o[2 * d + h] = __builtin_bit_cast(uint32_t, __builtin_convertvector(pw, bt_bf2_t));   // rounding 2
    }
  }
  a0 = u32x4{o[0], o[1], o[2], o[3]};
  a1 = u32x4{o[4], o[5], o[6], o[7]};
}

// ---- the loader wave: the packed chunks, D - 1 chunks ahead of the compute waves; nothing else in its queue ----------------------------
// LDS weight slot of a chunk: [block b][unit u][16 / PER rows][64 bytes] — (b HU + u) UB bytes in; one DMA instruction = PER (block, unit) pairs.
template <int NBITS, int HU, int D>
__device__ __forceinline__ void bt_loader(const BtArgs& a, const BtUnit (&un)[BT_HUMAX], uint8_t* wring, int lane, int nchunks, int BPC, unsigned long long* t_) {
  constexpr int PER = 8 / NBITS, UR = 16 / PER, UB = UR * 64, LPP = 64 / PER;   // rows per unit, bytes per (unit, block), lanes per pair
  const int K = a.K, nblocks = K / 64;
  const int chw = BPC * HU * UB;          // bytes per chunk slot
  const int nW = chw / 1024;              // DMA instructions per chunk (BPC HU / PER; an integer for BPC = 8 or 4)
  // ---- per-lane source of every DMA instruction of a chunk (block 0 of chunk 0) ----
  constexpr int NWMAX = HU * 8 / PER;        // nW at 8 blocks per chunk (half of it at 4)
  const uint8_t* wsrc[NWMAX];
  int wblk[NWMAX];
#pragma unroll
  for (int d = 0; d < NWMAX; ++d) {
    const int q = d * PER + lane / LPP;            // (block, unit) pair of this lane
    const int qc = q < BPC * HU ? q : BPC * HU - 1;
    const int b = qc / HU, u = qc - b * HU, ll = lane % LPP, row = ll >> 2, col = ll & 3;
    const uint8_t* base = pick(u == 0, un[0].Wq, pick(u == 1, un[1].Wq, un[2].Wq));
    const int prow0 = pick(u == 0, un[0].prow0, pick(u == 1, un[1].prow0, un[2].prow0));
    wsrc[d] = base + static_cast<int64_t>(prow0 + row) * K + col * 16;
    wblk[d] = b;
  }
  auto issue_w = [&](int c) {   // c < nchunks
#pragma unroll
    for (int d = 0; d < NWMAX; ++d) {
      if (d < nW) {   // (wave-uniform)
        int j = c * BPC + wblk[d];
        j = j < nblocks ? j : nblocks - 1;       // K not a multiple of the chunk: the row's last block again, never consumed
        bt_dma16_nt(wsrc[d] + static_cast<int64_t>(j) * 64, wring + (c % D) * chw + d * 1024);
      }
    }
  };
#pragma unroll
  for (int c = 0; c < D - 1; ++c)
    if (c < nchunks) issue_w(c);
  BT_TS(2);
  for (int c = 0; c < nchunks; ++c) {
    // everything up to chunk c has landed: the instructions issued after it are those of chunks c + 1 .. c + D - 2 (near the end of the row
    // fewer are behind it: wait for all of them)
#ifdef BT_LAB_FINE
    if (c == 2) { t_[1] = __builtin_amdgcn_s_memrealtime(); }
#endif
    if (c + D - 2 < nchunks) {
      if (BPC == 8) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 2) * (8 * HU / PER)) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 2) * (4 * HU / PER > 0 ? 4 * HU / PER : 1)) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#ifdef BT_LAB_FINE
    if (c == 2) { t_[2] = __builtin_amdgcn_s_memrealtime(); }
#endif
    __builtin_amdgcn_s_barrier();              // b_c: chunk c is in LDS for everyone; the compute waves have finished reading chunk c - 1
#ifdef BT_LAB_FINE
    if (c == 2) { t_[3] = __builtin_amdgcn_s_memrealtime(); }
#elif defined(BT_LAB_TS)
    if (c == 0) BT_TS(3); else if (c == 1) BT_TS(4); else if (c == nchunks - 1) BT_TS(5);
#endif
    if (c + D - 1 < nchunks) issue_w(c + D - 1);   // into the slot of chunk c - 1
#ifdef BT_LAB_FINE
    if (c == 2) { t_[5] = __builtin_amdgcn_s_memrealtime(); }
    if (c == 3) { t_[6] = __builtin_amdgcn_s_memrealtime(); }
#endif
  }
  BT_TS(6);
}

template <int NBITS, int MTW, bool BF>
__global__ __launch_bounds__(BT_T, 1) void batch_f16_kernel(const BtArgs a) {
  constexpr int PER = 8 / NBITS, UR = 16 / PER, UB = UR * 64;
  constexpr int XS = MTW * 2048;                       // bytes of one x block of a compute wave: 16 MTW token rows x 128 bytes
  extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = a.K, M = a.M, HU = a.HU, MH = a.MH, RS = a.RS;
  const int BPC = BT_CW / MH;                          // blocks per chunk: 8, or 4 with two token halves
  const int nblocks = K / 64, nchunks = (nblocks + BPC - 1) / BPC;
  const int D = HU == 1 ? 8 : (HU == 2 ? 5 : 4);       // weight chunks in the ring
  const int chw = BPC * HU * UB;
  uint8_t* const xring = lds;                                        // [BT_CW][BT_R][XS]
  uint8_t* const wring = lds + BT_CW * BT_R * XS;                    // [D][chw]
  uint8_t* const zmeta = wring + D * chw;                            // [HU 16 rows][RS]
  uint8_t* const smeta = zmeta + HU * 16 * RS;
  uint8_t* const dummy = smeta + HU * 16 * RS;                       // [BT_DUMMY] landing place of the warm-up loads
  float* const red = reinterpret_cast<float*>(lds);                  // after the loop: [MH][8 classes][HU][MTW][64 lanes] x 4 floats

#ifdef BT_LAB_TS
  unsigned long long t_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  BT_TS(0);
#else
  unsigned long long* t_ = nullptr;
#endif
  BtUnit un[BT_HUMAX];
#pragma unroll
  for (int u = 0; u < BT_HUMAX; ++u) {
    int U = static_cast<int>(blockIdx.x) * HU + (u < HU ? u : HU - 1);
    U = U < a.total_units ? U : a.total_units - 1;     // the last workgroup's spare units repeat the last one (never stored)
    un[u] = bt_unit<PER>(a, U);
  }

  BT_TS(1);
  // (the loader is the workgroup's FIRST wave: the last one starts ~0.9 us after the first — measured — and the first packed chunk is what everybody waits for)
  if (wave == 0) {
    // ================================ loader wave ================================
    if (HU == 1) bt_loader<NBITS, 1, 8>(a, un, wring, lane, nchunks, BPC, t_);
    else if (HU == 2) bt_loader<NBITS, 2, 5>(a, un, wring, lane, nchunks, BPC, t_);
    else bt_loader<NBITS, 3, 4>(a, un, wring, lane, nchunks, BPC, t_);
    __builtin_amdgcn_s_barrier();   // e1 (below)
    __builtin_amdgcn_s_barrier();   // e2
  } else {
    // ================================ compute waves ================================
    const int r = lane & 15, c4 = lane >> 4;
    const int cw = wave - 1;                              // compute wave index 0..7
    const int b = cw % BPC, mh = cw / BPC, m0 = mh * 32;
    const int prow = r % UR, slab = r / UR;
    const uint32_t shl = static_cast<uint32_t>(NBITS * (PER - 1 - slab));   // the lane's slab field -> bit 0 of every byte
    uint8_t* const xmine = xring + cw * BT_R * XS;
    // x pieces: piece p fills token rows 8 p .. 8 p + 7 of the block; lane -> (row, 16-byte position); the chunk a position holds is XOR-ed
    const half_t* xsrc[2 * MTW];
#pragma unroll
    for (int p = 0; p < 2 * MTW; ++p) {
      const int row = 8 * p + (lane >> 3), pos = lane & 7, ch = pos ^ bt_swz(row);
      const int tok = m0 + row;
      xsrc[p] = a.x + static_cast<int64_t>(tok < M ? tok : 0) * K + ch * 8;   // token rows past M read row 0: their columns are never stored
    }
    auto issue_x = [&](int c) {   // c < nchunks
      int j = c * BPC + b;
      j = j < nblocks ? j : nblocks - 1;
#pragma unroll
      for (int p = 0; p < 2 * MTW; ++p) bt_dma16(xsrc[p] + static_cast<int64_t>(j) * 64, xmine + (c % BT_R) * XS + p * 1024);
    };
    f32x4 acc[2][BT_HUMAX][MTW];   // [chunk parity (two-half mode only)][unit][token tile]
#pragma unroll
    for (int pa = 0; pa < 2; ++pa)
#pragma unroll
      for (int u = 0; u < BT_HUMAX; ++u)
#pragma unroll
        for (int t = 0; t < MTW; ++t) acc[pa][u][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto mfma = [&](const u32x4& A, const u32x4& B, f32x4 C) {
      if constexpr (BF) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bt_bf8_t, A), __builtin_bit_cast(bt_bf8_t, B), C, 0, 0, 0);
      else return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_t, A), __builtin_bit_cast(h8_t, B), C, 0, 0, 0);
    };
    // One chunk, in the order that keeps the chain behind the barrier short (measured with in-kernel stamps: reads -> rebuild -> MFMA ->
    // DMA issue behind the barrier made a chunk 0.7 us whatever M): in FRONT of the barrier — this wave's x block has landed (its own queue),
    // its B fragments go to registers, and the DMA of the block two chunks ahead is issued into the slot just read; BEHIND it only the packed
    // bytes -> rebuild -> MFMA.  The group constants of a chunk are read one chunk early (they are all in LDS from barrier 0 on).
    uint32_t zs_next[BT_HUMAX];   // (zero | scale << 16) of the NEXT chunk's group, per unit
    auto read_meta = [&](int c) {
      int j = c * BPC + b;
      j = j < nblocks ? j : nblocks - 1;
#pragma unroll
      for (int u = 0; u < BT_HUMAX; ++u)
        if (u < HU) {
          const uint32_t zb = *reinterpret_cast<const uint16_t*>(zmeta + (u * 16 + r) * RS + j * 2);
          const uint32_t sb = *reinterpret_cast<const uint16_t*>(smeta + (u * 16 + r) * RS + j * 2);
          zs_next[u] = zb | (sb << 16);
        }
    };
    auto chunk = [&](int cc, auto parity) {
      constexpr int pa = decltype(parity)::value;
      // ---- in front of the barrier ----
      // x block cc has landed (this wave's own pieces; near the end of the row fewer blocks are behind it: wait for all)
      if (cc + BT_R <= nchunks) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((BT_R - 1) * 2 * MTW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const uint8_t* xs = xmine + (cc % BT_R) * XS;
      u32x4 f0[MTW], f1[MTW];
#pragma unroll
      for (int t = 0; t < MTW; ++t) {
        const int row = 16 * t + r;
        f0[t] = *reinterpret_cast<const u32x4*>(xs + row * 128 + (((2 * c4) ^ bt_swz(row)) << 4));
        f1[t] = *reinterpret_cast<const u32x4*>(xs + row * 128 + (((2 * c4 + 1) ^ bt_swz(row)) << 4));
      }
      uint32_t zs[BT_HUMAX];
#pragma unroll
      for (int u = 0; u < BT_HUMAX; ++u) zs[u] = zs_next[u];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the fragment reads have left the LDS before the DMA below may overwrite the block
      __builtin_amdgcn_sched_barrier(0);
      if (cc + BT_R < nchunks) issue_x(cc + BT_R);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();                          // b_cc: the loader's chunk cc has landed
#ifdef BT_LAB_TS
      if (cc == 0) BT_TS(3); else if (cc == 1) BT_TS(4); else if (cc == nchunks - 1) BT_TS(5);
#endif
      __builtin_amdgcn_sched_barrier(0);
      // ---- behind the barrier ----
      if (cc == 0) {   // (the constants became everybody's at this barrier: chunk 0 reads its own here)
        read_meta(0);
#pragma unroll
        for (int u = 0; u < BT_HUMAX; ++u) zs[u] = zs_next[u];
      }
      if (cc * BPC + b < nblocks) {
        const uint8_t* ws = wring + (cc % D) * chw + b * HU * UB + prow * 64 + c4 * 16;
        u32x4 raw[BT_HUMAX];
#pragma unroll
        for (int u = 0; u < BT_HUMAX; ++u)
          if (u < HU) raw[u] = *reinterpret_cast<const u32x4*>(ws + u * UB);
        if (cc + 1 < nchunks) read_meta(cc + 1);
#pragma unroll
        for (int u = 0; u < BT_HUMAX; ++u) {
          if (u < HU) {   // (wave-uniform)
            const u32x4 w = u32x4{raw[u][0] >> shl, raw[u][1] >> shl, raw[u][2] >> shl, raw[u][3] >> shl};
            u32x4 a0, a1;
            if constexpr (BF) bt_rebuild_bf16<NBITS>(w, static_cast<uint16_t>(zs[u]), static_cast<uint16_t>(zs[u] >> 16), a0, a1);
            else bt_rebuild_f16<NBITS>(w, __builtin_bit_cast(half_t, static_cast<uint16_t>(zs[u])), __builtin_bit_cast(half_t, static_cast<uint16_t>(zs[u] >> 16)), a0, a1);
#pragma unroll
            for (int t = 0; t < MTW; ++t) {
              acc[pa][u][t] = mfma(a0, f0[t], acc[pa][u][t]);
              acc[pa][u][t] = mfma(a1, f1[t], acc[pa][u][t]);
            }
          }
        }
      } else if (cc + 1 < nchunks) {
        read_meta(cc + 1);
      }
    };
    // ---- group constants of the tile: rows (u, i) = (unit, A-tile row), 128-byte half-slots hs = row SPR + seg, dealt over the compute
    //      waves.  They sit in front of the wave's first x block in its queue, so every wave has its share in LDS before it arrives at
    //      barrier 0 — after which every wave may read all of them.  16 bytes per lane where a row's constants are whole 16-byte pieces ----
    {
      const int G = a.G, SPR = RS / 128, nhs = HU * 16 * SPR;
      const bool wide = (2 * G) % 16 == 0;
      const int per_instr = wide ? 8 : 2;                        // half-slots one DMA instruction fills
      const int n_instr = (nhs + per_instr - 1) / per_instr;     // per tensor
      for (int q = cw; q < 2 * n_instr; q += BT_CW) {
        const bool is_s = q >= n_instr;
        const int hs0 = (is_s ? q - n_instr : q) * per_instr;
        int hs = hs0 + (wide ? lane >> 3 : lane >> 5);
        hs = hs < nhs ? hs : nhs - 1;
        const int row = hs / SPR, seg = hs - row * SPR, u = row >> 4, i = row & 15;
        const int prow0 = pick(u == 0, un[0].prow0, pick(u == 1, un[1].prow0, un[2].prow0));
        const int Nl = pick(u == 0, un[0].N, pick(u == 1, un[1].N, un[2].N));
        const half_t* zb = pick(u == 0, un[0].zero, pick(u == 1, un[1].zero, un[2].zero));
        const half_t* sb = pick(u == 0, un[0].scale, pick(u == 1, un[1].scale, un[2].scale));
        const int64_t n = prow0 + (i % UR) + static_cast<int64_t>(i / UR) * (Nl / PER);
        const uint8_t* rowp = reinterpret_cast<const uint8_t*>((is_s ? sb : zb) + n * G);
        uint8_t* dst = (is_s ? smeta : zmeta) + hs0 * 128;
        if (wide) {
          int off = seg * 128 + (lane & 7) * 16;
          off = off < 2 * G - 16 ? off : 2 * G - 16;            // past the row's constants: its last piece again (lands in the padding)
          bt_dma16(rowp + off, dst);
        } else {
          int off = seg * 128 + (lane & 31) * 4;
          off = off < 2 * G - 4 ? off : 2 * G - 4;
          bt_dma4(rowp + off, dst);
        }
      }
    }
    if (BT_WARM_DIV > 0) {   // lab (see BT_WARM_DIV)
      const int xlines = M * (K / 64);                          // 128-byte lines of x
      const int phase = (static_cast<int>(blockIdx.x) >> 3) % (BT_WARM_DIV > 0 ? BT_WARM_DIV : 1);
      for (int t = cw * BT_WARM_DIV + phase; t * 64 < xlines; t += BT_CW * BT_WARM_DIV) {
        int line = t * 64 + lane;
        line = line < xlines ? line : xlines - 1;
        bt_dma4(a.x + static_cast<int64_t>(line) * 64, dummy);
      }
    }
    // ---- main loop: one chunk per iteration; x blocks BT_R ahead in the wave's own queue ----
#pragma unroll
    for (int c = 0; c < BT_R; ++c)
      if (c < nchunks) issue_x(c);
    BT_TS(2);
#pragma unroll
    for (int u = 0; u < BT_HUMAX; ++u) zs_next[u] = 0u;
    for (int c = 0; c < nchunks; c += 2) {
      chunk(c, std::integral_constant<int, 0>{});
      if (c + 1 < nchunks) {   // (wave-uniform; the barrier count is the same for every wave: nchunks)
        if (MH == 2) chunk(c + 1, std::integral_constant<int, 1>{});
        else chunk(c + 1, std::integral_constant<int, 0>{});
      }
    }
    BT_TS(6);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // e1: every ring is dead (all waves past their last read, every DMA landed): LDS becomes the reduction buffer
    // ---- block classes meet in LDS: class = k-block index % 8 (one-half mode: the wave; two-half mode: block + 4 x chunk parity) ----
#pragma unroll
    for (int pa = 0; pa < 2; ++pa) {
      if (pa == 0 || MH == 2) {
        const int cls = MH == 1 ? cw : b + 4 * pa;
#pragma unroll
        for (int u = 0; u < BT_HUMAX; ++u)
          if (u < HU)
#pragma unroll
            for (int t = 0; t < MTW; ++t)
              *reinterpret_cast<f32x4*>(red + ((((mh * BT_RED_CLASSES + cls) * HU + u) * MTW + t) * 64 + lane) * 4) = acc[pa][u][t];
      }
    }
    __builtin_amdgcn_s_barrier();   // e2
  }
  // ---- every thread: sum the eight classes of its items in class order, round once, + bias, store.  Item = (token half, unit, token tile,
  //      lane of the accumulator layout): lane (column r = token, rows 4 c + i = A-tile rows = (unit row, slab) pairs) ----
  const int items = MH * HU * MTW * 64;
  for (int it = tid; it < items; it += BT_T) {
    const int ln = it & 63, t = (it >> 6) % MTW, u = ((it >> 6) / MTW) % HU, mh = (it >> 6) / (MTW * HU);
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cls = 0; cls < BT_RED_CLASSES; ++cls) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(red + ((((mh * BT_RED_CLASSES + cls) * HU + u) * MTW + t) * 64 + ln) * 4);
      sum[0] += v[0]; sum[1] += v[1]; sum[2] += v[2]; sum[3] += v[3];
    }
    const int U = static_cast<int>(blockIdx.x) * HU + u;
    const int m = mh * 32 + t * 16 + (ln & 15);
    if (U >= a.total_units || m >= M) continue;
    struct { const half_t* bias; half_t* y; int N, prow0; } c;   // (field by field: a select between structs goes through memory)
    c.bias = pick(u == 0, un[0].bias, pick(u == 1, un[1].bias, un[2].bias));
    c.y = pick(u == 0, un[0].y, pick(u == 1, un[1].y, un[2].y));
    c.N = pick(u == 0, un[0].N, pick(u == 1, un[1].N, un[2].N));
    c.prow0 = pick(u == 0, un[0].prow0, pick(u == 1, un[1].prow0, un[2].prow0));
    const int i0 = 4 * (ln >> 4);                          // A-tile rows i0 .. i0 + 3: one slab, four consecutive unit rows (UR >= 4)
    const int n = c.prow0 + (i0 % UR) + (i0 / UR) * (c.N / PER);
    uint16_t o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (BF) {
        uint16_t v = f32_to_bf16(sum[i]);
        if (c.bias) v = f32_to_bf16(bf16_to_f32(v) + bf16_to_f32(reinterpret_cast<const uint16_t*>(c.bias)[n + i]));
        o[i] = v;
      } else {
        half_t v = static_cast<half_t>(sum[i]);
        if (c.bias) v = v + c.bias[n + i];   // `out += bias` on the rounded matmul result (quantize.py:896-897)
        o[i] = __builtin_bit_cast(uint16_t, v);
      }
    }
    *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(c.y) + static_cast<int64_t>(m) * c.N + n) = *reinterpret_cast<u32x2*>(o);
  }
#ifdef BT_LAB_TS
  BT_TS(7);
  if (lane == 0 && a.ts) { const int w_ = static_cast<int>(blockIdx.x) * (BT_CW + 1) + wave; for (int q = 0; q < 8; ++q) a.ts[w_ * 8 + q] = t_[q]; }
#endif
}

// ---- host side ----------------------------------------------------------------------------------------------------------------
static int bt_num_cus() {
  static int n_cus = 0;
  if (n_cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) n_cus = n;
    else n_cus = 256;
  }
  return n_cus;
}
static size_t bt_lds_bytes(int nbits, int mtw, int hu, int mh, int64_t K) {
  const int per = 8 / nbits, ub = (16 / per) * 64, bpc = BT_CW / mh, d = hu == 1 ? 8 : (hu == 2 ? 5 : 4);
  const size_t rs = static_cast<size_t>((2 * (K / 64) + 127) / 128) * 128;
  const size_t rings = static_cast<size_t>(BT_CW) * BT_R * mtw * 2048 + static_cast<size_t>(d) * bpc * hu * ub + 2 * static_cast<size_t>(hu) * 16 * rs + BT_DUMMY;
  const size_t red = static_cast<size_t>(mh) * BT_RED_CLASSES * hu * mtw * 1024;
  return rings > red ? rings : red;
}

bool batch_covers(int nbits, int64_t M, int64_t K, int64_t group_size, const int64_t* N, int n_layers, int dtype) {
  if ((nbits != 4 && nbits != 2) || group_size != 64 || M < 5 || M > 64 || K % 64 != 0 || K < 512 || K > (1 << 20)) return false;
  if (dtype != HQQ_F16 && dtype != HQQ_BF16) return false;
  for (int i = 0; i < n_layers; ++i)
    if (N[i] <= 0 || N[i] % 16 != 0) return false;   // whole units: N / PER a multiple of 16 / PER
  return bt_lds_bytes(nbits, M > 16 ? 2 : 1, 1, M > 32 ? 2 : 1, K) <= 160 * 1024;
}

// units per workgroup: as many as it takes to put the launch on the chip in one round, within BT_HUMAX and the LDS budget (shapes only)
static int bt_choose_hu(int nbits, int mtw, int mh, int64_t K, int64_t units) {
  int hu = static_cast<int>((units + bt_num_cus() - 1) / bt_num_cus());
  hu = hu > BT_HUMAX ? BT_HUMAX : (hu < 1 ? 1 : hu);
  while (hu > 1 && bt_lds_bytes(nbits, mtw, hu, mh, K) > 160 * 1024) --hu;
  return hu;
}

// is the launch big enough for this kernel to be the better one?  (no K split: a launch of few units leaves CUs idle — skinny.hip cuts K there)
bool batch_prefers(int nbits, int n_layers, const int64_t* N) {
  const int per = 8 / nbits;
  int64_t units = 0;
  for (int i = 0; i < n_layers; ++i) units += N[i] / 16;   // (N / per) / (16 / per)
  (void)per;
  return units >= bt_num_cus() / 2;
}

template <int NBITS, int MTW, bool BF>
static int bt_launch(const BtArgs& a, size_t lds, unsigned grid, hipStream_t st) {
  auto kern = batch_f16_kernel<NBITS, MTW, BF>;
  if (lds > 64 * 1024) {
    static LdsRaised raised;
    if (const int rc = raise_lds_limit(raised, reinterpret_cast<const void*>(kern), 160 * 1024, "hqq_hip_gemv")) return rc;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(BT_T), lds, st, a);
  return check_launch("hqq_hip_gemv");
}

int batch_run(int nbits, int n_layers, const void* x, const void* const* Wq, const void* const* scale, const void* const* zero, const void* const* bias,
              void* const* y, const int64_t* N, int64_t M, int64_t K, int dtype, hipStream_t st) {
  BtArgs a;
  int64_t units = 0;
  for (int i = 0; i < n_layers; ++i) {
    units += N[i] / 16;
    if (units > INT32_MAX / 4 || N[i] * (K / 64) > INT32_MAX) { set_error("hqq_hip_gemv: size overflow"); return HQQ_ERR_SHAPE; }
    a.Wq[i] = static_cast<const uint8_t*>(Wq[i]);
    a.scale[i] = static_cast<const half_t*>(scale[i]);
    a.zero[i] = static_cast<const half_t*>(zero[i]);
    a.bias[i] = bias ? static_cast<const half_t*>(bias[i]) : nullptr;
    a.y[i] = static_cast<half_t*>(y[i]);
    a.N[i] = static_cast<int>(N[i]);
    a.unit_end[i] = static_cast<int>(units);
  }
  for (int i = n_layers; i < BT_MAXL; ++i) {
    a.Wq[i] = a.Wq[n_layers - 1]; a.scale[i] = a.scale[n_layers - 1]; a.zero[i] = a.zero[n_layers - 1]; a.bias[i] = a.bias[n_layers - 1];
    a.y[i] = a.y[n_layers - 1]; a.N[i] = a.N[n_layers - 1]; a.unit_end[i] = a.unit_end[n_layers - 1];
  }
  a.x = static_cast<const half_t*>(x);
  a.M = static_cast<int>(M);
  a.K = static_cast<int>(K);
  a.G = static_cast<int>(K / 64);
  a.total_units = static_cast<int>(units);
  a.MH = M > 32 ? 2 : 1;
  const int mt = static_cast<int>((M + 15) / 16);
  const int mtw = a.MH == 2 ? 2 : mt;                 // token tiles per compute wave (two halves of 32 tokens beyond 32 rows)
  a.HU = bt_choose_hu(nbits, mtw, a.MH, K, units);
  a.RS = static_cast<int>((2 * (K / 64) + 127) / 128) * 128;
#ifdef BT_LAB_TS
  a.ts = g_bt_lab_ts;
#endif
  const size_t lds = bt_lds_bytes(nbits, mtw, a.HU, a.MH, K);
  const unsigned grid = static_cast<unsigned>((units + a.HU - 1) / a.HU);
  const bool bf = dtype == HQQ_BF16;
#define BT_GO(NB) (mtw == 1 ? (bf ? bt_launch<NB, 1, true>(a, lds, grid, st) : bt_launch<NB, 1, false>(a, lds, grid, st)) \
                            : (bf ? bt_launch<NB, 2, true>(a, lds, grid, st) : bt_launch<NB, 2, false>(a, lds, grid, st)))
  return nbits == 4 ? BT_GO(4) : BT_GO(2);
#undef BT_GO
}

}  // namespace bt
}  // namespace hqq
