// gemv_shared.h — what the decode launch kernels (gemv.hip: one launch per layer group; gemv_w3s.hip / gemv_block.hip: the same text at 3 bits / with the decoder block's glue folded in) have in common: launch constants, the kernel-argument layout, the layer pick, the unit a wave keeps in flight.
#pragma once
#include "decode_common.h"

namespace hqq {

#ifndef GV_WAVES_PER_WG
#define GV_WAVES_PER_WG 4
#endif
#ifndef GV_WG_PER_CU
#define GV_WG_PER_CU 4
#endif
constexpr int GV_WAVES = GV_WAVES_PER_WG;  // waves per workgroup (256 threads; 4 workgroups per CU measured best)
constexpr int GV_KSTEP = 1024;            // k covered by one wave load instruction: 64 lanes x 16 bytes
#ifndef GV_U_LOADS
#define GV_U_LOADS 2
#endif
#ifndef GV_NF_UNITS
#define GV_NF_UNITS 2
#endif
constexpr int GV_U = GV_U_LOADS;          // load instructions per unit
constexpr int GV_NF = GV_NF_UNITS;        // units a wave keeps in flight (the streaming loop is a ring of GV_NF register sets)
constexpr int GV_UNIT = GV_KSTEP * GV_U;  // k per unit
constexpr int GV_MAXL = HQQ_GEMV_MAX_GROUP;
constexpr int GV_LDS_MAX = 144 * 1024;    // x staging budget per workgroup
constexpr int GV_EXACT_ROWWISE_MAX_M = 4;  // EXACT mode: 2*per MFMAs per x row and KiB; beyond this the tile kernel (gemv_mfma.hip) takes over

// Kernel arguments, structure-of-arrays so that one batch of scalar loads fetches every layer's fields and the current
// layer is picked with scalar selects (a dependent descriptor load costs ~500 cycles of pure latency per row otherwise).
struct GvArgs {
  const uint8_t* Wq[GV_MAXL];
  const half_t* scale[GV_MAXL];
  const half_t* zero[GV_MAXL];
  const half_t* bias[GV_MAXL];
  half_t* y[GV_MAXL];
  int N[GV_MAXL];          // out_features
  int prow_end[GV_MAXL];   // end (exclusive) of layer i's packed rows in the group's concatenated row space;
                           // unused entries repeat the last layer
  const half_t* x;
  int K, gs, G /* K / gs */, total_prow;
  int red_off;  // byte offset of the K-split reduction buffer in LDS
  int ksplit;   // 1: the workgroup's waves share ONE packed row (units interleaved), partial sums meet in LDS — for few-row / long-K layers
};

// What the kernel receives.  GvIn — everything the streaming loop reads — travels as plain scalar kernel parameters (GV_IN_PARAMS
// below; the struct is rebuilt from them inside the kernel), is fetched by one batch of scalar loads at the top and lives in SGPRs.
// GvOut — where a finished row goes — stays a struct and is read with an indexed scalar load when a row ends (once per row, scalar
// cache) instead of occupying 16 more SGPRs for the whole kernel (with them the loop spilled SGPRs into VGPR lanes: ~25 v_readlane
// per row).  Kept apart because one indexed access makes the compiler treat a whole argument struct as memory and stage its loads
// behind each other.
struct GvIn {
  const uint8_t* Wq[GV_MAXL];
  const half_t* scale[GV_MAXL];
  const half_t* zero[GV_MAXL];
  int N[GV_MAXL];
  int prow_end[GV_MAXL];
  const half_t* x;
  int K, gs, G, total_prow, red_off, ksplit;
};
struct GvOut {
  const half_t* bias[GV_MAXL];
  half_t* y[GV_MAXL];
};
static_assert(GV_MAXL == 4, "the scalar parameter list below spells four layers out");
// what a wave of layer 0 needs for its x loads and its first unit's requests leads the parameter list — 14 dwords, as many as the hardware
// preloads into SGPRs at wave launch (-mllvm -amdgpu-kernarg-preload-count=16, Makefile) — so that those requests do not wait for the scalar
// loads of the rest (round 3: -3.9 % per token on the 7B stack)
#define GV_IN_PARAMS                                                                                                                    \
  const half_t *x_, int K_, int G_, int total_, int pe0, int N0, int ksplit_, const uint8_t *Wq0, const half_t *sc0, const half_t *ze0, \
      int gs_, int red_off_, const uint8_t *Wq1, const uint8_t *Wq2, const uint8_t *Wq3, const half_t *sc1, const half_t *sc2,          \
      const half_t *sc3, const half_t *ze1, const half_t *ze2, const half_t *ze3, int N1, int N2, int N3, int pe1, int pe2, int pe3
#define GV_IN_ARGS(in)                                                                                                                       \
  (in).x, (in).K, (in).G, (in).total_prow, (in).prow_end[0], (in).N[0], (in).ksplit, (in).Wq[0], (in).scale[0], (in).zero[0], (in).gs,        \
      (in).red_off, (in).Wq[1], (in).Wq[2], (in).Wq[3], (in).scale[1], (in).scale[2], (in).scale[3], (in).zero[1], (in).zero[2], (in).zero[3], \
      (in).N[1], (in).N[2], (in).N[3], (in).prow_end[1], (in).prow_end[2], (in).prow_end[3]
#define GV_IN_PACK \
  GvIn { {Wq0, Wq1, Wq2, Wq3}, {sc0, sc1, sc2, sc3}, {ze0, ze1, ze2, ze3}, {N0, N1, N2, N3}, {pe0, pe1, pe2, pe3}, x_, K_, gs_, G_, total_, red_off_, ksplit_ }

// gemv_block.hip's rotary epilogue (kernel parameter, read when a row ends): cos / sin [head_dim] of the position, the position itself in device memory
struct GbRope {
  const uint16_t* cos;
  const uint16_t* sin;
  const int64_t* pos;
  int hd, cache_len;
};

// the layer a wave is currently streaming (all wave-uniform -> SGPRs)
struct LayerCtx {
  const uint8_t* Wq;
  const half_t* scale;
  const half_t* zero;
  int N, row0, end;
};

__device__ __forceinline__ LayerCtx select_layer(const GvIn& a, int prow) {
  LayerCtx c{a.Wq[0], a.scale[0], a.zero[0], a.N[0], 0, a.prow_end[0]};
#pragma unroll
  for (int i = 1; i < GV_MAXL; ++i) {
    const bool in = prow >= a.prow_end[i - 1];   // entries past the last layer repeat it: never true for prow < total
    c.Wq = pick(in, a.Wq[i], c.Wq);
    c.scale = pick(in, a.scale[i], c.scale);
    c.zero = pick(in, a.zero[i], c.zero);
    c.N = pick(in, a.N[i], c.N);
    c.row0 = pick(in, a.prow_end[i - 1], c.row0);
    c.end = pick(in, a.prow_end[i], c.end);
  }
  return c;
}

// what a finished row needs of its layer
struct OutCtx {
  const half_t* bias;
  half_t* y;
  int N, row0;
};
__device__ __forceinline__ OutCtx select_out(const GvIn& a, const GvOut& o, int prow) {
  int row0 = 0, N = a.N[0];
#pragma unroll
  for (int i = 1; i < GV_MAXL; ++i) {
    const bool in = prow >= a.prow_end[i - 1];
    row0 = pick(in, a.prow_end[i - 1], row0);
    N = pick(in, a.N[i], N);
  }
  // the layer's index on the SCALAR side: written in C++ (a sum, or a chain of selects, of the three comparisons) the compiler computed it with
  // v_cndmask / v_addc and read it back with v_readfirstlane at every row end.  (The pointers are then read with an indexed SCALAR load; pointer arithmetic on
  // the argument struct instead made them a vector load + v_readfirstlane whose vmcnt wait also waited for the next units' weights.)
  int li;
  asm("s_cmp_ge_i32 %1, %2\n\ts_cselect_b32 %0, 1, 0\n\ts_cmp_ge_i32 %1, %3\n\ts_cselect_b32 %0, 2, %0\n\ts_cmp_ge_i32 %1, %4\n\ts_cselect_b32 %0, 3, %0"
      : "=&s"(li) : "s"(__builtin_amdgcn_readfirstlane(prow)), "s"(a.prow_end[0]), "s"(a.prow_end[1]), "s"(a.prow_end[2]) : "scc");
  const half_t* bias = o.bias[li];
  half_t* y = o.y[li];
  return OutCtx{bias, y, N, row0};
}

// (zero, scale) of a group as fetched — two 2-byte loads — as ONE dword z | sc << 16.  Written as the build of a two-element 16-bit vector:
// the compiler packs that with one v_perm_b32 and does not mask the registers' upper halves first (as it does for `z | sc << 16` on
// zero-extended values: v_and + v_lshl_or, or v_lshlrev + an SDWA or).
typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_zs(uint32_t z, uint32_t sc) {
  return __builtin_bit_cast(uint32_t, u16x2{static_cast<uint16_t>(z), static_cast<uint16_t>(sc)});
}

// everything a wave has in flight for one unit: GV_U x 16 bytes of packed weights per lane + the unit's meta
template <int PER, bool GS64>
struct Unit {
  u32x4 w[GV_U];
  // raw 2-byte loads, combined only when consumed (combining at issue time would wait on the loads)
  // GS64: z[s], sc[s] = zero / scale of group (unit's first group + lane) of slab s
  // else: z[u * PER + s], sc[..] = those of the group the lane's own 16 k-values of load u fall into
  // (32-bit holders of the zero-extended 2-byte loads: a uint16_t carried round the loop gets masked — and so waited for — where the
  // compiler places the phi, in front of the next unit's requests)
  uint32_t z[GS64 ? PER : GV_U * PER];
  uint32_t sc[GS64 ? PER : GV_U * PER];
};

}  // namespace hqq
