// gemv_mfma.hip — fused unpack -> dequantize -> skinny GEMM on the matrix cores for small batches (decode with several
// sequences): hqq_hip_gemv / hqq_hip_gemv_grouped route 5 <= M <= 16 activation rows here (M <= 4 stays on the row-per-wave
// kernel of gemv.hip, which streams each packed row contiguously).  gfx950.
//
// Reference chain replaced (axis=1): BitPack.unpack_* -> (W_r - zero) * scale -> torch.matmul(x, W.t()) (+ bias)
//   hqq/core/bitpack.py:31-64, hqq/core/quantize.py:183-199, :880-898; patching.py:82-86 ("TODO GEMV use-case").
//
// Why a tile kernel here.  Rebuilding each weight exactly as the reference does — w = round16(round16(q - z) * s), 4
// packed-fp16 ops per weight pair — costs ~130 ns per wave-KiB on gfx950 whatever M is; the row-per-wave kernel then spends
// 2*per MFMAs per KiB *per activation row*, which passes the VALU time around M = 4.  With 16 packed rows per wave the weights
// are the MFMA A operand and all M <= 16 activation rows ride in one B operand: the cost no longer depends on M.  The weights
// it multiplies are bit-identical to hqq_hip_dequantize / Quantizer.dequantize (only the fp32 summation order differs from a
// BLAS GEMM).  Round-1 status: 1.0-1.8 TB/s (a wave instruction touches 16 rows x 64 B, i.e. 16 DRAM pages at once, and x is
// re-read from L2 per tile); next: stage the packed bytes through LDS so that global reads stay row-contiguous.
//
// Data layout, consumed as the reference stores it (no repacking):
//   Wq     [N/per, K] bytes; byte (p, k) holds W_q[p + s*N/per, k] for slab s at bit 8 - nbits*(s+1)
//   scale  [N*G], zero [N*G] fp16, G = K/group_size; output row n uses [n*G, (n+1)*G)
//
// Work decomposition.  A *tile* is 16 packed rows (-> 16*per output rows); a workgroup of KS waves owns one tile at a time
// and its waves split K into KS contiguous slices (KS is picked per launch so that even a 4096x4096 layer fills all 1024
// SIMDs).  A wave walks its slice in 64-k blocks: lane (r = lane & 15, c = lane >> 4) loads the 16 packed bytes of row r
// at k = 64*kb + 16*c (global_load_dwordx4, non-temporal; a wave instruction covers 16 rows x 64 contiguous bytes),
// dequantises them in registers, and feeds two MFMAs per slab (k-octets 16c + 0..7 and 16c + 8..15).  Loads are issued
// two blocks (one *unit*, 2 KiB of weights per wave) ahead of their use, ping-pong between two register sets.
//   x      [M, K] is read straight from global memory (it is a few hundred KiB at most and L2-resident): lane (r, c) loads
//          the two k-octets it needs of activation row r together with the weights of the same unit and permutes them in
//          registers into the k order the nibble extraction produces ((k0,k2),(k1,k3),(k4,k6),(k5,k7)); lanes of the unused
//          columns M..15 hold zeros.  No LDS staging, hence no limit on M*K and no extra passes over the weights.
//   meta   group_size 64: one 64-k block is exactly one group; per unit each lane fetches (zero, scale) of group
//          (2*unit + c) of its row with 2-byte loads and the unit's blocks pick theirs with ds_bpermute.
//          other group sizes: fetched per lane and block.
// The KS partial accumulators of a tile are summed through LDS; wave 0 rounds to fp16, adds the bias and stores.
// Several layers that read the same x (q/k/v, gate/up) form one launch: their tiles are concatenated.
#include "hqq_common.h"

namespace hqq {

constexpr int GM_MAXL = HQQ_GEMV_MAX_GROUP;
constexpr int GM_UB = 2;              // 64-k blocks per unit

typedef _Float16 h8_t __attribute__((ext_vector_type(8)));

struct GmArgs {
  const uint8_t* Wq[GM_MAXL];
  const half_t* scale[GM_MAXL];
  const half_t* zero[GM_MAXL];
  const half_t* bias[GM_MAXL];
  half_t* y[GM_MAXL];
  int N[GM_MAXL];          // out_features
  int tile_end[GM_MAXL];   // end (exclusive) of layer i's tiles in the group's concatenated tile space (unused entries repeat the last)
  const half_t* x;
  int M, K, gs, total_tiles;
};

struct GmLayer {   // the layer a workgroup is currently streaming (workgroup-uniform -> SGPRs)
  const uint8_t* Wq;
  const half_t* scale;
  const half_t* zero;
  const half_t* bias;
  half_t* y;
  int N, tile0;
};

__device__ __forceinline__ GmLayer gm_select(const GmArgs& a, int tile) {
  GmLayer c{a.Wq[0], a.scale[0], a.zero[0], a.bias[0], a.y[0], a.N[0], 0};
#pragma unroll
  for (int i = 1; i < GM_MAXL; ++i) {
    const bool in = tile >= a.tile_end[i - 1];
    c.Wq = in ? a.Wq[i] : c.Wq;
    c.scale = in ? a.scale[i] : c.scale;
    c.zero = in ? a.zero[i] : c.zero;
    c.bias = in ? a.bias[i] : c.bias;
    c.y = in ? a.y[i] : c.y;
    c.N = in ? a.N[i] : c.N;
    c.tile0 = in ? a.tile_end[i - 1] : c.tile0;
  }
  return c;
}

__device__ __forceinline__ half2_t gm_h2(uint32_t u) { return __builtin_bit_cast(half2_t, u); }
__device__ __forceinline__ uint32_t gm_u32(half2_t h) { return __builtin_bit_cast(uint32_t, h); }

// exact integer levels of slab S for the byte pairs (b0,b2) [word] / (b1,b3) [word >> 8] of a packed dword as fp16:
// (word & mask) | 0x6400 is the fp16 number 1024 + q * 2^sh; one packed fma removes the bias exactly.
template <int NBITS, int S>
__device__ __forceinline__ half2_t gm_levels(uint32_t word_or_shifted, uint32_t magic) {
  constexpr int per = 8 / NBITS;
  constexpr int sh = NBITS * (per - 1 - S);
  constexpr uint32_t m1 = ((NBITS == 8) ? 0xFFu : ((1u << NBITS) - 1u)) << sh;
  constexpr uint32_t m = m1 | (m1 << 16);
  uint32_t b;
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(b) : "v"(word_or_shifted), "s"(m), "v"(magic));
  constexpr float inv = 1.0f / static_cast<float>(1 << sh);
  const half2_t k1 = {static_cast<half_t>(inv), static_cast<half_t>(inv)};
  const half2_t k2 = {static_cast<half_t>(-1024.0f * inv), static_cast<half_t>(-1024.0f * inv)};
  return __builtin_elementwise_fma(gm_h2(b), k1, k2);
}

__device__ __forceinline__ u32x4 gm_permute_x8(u32x4 v) {   // (k0..k7) -> (k0,k2,k1,k3,k4,k6,k5,k7)
  u32x4 r;
  r.x = (v.x & 0xFFFFu) | (v.y << 16);
  r.y = (v.x >> 16) | (v.y & 0xFFFF0000u);
  r.z = (v.z & 0xFFFFu) | (v.w << 16);
  r.w = (v.z >> 16) | (v.w & 0xFFFF0000u);
  return r;
}

// one 64-k block of one slab: dequantise the lane's 16 weights exactly as Quantizer.dequantize does (two fp16 roundings)
// and contract them with the activation octets on the matrix core
template <int NBITS, int S, int PER>
struct GmSlab {
  static __device__ __forceinline__ void run(const u32x4& w, const uint32_t (&zs)[PER], const h8_t& b0, const h8_t& b1,
                                             f32x4 (&acc)[PER], uint32_t magic) {
    const half2_t pr = gm_h2(zs[S]);
    const half2_t zz = {pr.x, pr.x}, ss = {pr.y, pr.y};
    uint32_t o[8];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const half2_t q0 = gm_levels<NBITS, S>(w[d], magic);        // bytes (4d+0, 4d+2)
      const half2_t q1 = gm_levels<NBITS, S>(w[d] >> 8, magic);   // bytes (4d+1, 4d+3)
      o[2 * d] = gm_u32((q0 - zz) * ss);
      o[2 * d + 1] = gm_u32((q1 - zz) * ss);
    }
    const h8_t a0 = __builtin_bit_cast(h8_t, u32x4{o[0], o[1], o[2], o[3]});   // k = 16c + 0..7 (permuted inside the octet)
    const h8_t a1 = __builtin_bit_cast(h8_t, u32x4{o[4], o[5], o[6], o[7]});   // k = 16c + 8..15
    acc[S] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, acc[S], 0, 0, 0);
    acc[S] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, acc[S], 0, 0, 0);
    if constexpr (S + 1 < PER) GmSlab<NBITS, S + 1, PER>::run(w, zs, b0, b1, acc, magic);
  }
};

template <int PER, bool GS64>
struct GmUnit {
  u32x4 w[GM_UB];
  u32x4 xv[GM_UB][2];   // this lane's two k-octets of activation row r per block (zeros for r >= M)
  // raw 2-byte loads, combined only when consumed.  GS64: the pair of group (4*unit + c) of the lane's row, per slab;
  // otherwise the pair of the group the lane's 16 k-values of block b fall into, [b * PER + s]
  uint16_t z[GS64 ? PER : GM_UB * PER];
  uint16_t sc[GS64 ? PER : GM_UB * PER];
};

// (a workgroup is KS <= min(32 / PER, 16) waves — gm_launch — so the 2- and 1-bit instantiations, whose four / eight slabs of accumulators do not fit
//  the 128 registers a 1024-thread bound leaves, declare what they are launched with: no scratch; until round 6 they spilled 22 / 239 registers)
template <int NBITS, bool GS64>
__global__ __launch_bounds__(NBITS == 2 ? 512 : (NBITS == 1 ? 256 : 1024)) void gemv_mfma_f16_kernel(const GmArgs a) {
  constexpr int PER = 8 / NBITS;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int KS = blockDim.x >> 6;                       // waves per workgroup = K slices per tile
  const int r = lane & 15, c = lane >> 4;
  const int K = a.K, gs = a.gs, G = K / gs, M = a.M;
  const int nb = K >> 6;                                // 64-k blocks per row
  // this wave's K slice, in blocks: [kb0, kb1)
  const int kb0 = static_cast<int>(static_cast<int64_t>(nb) * wave / KS);
  const int kb1 = static_cast<int>(static_cast<int64_t>(nb) * (wave + 1) / KS);
  const int nunits = (kb1 - kb0 + GM_UB - 1) / GM_UB;   // may be 0 when KS > nb

  f32x4* red = reinterpret_cast<f32x4*>(smem);   // LDS: reduction buffer [KS][PER][64] f32x4 only

  // Every issue() emits exactly GM_UB weight loads + the unit's meta loads, live or not, so that the waits the compiler
  // derives are exact vmcnt(<loads of the following unit>).  Blocks past the slice / rows past the layer re-read a valid
  // address; their contribution is discarded (row mask at the store) or multiplied by zero activations... see consume().
  auto issue = [&](GmUnit<PER, GS64>& un, const GmLayer& ly, int tile, int unit) {
    const int rows_per_slab = ly.N / PER;
    int p = (tile - ly.tile0) * 16 + r;                  // packed row inside the layer
    p = p < rows_per_slab ? p : rows_per_slab - 1;       // ragged last tile: duplicate the last row (masked at the store)
    const int kbu = kb0 + unit * GM_UB;
    if constexpr (GS64) {
      int g = kbu + c;
      g = g < kb1 ? g : kb0;
#pragma unroll
      for (int s = 0; s < PER; ++s) {
        const int64_t q = static_cast<int64_t>(p + s * rows_per_slab) * G + g;
        un.z[s] = __builtin_bit_cast(uint16_t, ly.zero[q]);
        un.sc[s] = __builtin_bit_cast(uint16_t, ly.scale[q]);
      }
    } else {
#pragma unroll
      for (int b = 0; b < GM_UB; ++b) {
        int kb = kbu + b;
        kb = kb < kb1 ? kb : kb0;
        const int g = (kb * 64 + c * 16) / gs;
#pragma unroll
        for (int s = 0; s < PER; ++s) {
          const int64_t q = static_cast<int64_t>(p + s * rows_per_slab) * G + g;
          un.z[b * PER + s] = __builtin_bit_cast(uint16_t, ly.zero[q]);
          un.sc[b * PER + s] = __builtin_bit_cast(uint16_t, ly.scale[q]);
        }
      }
    }
    const uint8_t* wrow = ly.Wq + static_cast<int64_t>(p) * K + c * 16;
#pragma unroll
    for (int b = 0; b < GM_UB; ++b) {
      int kb = kbu + b;
      kb = kb < kb1 ? kb : kb0;                          // past the slice: re-read its first block (skipped by consume)
      un.w[b] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow + static_cast<int64_t>(kb) * 64));
      const u32x4* xp = reinterpret_cast<const u32x4*>(a.x + static_cast<int64_t>(r < M ? r : 0) * K + kb * 64 + c * 16);
      un.xv[b][0] = xp[0];
      un.xv[b][1] = xp[1];
    }
  };

  int tile = blockIdx.x;
  const int total = a.total_tiles;
  GmLayer ly = gm_select(a, tile < total ? tile : total - 1);
  GmUnit<PER, GS64> ua, ub;
  const bool have_work = nunits > 0 && tile < total;
  if (have_work) issue(ua, ly, tile, 0);

  const bool col_live = r < M;   // this lane's MFMA column carries a real activation row

  uint32_t magic;
  asm volatile("v_mov_b32 %0, 0x64006400" : "=v"(magic));   // opaque to the optimiser: stays in a VGPR
  f32x4 acc[PER];
#pragma unroll
  for (int s = 0; s < PER; ++s) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto consume = [&](const GmUnit<PER, GS64>& cur, int unit) {
    const int kbu = kb0 + unit * GM_UB;
#pragma unroll
    for (int b = 0; b < GM_UB; ++b) {
      const int kb = kbu + b;
      uint32_t zs[PER];
#pragma unroll
      for (int s = 0; s < PER; ++s) {
        if constexpr (GS64) {
          const uint32_t mine = static_cast<uint32_t>(cur.z[s]) | (static_cast<uint32_t>(cur.sc[s]) << 16);
          zs[s] = __builtin_amdgcn_ds_bpermute((r + 16 * b) << 2, mine);   // lane (r, c = b) fetched block b's group
        } else {
          zs[s] = static_cast<uint32_t>(cur.z[b * PER + s]) | (static_cast<uint32_t>(cur.sc[b * PER + s]) << 16);
        }
      }
      if (kb < kb1) {   // wave-uniform
        const u32x4 zero4 = {0u, 0u, 0u, 0u};
        const h8_t b0 = __builtin_bit_cast(h8_t, col_live ? gm_permute_x8(cur.xv[b][0]) : zero4);
        const h8_t b1 = __builtin_bit_cast(h8_t, col_live ? gm_permute_x8(cur.xv[b][1]) : zero4);
        GmSlab<NBITS, 0, PER>::run(cur.w[b], zs, b0, b1, acc, magic);
      }
    }
  };

  // tile finished: sum the KS partial tiles through LDS; wave 0 rounds, adds the bias and stores
  auto finish = [&](const GmLayer& oly, int otile) {
#pragma unroll
    for (int s = 0; s < PER; ++s) {
      red[(wave * PER + s) * 64 + lane] = acc[s];
      acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    if (wave == 0) {
      const int rows_per_slab = oly.N / PER;
      const int p_base = (otile - oly.tile0) * 16 + c * 4;   // D layout: rows 4c + i, column r
#pragma unroll
      for (int s = 0; s < PER; ++s) {
        f32x4 t = red[s * 64 + lane];
        for (int w = 1; w < KS; ++w) {
          const f32x4 u = red[(w * PER + s) * 64 + lane];
          t[0] += u[0]; t[1] += u[1]; t[2] += u[2]; t[3] += u[3];
        }
        if (r < M) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int p = p_base + i;
            if (p < rows_per_slab) {
              const int n = p + s * rows_per_slab;
              half_t o = static_cast<half_t>(t[i]);
              if (oly.bias) o = o + oly.bias[n];   // `out += bias` on the rounded matmul result (quantize.py:896-897)
              oly.y[static_cast<int64_t>(r) * oly.N + n] = o;
            }
          }
        }
      }
    }
    __syncthreads();
  };

  if (tile < total) {
    if (nunits == 0) {
      // more waves than 64-k blocks (tiny K): this wave only takes part in the reductions
      for (; tile < total; tile += gridDim.x) {
        const GmLayer oly = gm_select(a, tile);
        finish(oly, tile);
      }
    } else {
      int unit = 0;
      GmLayer la = ly;
      for (;;) {
        // next unit: same tile, or the workgroup's next tile
        int t1 = tile, u1 = unit + 1;
        if (u1 == nunits) { u1 = 0; t1 = tile + gridDim.x; }
        if (t1 >= total) { consume(ua, unit); finish(la, tile); break; }
        if (u1 == 0) ly = gm_select(a, t1);
        const GmLayer lb = ly;
        issue(ub, lb, t1, u1);
        consume(ua, unit);
        if (u1 == 0) finish(la, tile);
        int t2 = t1, u2 = u1 + 1;
        if (u2 == nunits) { u2 = 0; t2 = t1 + gridDim.x; }
        if (t2 >= total) { consume(ub, u1); finish(lb, t1); break; }
        if (u2 == 0) ly = gm_select(a, t2);
        la = ly;
        issue(ua, la, t2, u2);
        consume(ub, u1);
        if (u2 == 0) finish(lb, t1);
        tile = t2;
        unit = u2;
      }
    }
  }
}

static int gm_num_cus() {
  static int n_cus = 0;
  if (n_cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) n_cus = n;
    else n_cus = 256;
  }
  return n_cus;
}

template <int NBITS, bool GS64>
static int gm_launch(GmArgs& a, hipStream_t st) {
  constexpr int PER = 8 / NBITS;
  const int nb = a.K >> 6;
  const int cus = gm_num_cus();
  // K slices per tile: enough waves to put >= 8 on every CU when the layer is small, at least one unit (4 blocks) each
  int ks = (cus * 8 + a.total_tiles - 1) / a.total_tiles;
  const int ks_max = nb / GM_UB > 0 ? nb / GM_UB : 1;
  ks = ks > ks_max ? ks_max : ks;
  const int ks_cap = 32 / PER < 16 ? 32 / PER : 16;      // reduction buffer <= 32 KiB
  ks = ks > ks_cap ? ks_cap : (ks < 1 ? 1 : ks);
  const size_t lds = static_cast<size_t>(ks) * PER * 64 * sizeof(f32x4);
  int wg_per_cu = 16 / ks;                               // <= 16 waves per CU (105 VGPRs -> 4 waves per SIMD)
  wg_per_cu = wg_per_cu < 1 ? 1 : wg_per_cu;
  const int cap = cus * wg_per_cu;
  const int grid = a.total_tiles < cap ? a.total_tiles : cap;
  auto kern = gemv_mfma_f16_kernel<NBITS, GS64>;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(ks * 64), lds, st, a);
  return check_launch("hqq_hip_gemv");
}

static int gm_dispatch(int nbits, GmArgs& a, hipStream_t st) {
  const bool gs64 = a.gs == 64;
  switch (nbits) {
    case 8: return gm_launch<8, false>(a, st);
    case 4: return gs64 ? gm_launch<4, true>(a, st) : gm_launch<4, false>(a, st);
    case 2: return gs64 ? gm_launch<2, true>(a, st) : gm_launch<2, false>(a, st);
    case 1: return gm_launch<1, false>(a, st);
  }
  return HQQ_ERR_NBITS;
}

// exact-weights skinny GEMM; called by hqq_hip_gemv_grouped (gemv.hip) after argument validation
int gemv_mfma_run(int nbits, int n_layers, const void* x, const void* const* Wq, const void* const* scale, const void* const* zero,
                  const void* const* bias, void* const* y, const int64_t* N, int64_t M, int64_t K, int64_t group_size, hipStream_t st) {
  const int per = 8 / nbits;
  GmArgs a;
  int64_t tiles = 0;
  for (int i = 0; i < n_layers; ++i) {
    tiles += (N[i] / per + 15) / 16;
    if (tiles > INT32_MAX) { set_error("hqq_hip_gemv: size overflow"); return HQQ_ERR_SHAPE; }
    a.Wq[i] = static_cast<const uint8_t*>(Wq[i]);
    a.scale[i] = static_cast<const half_t*>(scale[i]);
    a.zero[i] = static_cast<const half_t*>(zero[i]);
    a.bias[i] = bias ? static_cast<const half_t*>(bias[i]) : nullptr;
    a.y[i] = static_cast<half_t*>(y[i]);
    a.N[i] = static_cast<int>(N[i]);
    a.tile_end[i] = static_cast<int>(tiles);
  }
  for (int i = n_layers; i < GM_MAXL; ++i) {
    a.Wq[i] = a.Wq[n_layers - 1]; a.scale[i] = a.scale[n_layers - 1]; a.zero[i] = a.zero[n_layers - 1]; a.bias[i] = a.bias[n_layers - 1];
    a.y[i] = a.y[n_layers - 1]; a.N[i] = a.N[n_layers - 1]; a.tile_end[i] = a.tile_end[n_layers - 1];
  }
  a.K = static_cast<int>(K);
  a.gs = static_cast<int>(group_size);
  a.total_tiles = static_cast<int>(tiles);
  a.M = static_cast<int>(M);
  a.x = static_cast<const half_t*>(x);
  const int rc = gm_dispatch(nbits, a, st);
  if (rc) return rc;
  return 0;
}

}  // namespace hqq
