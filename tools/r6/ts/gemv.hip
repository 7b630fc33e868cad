// gemv.hip — the decode kernel: fused unpack -> dequantize -> GEMV for HQQLinear.forward with a few activation rows, gfx950,
// and the C entry points of the decode path (hqq_hip_gemv / hqq_hip_gemv_grouped), which also route to
// gemv3.hip (3-bit containers) and gemv_mfma.hip (5..16 rows).
//
// Replaces, for axis=1 layers, the reference's per-call chain
//   BitPack.unpack_*  -> (W_r - zero) * scale -> torch.matmul(x, W.t()) (+ bias)
//   (hqq/core/bitpack.py:31-64, hqq/core/quantize.py:183-199, :880-898; patching.py:82-86 "TODO GEMV use-case")
// which moves ~12.5 B/param through HBM, by one pass over the packed weights (0.5625 B/param at 4-bit).  HBM-bandwidth bound.
//
// Data layout consumed as stored by the reference (no repacking):
//   Wq     [N/per, K] bytes; byte (p, k) holds W_q[p + s*N/per, k] for slab s at bit 8 - nbits*(s+1)
//   scale  [N*G] , zero [N*G] in the compute dtype, G = K/group_size; row n uses [n*G, (n+1)*G)
//
// Launch shape: a *group* of up to GV_MAXL layers that read the same activation rows (q/k/v, gate/up, or a single layer) is
// one launch.  Their packed rows form one concatenated row space that a persistent grid (<= 4 workgroups of 4 waves per CU)
// strides over; one wave owns one packed row (-> `per` output rows) at a time and walks it in 2 KiB units
// (2 x buffer_load_dwordx4 per lane, non-temporal, 1 KiB per wave instruction; descriptor + scalar row offset + one 32-bit lane
// offset).  The loads of unit i+1 are issued before unit i is consumed (two register sets, no copies), every issue() emits the same
// number of loads and the streaming loop has one shape and one exit, so that the compiler's waits are exact `s_waitcnt vmcnt(n)` and
// none sits in front of a request; the very first unit is requested before x is staged.  Few-row / long-K layers switch to K-split:
// the workgroup's waves share one row.
//   x       staged once per workgroup in LDS, in the order the nibble extraction produces values
//   meta    per unit the (zero, scale) of the <= 64 groups it spans are fetched with one coalesced 2-byte load per lane and slab
//           and handed to the consuming lanes with ds_bpermute (group_size 64; other group sizes fetch per lane)
//   layers  kernel arguments are structure-of-arrays; the current layer is picked with scalar selects and lives in SGPRs
//
// Arithmetic, two modes (template parameter EXACT; per-call option bit HQQ_OPT_FACTORED):
//  EXACT (default)  every weight pair is rebuilt exactly as Quantizer.dequantize does it — (w & mask) | 0x6400 -> v_pk_fma (exact
//           level) -> v_pk_add(-zero) -> v_pk_mul(scale): two fp16 roundings, bit-identical to hqq_hip_dequantize / the reference
//           — and contracted on the matrix core: all 64 lanes hold the SAME output row, lane (i = l & 15, o = l >> 4) supplies
//           row i / k-octet o of the A operand (its 8 weights) and column i / k-octet o of B (its 8 x values); D[i][i] is the
//           partial dot product of lanes {i + 16 o}, the row result the sum of the diagonal.  15/16 of the MFMA flops are thrown
//           away on purpose: the matrix pipe is otherwise idle and the dot product costs no VALU slot.  bf16: same, with the two
//           roundings done through fp32 (v_cvt_pk_bf16_f32 / v_dot2_f32_bf16), gfx950 having no packed bf16 arithmetic.
//  FACTORED gfx950 issues packed-fp16 / dot2 / three-operand VALU at ~2/3 of the v_fma_f32 rate (tools/instr_bench.hip), so the
//           4 ops per weight pair of EXACT cap the kernel near 4.5 TB/s.  FACTORED takes the group affine map out of the dot
//           product:  sum_k x_k (q_k - z) s  =  (s/F) * ( sum_k x_k (1024 + F q_k)  -  (1024 + F z) * sum_k x_k ),  where
//           1024 + F q_k is the fp16 number obtained by OR-ing the exponent 0x6400 onto the masked nibble (F = 2^shift of the
//           slab) — one v_and_or_b32 + one v_dot2 per weight pair — and sum_k x_k per 16-k lane chunk is precomputed when x is
//           staged.  Everything after the nibble is fp32; no per-weight fp16 rounding, so results differ from the reference by
//           less than its own weight-rounding noise (<= 2^-10 * sum|x_k w_k|; tests state the tolerance).  ~20 % faster.
#include <type_traits>

#include "gemv_shared.h"
#include "w3s.h"
__device__ unsigned long long* g_lab_ts_dev = nullptr;
extern "C" int hqq_lab_set_ts(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_lab_ts_dev), &p, sizeof(p)); }

namespace hqq {

// one 16-byte weight vector (16 k-values of `PER` rows) against the lane's 16 x-values of M rows
template <int NBITS, int M, int S, int PER>
struct SlabLoop {
  static __device__ __forceinline__ void run(const u32x4& w, const float (&c1)[PER], const float (&c2)[PER], const half2_t (&xr)[M][8],
                                             const float (&xsum)[M], float (&acc)[M][PER], uint32_t magic) {
    float dot[M][2];   // two chains per row: v_dot2 results are needed ~8 cycles after issue
#pragma unroll
    for (int m = 0; m < M; ++m) dot[m][0] = dot[m][1] = 0.f;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const uint32_t word = w[d];
      const half2_t q0 = biased_levels<NBITS, S>(word, magic);        // bytes (4d+0, 4d+2)
      const half2_t q1 = biased_levels<NBITS, S>(word >> 8, magic);   // bytes (4d+1, 4d+3)
#pragma unroll
      for (int m = 0; m < M; ++m) {
        dot[m][0] = __builtin_amdgcn_fdot2(q0, xr[m][2 * d], dot[m][0], false);
        dot[m][1] = __builtin_amdgcn_fdot2(q1, xr[m][2 * d + 1], dot[m][1], false);
      }
    }
#pragma unroll
    for (int m = 0; m < M; ++m)
      acc[m][S] = __builtin_fmaf(c1[S], __builtin_fmaf(-c2[S], xsum[m], dot[m][0] + dot[m][1]), acc[m][S]);
    if constexpr (S + 1 < PER) SlabLoop<NBITS, M, S + 1, PER>::run(w, c1, c2, xr, xsum, acc, magic);
  }
};

// (SlabExactBF16 — the bf16 exact rebuild — lives in decode_common.h: gemv_block.hip compiles the same kernel text)
template <int NBITS, int S, int PER>
struct GroupConst {   // c1 = s / F, c2 = 1024 + F z  for every slab, from the raw fp16 bit patterns
  static __device__ __forceinline__ void run(const uint32_t* z, const uint32_t* sc, float (&c1)[PER], float (&c2)[PER]) {
    const float zf = static_cast<float>(__builtin_bit_cast(half_t, static_cast<uint16_t>(z[S])));
    const float sf = static_cast<float>(__builtin_bit_cast(half_t, static_cast<uint16_t>(sc[S])));
    c1[S] = sf * SlabF<NBITS, S>::invF;
    c2[S] = __builtin_fmaf(zf, SlabF<NBITS, S>::F, 1024.0f);
    if constexpr (S + 1 < PER) GroupConst<NBITS, S + 1, PER>::run(z, sc, c1, c2);
  }
};

// WPG: waves per workgroup.  4 (x 4 workgroups per CU) everywhere, except single layers of at most one packed row per wave of an
// 8 x 2 grid (o / down of a 7B block: 2048 packed rows), where 8 (x 2 per CU) measured -7 % per launch (fewer workgroups to dispatch and
// half as many copies of x staged per CU; grouped launches lose 11 % with it: profiles/r03_ab_w8x2.txt)
#define GV_KERNEL_NAME gemv_f16_kernel
#define GV_KERNEL_XPASS2 0
#include "gemv_kernel.inc"
#undef GV_KERNEL_NAME
#undef GV_KERNEL_XPASS2
#define GV_KERNEL_NAME gemv_f16_xp2_kernel
#define GV_KERNEL_XPASS2 1
#include "gemv_kernel.inc"
#undef GV_KERNEL_NAME
#undef GV_KERNEL_XPASS2

static int g_num_cus = 0;
static int num_cus() {
  if (g_num_cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) g_num_cus = n;
    else g_num_cus = 256;
  }
  return g_num_cus;
}

template <int NBITS, int M, bool GS64, bool EXACT, bool BF16 = false, bool SUB = false, int WPG = GV_WAVES>
static int launch_gemv_f16(const GvArgs& args, hipStream_t st) {
  constexpr int WG_PER_CU = GV_WG_PER_CU * GV_WAVES / WPG;   // the same 16 waves per CU
  if constexpr (WPG == GV_WAVES && M == 1 && GS64 && EXACT && !BF16 && (NBITS == 8 || NBITS == 4 || NBITS == 2)) {
    // a single layer with at most one packed row per wave of the wide grid, whose rows span at least one unit per wave: 8 waves x 2 per CU
    if (args.prow_end[0] == args.total_prow && args.total_prow <= num_cus() * 8 && args.total_prow * 2 > num_cus() * 8 && args.K >= GV_UNIT)
      return launch_gemv_f16<NBITS, M, GS64, EXACT, BF16, SUB, 8>(args, st);
  }
  constexpr int PER = 8 / NBITS;
  GvArgs a = args;
  const int nsteps = (a.K + GV_KSTEP - 1) / GV_KSTEP;
  const int nunits = (nsteps + GV_U - 1) / GV_U;
  const size_t xs_bytes = static_cast<size_t>(M) * nsteps * (GV_KSTEP * 2 + 64 * 4);   // x planes + per-chunk sums
  a.red_off = static_cast<int>((xs_bytes + 15) & ~static_cast<size_t>(15));
  const size_t lds = a.red_off + sizeof(float) * WPG * M * PER;             // + K-split reduction buffer
  auto kern = gemv_f16_kernel<NBITS, M, GS64, EXACT, BF16, SUB, WPG>;
  // a row of x longer than one pass of the workgroup's threads (the 11008-wide down projection): the variant that requests the first two
  // passes together (4096 x 11008: 8.3 -> 8.0 us; compiled as its own kernel so that the others keep their code, gemv_kernel.inc)
  int variant = 0;   // which compilation of the kernel text this launch uses (per-kernel caches below)
  if constexpr (EXACT && !BF16 && GS64) { if (nsteps * 64 > WPG * 64) { kern = gemv_f16_xp2_kernel<NBITS, M, GS64, EXACT, BF16, SUB, WPG>; variant = 1; } }
  int per_cu = static_cast<int>(160 * 1024 / (lds + 256));
  per_cu = per_cu > WG_PER_CU ? WG_PER_CU : (per_cu < 1 ? 1 : per_cu);
  {
    // registers bound the residency too (M = 4 exact needs 152 VGPRs: three workgroups per CU, not four): a persistent grid larger
    // than what is resident runs its surplus workgroups as a second round behind the first
    static int by_regs_v[2] = {0, 0};   // per instantiation and kernel variant
    int& by_regs = by_regs_v[variant];
    if (by_regs == 0) {
      hipFuncAttributes fa;
      by_regs = WG_PER_CU;
      if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kern)) == hipSuccess && fa.numRegs > 0) {
        const int regs = (fa.numRegs + 7) & ~7;                 // allocation granule
        const int waves_per_simd = 512 / regs;                  // unified VGPR/AGPR file of 512 per SIMD lane
        by_regs = waves_per_simd * 4 / WPG;                     // WPG waves per workgroup over 4 SIMDs
        by_regs = by_regs < 1 ? 1 : by_regs;
      } else {
        (void)hipGetLastError();
      }
    }
    per_cu = per_cu > by_regs ? by_regs : per_cu;
  }
  const int cap = num_cus() * per_cu;
  // few rows x long K (e.g. the 1024 x 28672 shard of a 70B down-projection): one row per wave would leave most of the chip idle
  // and each wave with 2-4 KiB in flight; let the workgroup's waves share a row instead
  // (the choice depends on the layer shape only, never on M: a row's result does not change with the batch it is computed in)
  a.ksplit = (nunits >= WPG && static_cast<int64_t>(a.total_prow) * 4 <= static_cast<int64_t>(num_cus()) * WG_PER_CU * WPG) ? 1 : 0;
  const int tiles = a.ksplit ? a.total_prow : (a.total_prow + WPG - 1) / WPG;
  const int grid = tiles < cap ? tiles : cap;
  if (lds > 64 * 1024) {
    static LdsRaised raised[2];   // per instantiation, kernel variant (and device)
    if (const int rc = raise_lds_limit(raised[variant], reinterpret_cast<const void*>(kern), GV_LDS_MAX, "hqq_hip_gemv")) return rc;
  }
  GvIn in;
  GvOut out;
  for (int i = 0; i < GV_MAXL; ++i) {
    in.Wq[i] = a.Wq[i]; in.scale[i] = a.scale[i]; in.zero[i] = a.zero[i]; in.N[i] = a.N[i]; in.prow_end[i] = a.prow_end[i];
    out.bias[i] = a.bias[i]; out.y[i] = a.y[i];
  }
  in.x = a.x; in.K = a.K; in.gs = a.gs; in.G = a.G; in.total_prow = a.total_prow; in.red_off = a.red_off; in.ksplit = a.ksplit;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(WPG * 64), lds, st, GV_IN_ARGS(in), out);
  return check_launch("hqq_hip_gemv");
}

template <int NBITS, bool GS64, bool EXACT, bool SUB = false>
static int dispatch_m(int M, const GvArgs& a, hipStream_t st) {
  switch (M) {
    case 1: return launch_gemv_f16<NBITS, 1, GS64, EXACT, false, SUB>(a, st);
    case 2: return launch_gemv_f16<NBITS, 2, GS64, EXACT, false, SUB>(a, st);
    case 3: return launch_gemv_f16<NBITS, 3, GS64, EXACT, false, SUB>(a, st);
    case 4: return launch_gemv_f16<NBITS, 4, GS64, EXACT, false, SUB>(a, st);
  }
  if constexpr (!EXACT) {
    switch (M) {
      case 5: return launch_gemv_f16<NBITS, 5, GS64, false>(a, st);
      case 6: return launch_gemv_f16<NBITS, 6, GS64, false>(a, st);
      case 7: return launch_gemv_f16<NBITS, 7, GS64, false>(a, st);
      case 8: return launch_gemv_f16<NBITS, 8, GS64, false>(a, st);
    }
  }
  return HQQ_ERR_SHAPE;
}

// bf16 compute dtype (exact weights only): 4-/2-bit, group_size 64 or generic, M <= 4
static int dispatch_bf16(int nbits, int M, const GvArgs& a, hipStream_t st) {
  const bool gs64 = a.gs == 64;
#define HQQ_BF16_CASE(NB, MM)                                                                         \
  if (nbits == NB && M == MM) return gs64 ? launch_gemv_f16<NB, MM, true, true, true>(a, st) : launch_gemv_f16<NB, MM, false, true, true>(a, st);
  HQQ_BF16_CASE(4, 1) HQQ_BF16_CASE(4, 2) HQQ_BF16_CASE(4, 3) HQQ_BF16_CASE(4, 4)
  HQQ_BF16_CASE(2, 1) HQQ_BF16_CASE(2, 2) HQQ_BF16_CASE(2, 3) HQQ_BF16_CASE(2, 4)
#undef HQQ_BF16_CASE
  set_error("hqq_hip_gemv: bf16 covers nbits 4/2 (got %d)", nbits);
  return HQQ_ERR_UNSUPPORTED;
}

template <bool EXACT, bool SUB = false>
static int dispatch(int nbits, int M, const GvArgs& a, hipStream_t st) {
  const bool gs64 = a.gs == 64;
  switch (nbits) {
    case 8: return dispatch_m<8, false, EXACT, SUB>(M, a, st);
    case 4: return gs64 ? dispatch_m<4, true, EXACT, SUB>(M, a, st) : dispatch_m<4, false, EXACT, SUB>(M, a, st);
    case 2: return gs64 ? dispatch_m<2, true, EXACT, SUB>(M, a, st) : dispatch_m<2, false, EXACT, SUB>(M, a, st);
    case 1: return dispatch_m<1, false, EXACT, SUB>(M, a, st);
  }
  return HQQ_ERR_NBITS;
}

// rows of x one launch can stage (LDS budget); larger M is served by several launches over row blocks of x
static int max_m_per_launch(int64_t K) {
  const int64_t nsteps = (K + GV_KSTEP - 1) / GV_KSTEP;
  const int64_t m = GV_LDS_MAX / (nsteps * (GV_KSTEP * 2 + 64 * 4));
  return static_cast<int>(m < 1 ? 0 : (m > 8 ? 8 : m));
}

}  // namespace hqq

namespace hqq {
int gemv_mfma_run(int nbits, int n_layers, const void* x, const void* const* Wq, const void* const* scale, const void* const* zero,
                  const void* const* bias, void* const* y, const int64_t* N, int64_t M, int64_t K, int64_t group_size, hipStream_t st);
bool skinny_covers(int nbits, int64_t M, int64_t K, int64_t group_size, const int64_t* N, int n_layers);
size_t skinny_workspace_bytes(int nbits, int n_layers, const int64_t* N, int64_t M, int64_t K, uint32_t opts);
int skinny_run(int nbits, int n_layers, const void* x, const void* const* Wq, const void* const* scale, const void* const* zero,
               const void* const* bias, void* const* y, const int64_t* N, int64_t M, int64_t K, int dtype, uint32_t opts, void* ws, size_t ws_bytes,
               hipStream_t st);
int gemv_w3s_run(int n_layers, const void* x, const void* const* Wq, const void* const* scale, const void* const* zero, const void* const* bias,
                 void* const* y, const int64_t* N, int64_t M, int64_t K, int64_t group_size, int dtype, uint32_t opts, hipStream_t st);
size_t gemv3_workspace_bytes(int n_layers, const int64_t* N, int64_t M, int64_t K, int64_t group_size, uint32_t opts);
int gemv3_run(int n_layers, const void* x, const void* const* Wq, const void* const* scale, const void* const* zero,
              const void* const* bias, void* const* y, const int64_t* N, int64_t M, int64_t K, int64_t group_size, uint32_t opts,
              void* ws, size_t ws_bytes, hipStream_t st);

// hqq_hip_meta_check: groups whose (zero, scale) cannot take the three-op weight rebuild (decode_common.h, SlabExact<.., SUB>):
// z 2^-J must be exact in fp16, s 2^J finite, |z| <= 2^15 (then q - z cannot overflow either), J = 9 - shift of the row's slab.
template <int NBITS>
__global__ __launch_bounds__(256) void meta_check_kernel(const half_t* __restrict__ scale, const half_t* __restrict__ zero, int64_t R, int G, int rows_per_slab,
                                                         uint32_t* __restrict__ fails) {
  constexpr int PER = 8 / NBITS;
  uint32_t bad = 0;
  for (int64_t r = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; r < R; r += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int n = static_cast<int>(r / G);
    const int slab = n / rows_per_slab;
    const int J = 9 - NBITS * (PER - 1 - slab);
    const half_t dn = static_cast<half_t>(1.0f / static_cast<float>(1 << J)), up = static_cast<half_t>(static_cast<float>(1 << J));
    const half_t z = zero[r], sc = scale[r];
    const half_t zp = z * dn;          // one fp16 rounding, as the kernels do it
    const half_t back = zp * up;       // exact when zp was (a power-of-two scaling up of a representable value)
    const half_t sp = sc * up;
    const float zf = static_cast<float>(z), spf = static_cast<float>(sp);
    const bool ok = (back == z) && (zf <= 32768.0f) && (zf >= -32768.0f) && (spf - spf == 0.0f);   // NaN anywhere fails
    bad += ok ? 0u : 1u;
  }
  if (bad) atomicAdd(fails, bad);   // rare
}
// 3-bit (slab-sharing kernel, gemv3s.hip): group row r sits in slab r / step, whose field lies e(slab) bits above a byte-pair boundary
__global__ __launch_bounds__(256) void meta_check3_kernel(const half_t* __restrict__ scale, const half_t* __restrict__ zero, int64_t R, int64_t step,
                                                          uint32_t* __restrict__ fails) {
  uint32_t bad = 0;
  for (int64_t r = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; r < R; r += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int t = static_cast<int>(r / step);
    const int J = 9 - static_cast<int>((0x0361472503ull >> (4 * t)) & 15ull);   // e of slabs 0..9 = 3,0,5,2,7,4,1,6,3,0 (S3Slab<T>::e)
    const half_t dn = static_cast<half_t>(1.0f / static_cast<float>(1 << J)), up = static_cast<half_t>(static_cast<float>(1 << J));
    const half_t z = zero[r], sc = scale[r];
    const half_t zp = z * dn;
    const half_t back = zp * up;
    const half_t sp = sc * up;
    const float zf = static_cast<float>(z), spf = static_cast<float>(sp);
    const bool ok = (back == z) && (zf <= 32768.0f) && (zf >= -32768.0f) && (spf - spf == 0.0f);
    bad += ok ? 0u : 1u;
  }
  if (bad) atomicAdd(fails, bad);
}
}  // namespace hqq

using namespace hqq;

extern "C" int hqq_hip_gemv_grouped(int nbits, int n_layers, const void* x, const void* const* Wq, const void* const* scale,
                                    const void* const* zero, const void* const* bias, void* const* y, const int64_t* N,
                                    int64_t M, int64_t K, int64_t group_size, int dtype, uint32_t opts, void* workspace, size_t workspace_bytes,
                                    void* stream) {
  clear_stale_error();
  if (opts & ~HQQ_OPT_ALL) { set_error("hqq_hip_gemv: unknown option bits 0x%x", opts & ~HQQ_OPT_ALL); return HQQ_ERR_SHAPE; }
  if (n_layers < 1 || n_layers > HQQ_GEMV_MAX_GROUP) { set_error("hqq_hip_gemv_grouped: n_layers=%d outside [1,%d]", n_layers, HQQ_GEMV_MAX_GROUP); return HQQ_ERR_SHAPE; }
  // 17..64 activation rows: only where the skinny-GEMM kernel (skinny.hip) applies
  if ((opts & HQQ_OPT_W3S) && nbits != 3) { set_error("hqq_hip_gemv: HQQ_OPT_W3S is a 3-bit layout (nbits=%d)", nbits); return HQQ_ERR_SHAPE; }
  const bool w3s = nbits == 3 && (opts & HQQ_OPT_W3S);   // the 3-bit stream layout runs through the 4-bit container's kernels (w3s.h)
  const bool skinny_ok = (N && (dtype == HQQ_F16 || dtype == HQQ_BF16) && skinny_covers(w3s ? 4 : nbits, M, K, group_size, N, n_layers));
  if (M < 1 || M > (skinny_ok ? HQQ_GEMV_MAX_M_SKINNY : HQQ_GEMV_MAX_M)) {
    set_error("hqq_hip_gemv: M=%lld outside [1,%d] (up to %d for fp16, 8-/4-/2-bit, group_size 64, K %% 256 == 0)", (long long)M, HQQ_GEMV_MAX_M, HQQ_GEMV_MAX_M_SKINNY);
    return HQQ_ERR_SHAPE;
  }
  if (K <= 0 || group_size <= 0 || K % group_size) { set_error("hqq_hip_gemv: bad K/group_size"); return HQQ_ERR_SHAPE; }
  if (w3s) {
    if (dtype != HQQ_F16 && dtype != HQQ_BF16) { set_error("hqq_hip_gemv: dtype %d not covered (fp16 / bf16)", dtype); return HQQ_ERR_UNSUPPORTED; }
    if (!x || !Wq || !scale || !zero || !y || !N) { set_error("hqq_hip_gemv: null argument"); return HQQ_ERR_SHAPE; }
    if (!aligned16(x)) { set_error("hqq_hip_gemv: pointers must be 16-byte aligned"); return HQQ_ERR_ALIGN; }
    if (M <= GV_EXACT_ROWWISE_MAX_M) return gemv_w3s_run(n_layers, x, Wq, scale, zero, bias, y, N, M, K, group_size, dtype, opts, as_stream(stream));
    if (!skinny_ok) { set_error("hqq_hip_gemv: 3-bit stream layout: M=%lld beyond %d rows needs group_size 64, K %% 256 == 0, K >= 512", (long long)M, GV_EXACT_ROWWISE_MAX_M); return HQQ_ERR_UNSUPPORTED; }
    for (int i = 0; i < n_layers; ++i) {
      if (N[i] <= 0 || N[i] % 2) { set_error("hqq_hip_gemv: needs N %% 2 == 0 (got N=%lld)", (long long)N[i]); return N[i] <= 0 ? HQQ_ERR_SHAPE : HQQ_ERR_UNSUPPORTED; }
      if (N[i] * (K / group_size) > INT32_MAX) { set_error("hqq_hip_gemv: size overflow"); return HQQ_ERR_SHAPE; }
      if (!Wq[i] || !scale[i] || !zero[i] || !y[i]) { set_error("hqq_hip_gemv: null layer pointer"); return HQQ_ERR_SHAPE; }
      if (!aligned16(Wq[i])) { set_error("hqq_hip_gemv: pointers must be 16-byte aligned"); return HQQ_ERR_ALIGN; }
    }
    return skinny_run(3, n_layers, x, Wq, scale, zero, bias, y, N, M, K, dtype, opts, workspace, workspace_bytes, as_stream(stream));
  }
  if (nbits == 3) {   // int32 containers, ten slabs: its own kernel (gemv3.hip), fp16, exact weights
    if (dtype != HQQ_F16) { set_error("hqq_hip_gemv: the fused 3-bit kernel covers fp16 (got dtype %d)", dtype); return HQQ_ERR_UNSUPPORTED; }
    if (!x || !Wq || !scale || !zero || !y || !N) { set_error("hqq_hip_gemv: null argument"); return HQQ_ERR_SHAPE; }
    if (!aligned16(x)) { set_error("hqq_hip_gemv: pointers must be 16-byte aligned"); return HQQ_ERR_ALIGN; }
    return gemv3_run(n_layers, x, Wq, scale, zero, bias, y, N, M, K, group_size, opts, workspace, workspace_bytes, as_stream(stream));
  }
  if (nbits != 4 && nbits != 2 && nbits != 8 && nbits != 1) { set_error("hqq_hip_gemv: nbits=%d not covered by the fused GEMV", nbits); return HQQ_ERR_UNSUPPORTED; }
  if (dtype != HQQ_F16 && dtype != HQQ_BF16) { set_error("hqq_hip_gemv: dtype %d not covered (fp16 / bf16)", dtype); return HQQ_ERR_UNSUPPORTED; }
  if (dtype == HQQ_BF16 && !skinny_ok && (M > GV_EXACT_ROWWISE_MAX_M || (nbits != 4 && nbits != 2))) {
    set_error("hqq_hip_gemv: bf16 covers nbits 4/2 and M <= %d (got nbits=%d M=%lld)", GV_EXACT_ROWWISE_MAX_M, nbits, (long long)M);
    return HQQ_ERR_UNSUPPORTED;
  }
  if (!x || !Wq || !scale || !zero || !y || !N) { set_error("hqq_hip_gemv: null argument"); return HQQ_ERR_SHAPE; }
  const int per = 8 / nbits;
  if (group_size % 16 || K % 16) { set_error("hqq_hip_gemv: needs group_size %% 16 == 0 (got gs=%lld)", (long long)group_size); return HQQ_ERR_UNSUPPORTED; }
  if (K > INT32_MAX / 2) { set_error("hqq_hip_gemv: size overflow"); return HQQ_ERR_SHAPE; }
  const bool exact = !(opts & HQQ_OPT_FACTORED) || dtype == HQQ_BF16;
  if (n_layers > 1 && !skinny_ok && (dtype == HQQ_F16 || dtype == HQQ_BF16) && (exact ? M > GV_EXACT_ROWWISE_MAX_M : M > 8)) {
    // a group in which only some layers meet the skinny kernel's conditions: launch the layers one by one, so that a layer is
    // served by the same kernel (same summation order, same bits) whether or not it was grouped
    bool any = false;
    for (int i = 0; i < n_layers; ++i) any = any || skinny_covers(nbits, M, K, group_size, N + i, 1);
    if (any) {
      for (int i = 0; i < n_layers; ++i) {
        const void* b1 = bias ? bias[i] : nullptr;
        const int rc = hqq_hip_gemv_grouped(nbits, 1, x, Wq + i, scale + i, zero + i, bias ? &b1 : nullptr, y + i, N + i, M, K, group_size, dtype, opts, workspace, workspace_bytes, stream);
        if (rc) return rc;
      }
      return 0;
    }
  }
  if (exact ? M > GV_EXACT_ROWWISE_MAX_M : (M > 8 && skinny_ok)) {   // (FACTORED: the row-per-wave kernel serves M <= 8 per launch)
    // more activation rows than the row-per-wave kernel contracts cheaply: the 16-row-tile MFMA kernel (needs K % 64 == 0)
    if (K % 64) { set_error("hqq_hip_gemv: M=%lld > %d needs K %% 64 == 0 (got K=%lld)", (long long)M, GV_EXACT_ROWWISE_MAX_M, (long long)K); return HQQ_ERR_UNSUPPORTED; }
    for (int i = 0; i < n_layers; ++i) {
      if (N[i] <= 0 || N[i] % per) { set_error("hqq_hip_gemv: needs N %% %d == 0 (got N=%lld)", per, (long long)N[i]); return N[i] <= 0 ? HQQ_ERR_SHAPE : HQQ_ERR_UNSUPPORTED; }
      if (N[i] * (K / group_size) > INT32_MAX) { set_error("hqq_hip_gemv: size overflow"); return HQQ_ERR_SHAPE; }
      if (!Wq[i] || !scale[i] || !zero[i] || !y[i]) { set_error("hqq_hip_gemv: null layer pointer"); return HQQ_ERR_SHAPE; }
      if (!aligned16(Wq[i])) { set_error("hqq_hip_gemv: pointers must be 16-byte aligned"); return HQQ_ERR_ALIGN; }
    }
    if (!aligned16(x)) { set_error("hqq_hip_gemv: pointers must be 16-byte aligned"); return HQQ_ERR_ALIGN; }
    if (skinny_ok) return skinny_run(nbits, n_layers, x, Wq, scale, zero, bias, y, N, M, K, dtype, opts, workspace, workspace_bytes, as_stream(stream));
    return gemv_mfma_run(nbits, n_layers, x, Wq, scale, zero, bias, y, N, M, K, group_size, as_stream(stream));
  }
  int m_max = max_m_per_launch(K);
  m_max = m_max > (exact ? GV_EXACT_ROWWISE_MAX_M : 8) ? (exact ? GV_EXACT_ROWWISE_MAX_M : 8) : m_max;
  if (m_max < 1) { set_error("hqq_hip_gemv: K=%lld too large to stage one row of x in LDS", (long long)K); return HQQ_ERR_UNSUPPORTED; }
  if (!aligned16(x)) { set_error("hqq_hip_gemv: pointers must be 16-byte aligned"); return HQQ_ERR_ALIGN; }
  GvArgs a;
  int64_t total = 0;
  for (int i = 0; i < n_layers; ++i) {
    if (N[i] <= 0 || N[i] % per) { set_error("hqq_hip_gemv: needs N %% %d == 0 (got N=%lld)", per, (long long)N[i]); return N[i] <= 0 ? HQQ_ERR_SHAPE : HQQ_ERR_UNSUPPORTED; }
    // (the kernel addresses a layer with 32-bit byte offsets from its base pointers: packed weights and meta below 4 GiB per layer)
    if (N[i] * (K / group_size) > INT32_MAX || (N[i] / per) * K > static_cast<int64_t>(UINT32_MAX)) { set_error("hqq_hip_gemv: size overflow"); return HQQ_ERR_SHAPE; }
    if (!Wq[i] || !scale[i] || !zero[i] || !y[i]) { set_error("hqq_hip_gemv: null layer pointer"); return HQQ_ERR_SHAPE; }
    if (!aligned16(Wq[i])) { set_error("hqq_hip_gemv: pointers must be 16-byte aligned"); return HQQ_ERR_ALIGN; }
    total += N[i] / per;
    if (total > INT32_MAX) { set_error("hqq_hip_gemv: size overflow"); return HQQ_ERR_SHAPE; }
    a.Wq[i] = static_cast<const uint8_t*>(Wq[i]);
    a.scale[i] = static_cast<const half_t*>(scale[i]);
    a.zero[i] = static_cast<const half_t*>(zero[i]);
    a.bias[i] = bias ? static_cast<const half_t*>(bias[i]) : nullptr;
    a.y[i] = static_cast<half_t*>(y[i]);
    a.N[i] = static_cast<int>(N[i]);
    a.prow_end[i] = static_cast<int>(total);
  }
  for (int i = n_layers; i < GV_MAXL; ++i) {
    a.Wq[i] = a.Wq[n_layers - 1]; a.scale[i] = a.scale[n_layers - 1]; a.zero[i] = a.zero[n_layers - 1]; a.bias[i] = a.bias[n_layers - 1];
    a.y[i] = a.y[n_layers - 1]; a.N[i] = a.N[n_layers - 1]; a.prow_end[i] = a.prow_end[n_layers - 1];
  }
  a.K = static_cast<int>(K);
  a.gs = static_cast<int>(group_size);
  a.G = static_cast<int>(K / group_size);
  a.total_prow = static_cast<int>(total);
  hipStream_t st = as_stream(stream);
  // x rows beyond the LDS budget of one launch are served by further launches over row blocks of x / y
  for (int64_t m0 = 0; m0 < M; m0 += m_max) {
    const int mm = static_cast<int>(M - m0 < m_max ? M - m0 : m_max);
    GvArgs b = a;
    b.x = static_cast<const half_t*>(x) + m0 * K;
    for (int i = 0; i < GV_MAXL; ++i) b.y[i] = a.y[i] + m0 * a.N[i];
    const int rc = dtype == HQQ_BF16 ? dispatch_bf16(nbits, mm, b, st) : exact ? ((opts & HQQ_OPT_META_SCALABLE) ? dispatch<true, true>(nbits, mm, b, st) : dispatch<true>(nbits, mm, b, st)) : dispatch<false>(nbits, mm, b, st);
    if (rc) return rc;
  }
  return 0;
}

extern "C" int hqq_hip_gemv(int nbits, const void* x, const void* Wq, const void* scale, const void* zero, const void* bias,
                            void* y, int64_t M, int64_t N, int64_t K, int64_t group_size, int dtype, uint32_t opts, void* workspace,
                            size_t workspace_bytes, void* stream) {
  const void* b1[1] = {bias};
  return hqq_hip_gemv_grouped(nbits, 1, x, &Wq, &scale, &zero, bias ? b1 : nullptr, &y, &N, M, K, group_size, dtype, opts, workspace, workspace_bytes, stream);
}

extern "C" size_t hqq_hip_gemv_workspace_bytes(int nbits, int n_layers, const int64_t* N, int64_t M, int64_t K, int64_t group_size, int dtype, uint32_t opts) {
  if (!N || n_layers < 1 || n_layers > HQQ_GEMV_MAX_GROUP || M < 1 || K <= 0 || group_size <= 0) return 0;
  if (nbits == 3 && (opts & HQQ_OPT_W3S)) {
    if (M <= GV_EXACT_ROWWISE_MAX_M) return 0;
    return ((dtype == HQQ_F16 || dtype == HQQ_BF16) && skinny_covers(4, M, K, group_size, N, n_layers)) ? skinny_workspace_bytes(3, n_layers, N, M, K, opts) : 0;
  }
  if (nbits == 3) return gemv3_workspace_bytes(n_layers, N, M, K, group_size, opts);
  if ((dtype == HQQ_F16 || dtype == HQQ_BF16) && skinny_covers(nbits, M, K, group_size, N, n_layers)) return skinny_workspace_bytes(nbits, n_layers, N, M, K, opts);
  if (n_layers > 1 && M > GV_EXACT_ROWWISE_MAX_M) {   // a partly covered group is launched layer by layer (hqq_hip_gemv_grouped)
    size_t most = 0;
    for (int i = 0; i < n_layers; ++i)
      if (skinny_covers(nbits, M, K, group_size, N + i, 1)) { const size_t b = skinny_workspace_bytes(nbits, 1, N + i, M, K, opts); most = b > most ? b : most; }
    return most;
  }
  return 0;
}

extern "C" int hqq_hip_meta_check(int nbits, const void* scale, const void* zero, int64_t N, int64_t K, int64_t group_size, int dtype,
                                  uint32_t* fail_count, void* stream) {
  clear_stale_error();
  if (!scale || !zero || !fail_count) { set_error("hqq_hip_meta_check: null argument"); return HQQ_ERR_SHAPE; }
  if (N <= 0 || K <= 0 || group_size <= 0 || K % group_size) { set_error("hqq_hip_meta_check: bad N/K/group_size"); return HQQ_ERR_SHAPE; }
  if (nbits != 8 && nbits != 4 && nbits != 3 && nbits != 2 && nbits != 1) { set_error("hqq_hip_meta_check: nbits=%d has no three-op rebuild", nbits); return HQQ_ERR_UNSUPPORTED; }
  if (dtype != HQQ_F16) { set_error("hqq_hip_meta_check: the three-op rebuild is an fp16 sequence (dtype %d)", dtype); return HQQ_ERR_UNSUPPORTED; }
  if (nbits == 3) {
    const int64_t R3 = N * (K / group_size);
    if (R3 > INT32_MAX) { set_error("hqq_hip_meta_check: size overflow"); return HQQ_ERR_SHAPE; }
    hipStream_t st3 = as_stream(stream);
    hipError_t e3 = hipMemsetAsync(fail_count, 0, sizeof(uint32_t), st3);
    if (e3 != hipSuccess) { set_error("hqq_hip_meta_check: hipMemsetAsync: %s", hipGetErrorString(e3)); return static_cast<int>(e3); }
    const int grid3 = static_cast<int>((R3 + 255) / 256 > 2048 ? 2048 : (R3 + 255) / 256);
    hipLaunchKernelGGL(meta_check3_kernel, dim3(grid3), dim3(256), 0, st3, static_cast<const half_t*>(scale), static_cast<const half_t*>(zero), R3, (R3 + 9) / 10, fail_count);
    return check_launch("hqq_hip_meta_check");
  }
  const int per = 8 / nbits;
  if (N % per) { set_error("hqq_hip_meta_check: N must divide by %d", per); return HQQ_ERR_SHAPE; }
  const int64_t G = K / group_size, R = N * G;
  if (R > INT32_MAX) { set_error("hqq_hip_meta_check: size overflow"); return HQQ_ERR_SHAPE; }
  hipStream_t st = as_stream(stream);
  hipError_t e = hipMemsetAsync(fail_count, 0, sizeof(uint32_t), st);
  if (e != hipSuccess) { set_error("hqq_hip_meta_check: hipMemsetAsync: %s", hipGetErrorString(e)); return static_cast<int>(e); }
  const int grid = static_cast<int>((R + 255) / 256 > 2048 ? 2048 : (R + 255) / 256);
  const half_t* sp = static_cast<const half_t*>(scale);
  const half_t* zp = static_cast<const half_t*>(zero);
  switch (nbits) {
    case 8: hipLaunchKernelGGL(meta_check_kernel<8>, dim3(grid), dim3(256), 0, st, sp, zp, R, static_cast<int>(G), static_cast<int>(N / per), fail_count); break;
    case 4: hipLaunchKernelGGL(meta_check_kernel<4>, dim3(grid), dim3(256), 0, st, sp, zp, R, static_cast<int>(G), static_cast<int>(N / per), fail_count); break;
    case 2: hipLaunchKernelGGL(meta_check_kernel<2>, dim3(grid), dim3(256), 0, st, sp, zp, R, static_cast<int>(G), static_cast<int>(N / per), fail_count); break;
    case 1: hipLaunchKernelGGL(meta_check_kernel<1>, dim3(grid), dim3(256), 0, st, sp, zp, R, static_cast<int>(G), static_cast<int>(N / per), fail_count); break;
  }
  return check_launch("hqq_hip_meta_check");
}

// every group of a layer in the 3-bit stream layout: zero 2^-9 exact, scale 2^9 finite (the largest J of w3s.h's three field offsets;
// the smaller ones follow) — meta_check_kernel<8> checks exactly J = 9 for every row
extern "C" int hqq_hip_w3s_meta_check(const void* scale, const void* zero, int64_t N, int64_t K, uint32_t* fail_count, void* stream) {
  clear_stale_error();
  if (!scale || !zero || !fail_count) { set_error("hqq_hip_w3s_meta_check: null argument"); return HQQ_ERR_SHAPE; }
  if (N <= 0 || K <= 0 || K % 64) { set_error("hqq_hip_w3s_meta_check: bad N/K"); return HQQ_ERR_SHAPE; }
  const int64_t G = K / 64, R = N * G;
  if (R > INT32_MAX) { set_error("hqq_hip_w3s_meta_check: size overflow"); return HQQ_ERR_SHAPE; }
  hipStream_t st = as_stream(stream);
  hipError_t e = hipMemsetAsync(fail_count, 0, sizeof(uint32_t), st);
  if (e != hipSuccess) { set_error("hqq_hip_w3s_meta_check: hipMemsetAsync: %s", hipGetErrorString(e)); return static_cast<int>(e); }
  const int grid = static_cast<int>((R + 255) / 256 > 2048 ? 2048 : (R + 255) / 256);
  hipLaunchKernelGGL(meta_check_kernel<8>, dim3(grid), dim3(256), 0, st, static_cast<const half_t*>(scale), static_cast<const half_t*>(zero), R, static_cast<int>(G),
                     static_cast<int>(N), fail_count);
  return check_launch("hqq_hip_w3s_meta_check");
}
