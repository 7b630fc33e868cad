// gemv_w3s.hip — the decode kernel (gemv_kernel.inc, the text of gemv.hip's graded kernel) compiled for 3-bit layers in the STREAM layout
// of this build (w3s.h): fused unpack -> dequantize -> GEMV for HQQLinear.forward with 1..4 activation rows, fp16 and bf16, gfx950.
//
// Replaces, for axis=1 3-bit layers, BitPack.unpack_3bit_32 -> (W_r - zero) * scale -> torch.matmul(x, W.t()) (+ bias)
//   (hqq/core/bitpack.py:95-110, hqq/core/quantize.py:183-199, :880-898) by ONE pass over N K 3/8 bytes of levels + the group constants.
// The reference container (ten unrelated row slabs per int32) needed three kernels, fp32 scratch partials and a second launch
// (gemv3.hip / gemv3s.hip: 0.16 of the HBM roofline on the 7B stack); HQQLinearHIP re-lays the levels out when a layer is patched
// (hqq_hip_w3s_pack) — as every optimised backend of the reference does (hqq/backends/torchao.py:202-241, marlin.py:74-123) — and
// the layer then streams through the same row-per-wave structure as a 4-bit one: two row slabs per packed row, 12 bytes per lane and load.
// Same weights bit for bit (round16(round16(q - z) * s)); compiled as its own translation unit so that gemv.hip's code stays as it is.
#include <type_traits>

#include "gemv_shared.h"
#include "w3s.h"

namespace hqq {

// names the kernel text refers to in branches this file never instantiates (other bit widths, the factored arithmetic)
template <int NBITS, int M, int S, int PER> struct SlabLoop;
template <int NBITS, int S, int PER> struct GroupConst;
template <int NBITS, int M, int S, int PER> struct SlabExactBF16;
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

#define GV_KERNEL_NAME gemv_w3s_kernel
#define GV_KERNEL_XPASS2 0
#include "gemv_kernel.inc"
#undef GV_KERNEL_NAME
#undef GV_KERNEL_XPASS2
#define GV_KERNEL_NAME gemv_w3s_xp2_kernel
#define GV_KERNEL_XPASS2 1
#include "gemv_kernel.inc"
#undef GV_KERNEL_NAME
#undef GV_KERNEL_XPASS2

static int w3s_num_cus() {
  static int n_cus = 0;
  if (n_cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) n_cus = n;
    else n_cus = 256;
  }
  return n_cus;
}

// launch geometry as gemv.hip's launch_gemv_f16: a persistent grid of <= 16 waves per CU striding over the group's packed rows; single
// layers of at most one packed row per wave of the 8 x 2 shape take 8 waves per workgroup; few rows x long K share a row between the
// waves of a workgroup (the choice depends on the layer shapes only, never on M)
template <int M, bool BF16, bool SUB, int WPG = GV_WAVES>
static int launch_w3s(const GvArgs& args, hipStream_t st) {
  constexpr int WG_PER_CU = GV_WG_PER_CU * GV_WAVES / WPG;
  if constexpr (WPG == GV_WAVES && M == 1 && !BF16) {
    if (args.prow_end[0] == args.total_prow && args.total_prow <= w3s_num_cus() * 8 && args.total_prow * 2 > w3s_num_cus() * 8 && args.K >= GV_UNIT)
      return launch_w3s<M, BF16, SUB, 8>(args, st);
  }
  constexpr int PER = 2;
  GvArgs a = args;
  const int nsteps = (a.K + GV_KSTEP - 1) / GV_KSTEP;
  const int nunits = (nsteps + GV_U - 1) / GV_U;
  const size_t xs_bytes = static_cast<size_t>(M) * nsteps * (GV_KSTEP * 2 + 64 * 4);
  a.red_off = static_cast<int>((xs_bytes + 15) & ~static_cast<size_t>(15));
  const size_t lds = a.red_off + sizeof(float) * WPG * M * PER;
  auto kern = gemv_w3s_kernel<3, M, true, true, BF16, SUB, WPG>;
  int variant = 0;
  if (nsteps * 64 > WPG * 64) { kern = gemv_w3s_xp2_kernel<3, M, true, true, BF16, SUB, WPG>; variant = 1; }
  int per_cu = static_cast<int>(160 * 1024 / (lds + 256));
  per_cu = per_cu > WG_PER_CU ? WG_PER_CU : (per_cu < 1 ? 1 : per_cu);
  {
    static int by_regs_v[2] = {0, 0};   // per instantiation and kernel variant: registers bound the residency too
    int& by_regs = by_regs_v[variant];
    if (by_regs == 0) {
      hipFuncAttributes fa;
      by_regs = WG_PER_CU;
      if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kern)) == hipSuccess && fa.numRegs > 0) {
        const int regs = (fa.numRegs + 7) & ~7;
        by_regs = (512 / regs) * 4 / WPG;
        by_regs = by_regs < 1 ? 1 : by_regs;
      } else {
        (void)hipGetLastError();
      }
    }
    per_cu = per_cu > by_regs ? by_regs : per_cu;
  }
  const int cap = w3s_num_cus() * per_cu;
  a.ksplit = (nunits >= WPG && static_cast<int64_t>(a.total_prow) * 4 <= static_cast<int64_t>(w3s_num_cus()) * WG_PER_CU * WPG) ? 1 : 0;
  const int tiles = a.ksplit ? a.total_prow : (a.total_prow + WPG - 1) / WPG;
  const int grid = tiles < cap ? tiles : cap;
  if (lds > 64 * 1024) {
    static LdsRaised raised[2];
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

template <bool BF16, bool SUB>
static int w3s_dispatch_m(int M, const GvArgs& a, hipStream_t st) {
  switch (M) {
    case 1: return launch_w3s<1, BF16, SUB>(a, st);
    case 2: return launch_w3s<2, BF16, SUB>(a, st);
    case 3: return launch_w3s<3, BF16, SUB>(a, st);
    case 4: return launch_w3s<4, BF16, SUB>(a, st);
  }
  return HQQ_ERR_SHAPE;
}

// rows of x one launch can stage (LDS budget), as gemv.hip
static int w3s_max_m(int64_t K) {
  const int64_t nsteps = (K + GV_KSTEP - 1) / GV_KSTEP;
  const int64_t m = GV_LDS_MAX / (nsteps * (GV_KSTEP * 2 + 64 * 4));
  return static_cast<int>(m < 1 ? 0 : (m > 4 ? 4 : m));
}

// hqq_hip_gemv_grouped for nbits = 3 with HQQ_OPT_W3S, 1 <= M <= 4 (arguments validated by the caller up to what is checked here)
int gemv_w3s_run(int n_layers, const void* x, const void* const* Wq, const void* const* scale, const void* const* zero, const void* const* bias,
                 void* const* y, const int64_t* N, int64_t M, int64_t K, int64_t group_size, int dtype, uint32_t opts, hipStream_t st) {
  if (group_size != 64 || K % 64) { set_error("hqq_hip_gemv: the 3-bit stream layout covers group_size 64, K %% 64 == 0"); return HQQ_ERR_UNSUPPORTED; }
  if (K > INT32_MAX / 2) { set_error("hqq_hip_gemv: size overflow"); return HQQ_ERR_SHAPE; }
  const int m_max = w3s_max_m(K);
  if (m_max < 1) { set_error("hqq_hip_gemv: K=%lld too large to stage one row of x in LDS", (long long)K); return HQQ_ERR_UNSUPPORTED; }
  GvArgs a;
  int64_t total = 0;
  for (int i = 0; i < n_layers; ++i) {
    if (N[i] <= 0 || N[i] % 2) { set_error("hqq_hip_gemv: the 3-bit stream layout needs N %% 2 == 0 (got N=%lld)", (long long)N[i]); return N[i] <= 0 ? HQQ_ERR_SHAPE : HQQ_ERR_UNSUPPORTED; }
    if (N[i] * (K / 64) > INT32_MAX || (N[i] / 2) * (K / 4) * 3 > static_cast<int64_t>(UINT32_MAX)) { set_error("hqq_hip_gemv: size overflow"); return HQQ_ERR_SHAPE; }
    if (!Wq[i] || !scale[i] || !zero[i] || !y[i]) { set_error("hqq_hip_gemv: null layer pointer"); return HQQ_ERR_SHAPE; }
    if (!aligned16(Wq[i])) { set_error("hqq_hip_gemv: pointers must be 16-byte aligned"); return HQQ_ERR_ALIGN; }
    total += N[i] / 2;
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
  a.gs = 64;
  a.G = static_cast<int>(K / 64);
  a.total_prow = static_cast<int>(total);
  const bool sub = (opts & HQQ_OPT_META_SCALABLE) && dtype == HQQ_F16;
  for (int64_t m0 = 0; m0 < M; m0 += m_max) {
    const int mm = static_cast<int>(M - m0 < m_max ? M - m0 : m_max);
    GvArgs b = a;
    b.x = static_cast<const half_t*>(x) + m0 * K;
    for (int i = 0; i < GV_MAXL; ++i) b.y[i] = a.y[i] + m0 * a.N[i];
    const int rc = dtype == HQQ_BF16 ? w3s_dispatch_m<true, false>(mm, b, st) : (sub ? w3s_dispatch_m<false, true>(mm, b, st) : w3s_dispatch_m<false, false>(mm, b, st));
    if (rc) return rc;
  }
  return 0;
}

}  // namespace hqq
