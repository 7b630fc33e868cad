// gemv_block.hip — the decoder block's launches with the steps either side of them folded in: the decode kernel's text (gemv_kernel.inc,
// gemv.hip's graded kernel) compiled with an RMSNorm prologue, a residual-add epilogue and a SiLU * up epilogue.  One activation row, fp16 /
// bf16, 4- / 3-bit (stream layout) / 2-bit, group_size 64, exact weights.  gfx950.  (round 5.)
//
// What it replaces: in the reference's generate loop (hqq/utils/generation_hf.py:117-540, Readme.md:153) HF's LlamaDecoderLayer.forward
// (transformers models/llama/modeling_llama.py) around HQQLinear.forward — input_layernorm -> q|k|v -> rotary / cache / attention -> o ->
// residual add -> post_attention_layernorm -> gate|up -> act_fn(gate) * up -> down -> residual add.  Round 4 ran that as 9 kernels per block
// (csrc/block.hip: add_rmsnorm x 2, rope_cache, silu_mul + 4 GEMV launches + attention); every glue launch is a dependent boundary of
// ~2.3 us around ~0.2 us of arithmetic.  Here:
//   q|k|v      prologue: LlamaRMSNorm(h) * w1 computed by every workgroup while its first weights are in flight   (HQQ_BLOCK_NORM)
//              epilogue (q and k in the rotary-paired row order, hqq_amd.ops.rotary_pair_layout): apply_rotary_pos_emb on q and k,
//              k / v written into the static cache at the position                                              (... | HQQ_BLOCK_ROPE)
//   o          epilogue: h[n] += y[n]                                                                             (HQQ_BLOCK_RESID)
//   gate|up    ONE paired layer (hqq_amd.ops.pair_layers: packed row p = gate row p | up row p), prologue as q|k|v,
//              epilogue: a[n] = silu(gate[n]) * up[n] — gate and up never reach memory                           (HQQ_BLOCK_NORM | HQQ_BLOCK_SILU)
//   down       epilogue: h[n] += y[n]                                                                             (HQQ_BLOCK_RESID)
// = 4 launches + attention per block (5 with hqq_hip_rope_cache as a launch of its own).  Arithmetic: block.hip's, rounding for rounding (block_math.h; this file is compiled with
// -ffp-contract=off like block.hip); the norm's fp32 sum of squares has its own fixed order (thread's elements, wave_sum, waves in order).
// The streaming loop is the decode kernel's, unchanged: each variant is its own compilation of the kernel text (gemv_kernel.inc).
#include <type_traits>

#include "gemv_shared.h"
#include "w3s.h"
#include "block_math.h"

#ifndef GB_NBITS
#error "compile with -DGB_NBITS=4|3|2 (Makefile)"
#endif

namespace hqq {

// names the kernel text refers to in branches this file never instantiates (the factored arithmetic)
template <int NBITS, int M, int S, int PER> struct SlabLoop;
template <int NBITS, int S, int PER> struct GroupConst;

#define GV_KERNEL_ROPE 0
// ---- RMSNorm prologue (one / two passes of the workgroup over the row) ----
#define GV_KERNEL_XPASS2 0
#define GV_KERNEL_RESID 0
#define GV_KERNEL_SILU 0
#define GV_KERNEL_NORM 1
#define GV_KERNEL_NAME gb_norm1_kernel
#include "gemv_kernel.inc"
#undef GV_KERNEL_NAME
#undef GV_KERNEL_NORM
#define GV_KERNEL_NORM 2
#define GV_KERNEL_NAME gb_norm2_kernel
#include "gemv_kernel.inc"
#undef GV_KERNEL_NAME
#undef GV_KERNEL_NORM
// ---- RMSNorm prologue + rotary / cache-write epilogue (q|k|v with q, k in the rotary-paired row order) ----
#undef GV_KERNEL_ROPE
#define GV_KERNEL_ROPE 1
#define GV_KERNEL_NORM 1
#define GV_KERNEL_NAME gb_norm1_rope_kernel
#include "gemv_kernel.inc"
#undef GV_KERNEL_NAME
#undef GV_KERNEL_NORM
#define GV_KERNEL_NORM 2
#define GV_KERNEL_NAME gb_norm2_rope_kernel
#include "gemv_kernel.inc"
#undef GV_KERNEL_NAME
#undef GV_KERNEL_NORM
#undef GV_KERNEL_ROPE
#define GV_KERNEL_ROPE 0
#undef GV_KERNEL_SILU
// ---- RMSNorm prologue + SiLU * up epilogue (the paired gate|up layer) ----
#define GV_KERNEL_SILU 1
#define GV_KERNEL_NORM 1
#define GV_KERNEL_NAME gb_norm1_silu_kernel
#include "gemv_kernel.inc"
#undef GV_KERNEL_NAME
#undef GV_KERNEL_NORM
#define GV_KERNEL_NORM 2
#define GV_KERNEL_NAME gb_norm2_silu_kernel
#include "gemv_kernel.inc"
#undef GV_KERNEL_NAME
#undef GV_KERNEL_NORM
#undef GV_KERNEL_SILU
#undef GV_KERNEL_RESID
// ---- residual-add epilogue (o, down; rows of x of one or several passes) ----
#define GV_KERNEL_NORM 0
#define GV_KERNEL_SILU 0
#define GV_KERNEL_RESID 1
#define GV_KERNEL_NAME gb_resid_kernel
#include "gemv_kernel.inc"
#undef GV_KERNEL_NAME
#undef GV_KERNEL_XPASS2
#define GV_KERNEL_XPASS2 1
#define GV_KERNEL_NAME gb_resid_xp2_kernel
#include "gemv_kernel.inc"
#undef GV_KERNEL_NAME
#undef GV_KERNEL_XPASS2
#undef GV_KERNEL_RESID
#undef GV_KERNEL_SILU
#undef GV_KERNEL_NORM
#undef GV_KERNEL_ROPE

namespace gb {

static int num_cus() {
  static int n_cus = 0;
  if (n_cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) n_cus = n;
    else n_cus = 256;
  }
  return n_cus;
}

struct Extra { const half_t* norm_w; float eps; GbRope rope; };

// launch geometry as gemv.hip's launch_gemv_f16 (the same persistent grid, the same 8 x 2 shape for single layers of at most one packed
// row per wave, the same few-rows / long-K row sharing): the folded variants must not stream differently from the kernels they replace
template <int KIND /* 0 norm, 1 norm + silu, 2 resid, 3 norm + rope */, bool BF16, bool SUB, int WPG = GV_WAVES>
static int launch(const GvArgs& args, const Extra& ex, hipStream_t st) {
  constexpr int NB = GB_NBITS;
  constexpr int PER = NB == 3 ? 2 : 8 / NB;
  constexpr int WG_PER_CU = GV_WG_PER_CU * GV_WAVES / WPG;
  if constexpr (WPG == GV_WAVES && KIND == 2 && !BF16) {
    if (args.prow_end[0] == args.total_prow && args.total_prow <= num_cus() * 8 && args.total_prow * 2 > num_cus() * 8 && args.K >= GV_UNIT)
      return launch<KIND, BF16, SUB, 8>(args, ex, st);
  }
  GvArgs a = args;
  const int nsteps = (a.K + GV_KSTEP - 1) / GV_KSTEP;
  const int nunits = (nsteps + GV_U - 1) / GV_U;
  const size_t xs_bytes = static_cast<size_t>(nsteps) * (GV_KSTEP * 2 + 64 * 4);
  a.red_off = static_cast<int>((xs_bytes + 15) & ~static_cast<size_t>(15));
  const size_t lds = a.red_off + sizeof(float) * WPG * (PER > 1 ? PER : 2);   // K-split reduction buffer; the norm parks WPG partial sums there
  const bool two_pass = nsteps * 64 > WPG * 64;                              // the row of x is longer than one pass of the workgroup's threads
  if (KIND != 2 && nsteps * 64 > 2 * WPG * 64) { set_error("hqq_hip_gemv_block: the folded RMSNorm covers K <= %d", 32 * WPG * 64); return HQQ_ERR_UNSUPPORTED; }
  const void* kern;
  int variant = two_pass ? 1 : 0;
  if constexpr (KIND == 0) kern = two_pass ? reinterpret_cast<const void*>(gb_norm2_kernel<NB, 1, true, true, BF16, SUB, WPG>) : reinterpret_cast<const void*>(gb_norm1_kernel<NB, 1, true, true, BF16, SUB, WPG>);
  else if constexpr (KIND == 3) kern = two_pass ? reinterpret_cast<const void*>(gb_norm2_rope_kernel<NB, 1, true, true, BF16, SUB, WPG>) : reinterpret_cast<const void*>(gb_norm1_rope_kernel<NB, 1, true, true, BF16, SUB, WPG>);
  else if constexpr (KIND == 1) kern = two_pass ? reinterpret_cast<const void*>(gb_norm2_silu_kernel<NB, 1, true, true, BF16, SUB, WPG>) : reinterpret_cast<const void*>(gb_norm1_silu_kernel<NB, 1, true, true, BF16, SUB, WPG>);
  else kern = two_pass ? reinterpret_cast<const void*>(gb_resid_xp2_kernel<NB, 1, true, true, BF16, SUB, WPG>) : reinterpret_cast<const void*>(gb_resid_kernel<NB, 1, true, true, BF16, SUB, WPG>);
  int per_cu = static_cast<int>(160 * 1024 / (lds + 256));
  per_cu = per_cu > WG_PER_CU ? WG_PER_CU : (per_cu < 1 ? 1 : per_cu);
  {
    static int by_regs_v[2] = {0, 0};   // per instantiation and kernel variant: registers bound the residency too
    int& by_regs = by_regs_v[variant];
    if (by_regs == 0) {
      hipFuncAttributes fa;
      by_regs = WG_PER_CU;
      if (hipFuncGetAttributes(&fa, kern) == hipSuccess && fa.numRegs > 0) {
        const int regs = (fa.numRegs + 7) & ~7;
        by_regs = (512 / regs) * 4 / WPG;
        by_regs = by_regs < 1 ? 1 : by_regs;
      } else {
        (void)hipGetLastError();
      }
    }
    per_cu = per_cu > by_regs ? by_regs : per_cu;
  }
  const int cap = num_cus() * per_cu;
  a.ksplit = (nunits >= WPG && static_cast<int64_t>(a.total_prow) * 4 <= static_cast<int64_t>(num_cus()) * WG_PER_CU * WPG) ? 1 : 0;
  const int tiles = a.ksplit ? a.total_prow : (a.total_prow + WPG - 1) / WPG;
  const int grid = tiles < cap ? tiles : cap;
  if (lds > 64 * 1024) {
    static LdsRaised raised[2];
    if (const int rc = raise_lds_limit(raised[variant], kern, GV_LDS_MAX, "hqq_hip_gemv_block")) return rc;
  }
  GvIn in;
  GvOut out;
  for (int i = 0; i < GV_MAXL; ++i) {
    in.Wq[i] = a.Wq[i]; in.scale[i] = a.scale[i]; in.zero[i] = a.zero[i]; in.N[i] = a.N[i]; in.prow_end[i] = a.prow_end[i];
    out.bias[i] = nullptr; out.y[i] = a.y[i];
  }
  in.x = a.x; in.K = a.K; in.gs = a.gs; in.G = a.G; in.total_prow = a.total_prow; in.red_off = a.red_off; in.ksplit = a.ksplit;
  if constexpr (KIND == 0) {
    if (two_pass) hipLaunchKernelGGL((gb_norm2_kernel<NB, 1, true, true, BF16, SUB, WPG>), dim3(grid), dim3(WPG * 64), lds, st, GV_IN_ARGS(in), out, ex.norm_w, ex.eps);
    else hipLaunchKernelGGL((gb_norm1_kernel<NB, 1, true, true, BF16, SUB, WPG>), dim3(grid), dim3(WPG * 64), lds, st, GV_IN_ARGS(in), out, ex.norm_w, ex.eps);
  } else if constexpr (KIND == 3) {
    if (two_pass) hipLaunchKernelGGL((gb_norm2_rope_kernel<NB, 1, true, true, BF16, SUB, WPG>), dim3(grid), dim3(WPG * 64), lds, st, GV_IN_ARGS(in), out, ex.norm_w, ex.eps, ex.rope);
    else hipLaunchKernelGGL((gb_norm1_rope_kernel<NB, 1, true, true, BF16, SUB, WPG>), dim3(grid), dim3(WPG * 64), lds, st, GV_IN_ARGS(in), out, ex.norm_w, ex.eps, ex.rope);
  } else if constexpr (KIND == 1) {
    if (two_pass) hipLaunchKernelGGL((gb_norm2_silu_kernel<NB, 1, true, true, BF16, SUB, WPG>), dim3(grid), dim3(WPG * 64), lds, st, GV_IN_ARGS(in), out, ex.norm_w, ex.eps);
    else hipLaunchKernelGGL((gb_norm1_silu_kernel<NB, 1, true, true, BF16, SUB, WPG>), dim3(grid), dim3(WPG * 64), lds, st, GV_IN_ARGS(in), out, ex.norm_w, ex.eps);
  } else {
    if (two_pass) hipLaunchKernelGGL((gb_resid_xp2_kernel<NB, 1, true, true, BF16, SUB, WPG>), dim3(grid), dim3(WPG * 64), lds, st, GV_IN_ARGS(in), out);
    else hipLaunchKernelGGL((gb_resid_kernel<NB, 1, true, true, BF16, SUB, WPG>), dim3(grid), dim3(WPG * 64), lds, st, GV_IN_ARGS(in), out);
  }
  return check_launch("hqq_hip_gemv_block");
}

template <int KIND>
static int by_dtype(const GvArgs& a, const Extra& ex, int dtype, uint32_t opts, hipStream_t st) {
  if (dtype == HQQ_BF16) return launch<KIND, true, false>(a, ex, st);
  if (opts & HQQ_OPT_META_SCALABLE) return launch<KIND, false, true>(a, ex, st);
  return launch<KIND, false, false>(a, ex, st);
}

}  // namespace gb

#define GB_CAT2(a, b) a##b
#define GB_CAT(a, b) GB_CAT2(a, b)
// one object per bit width (Makefile): gemv_block_run_4 / _3 / _2.  kind: 0 norm, 1 norm + silu, 2 resid, 3 norm + rope
int GB_CAT(gemv_block_run_, GB_NBITS)(int kind, const GvArgs& a, const half_t* norm_w, float eps, const GbRope& rope, int dtype, uint32_t opts, hipStream_t st) {
  const gb::Extra ex{norm_w, eps, rope};
  switch (kind) {
    case 0: return gb::by_dtype<0>(a, ex, dtype, opts, st);
    case 1: return gb::by_dtype<1>(a, ex, dtype, opts, st);
    case 3: return gb::by_dtype<3>(a, ex, dtype, opts, st);
    default: return gb::by_dtype<2>(a, ex, dtype, opts, st);
  }
}

}  // namespace hqq

#if GB_NBITS == 4
namespace hqq {
int gemv_block_run_3(int kind, const GvArgs& a, const half_t* norm_w, float eps, const GbRope& rope, int dtype, uint32_t opts, hipStream_t st);
int gemv_block_run_2(int kind, const GvArgs& a, const half_t* norm_w, float eps, const GbRope& rope, int dtype, uint32_t opts, hipStream_t st);
}
using namespace hqq;

extern "C" int hqq_hip_gemv_block(int nbits, int n_layers, const void* x, const void* norm_weight, float eps, const void* const* Wq, const void* const* scale,
                                  const void* const* zero, void* const* y, const int64_t* N, int64_t K, int64_t group_size, int dtype, uint32_t opts, uint32_t flags,
                                  const hqq_rope_t* rope_args, void* stream) {
  clear_stale_error();
  if (opts & ~HQQ_OPT_ALL) { set_error("hqq_hip_gemv_block: unknown option bits 0x%x", opts & ~HQQ_OPT_ALL); return HQQ_ERR_SHAPE; }
  const bool norm = flags & HQQ_BLOCK_NORM, resid = flags & HQQ_BLOCK_RESID, silu = flags & HQQ_BLOCK_SILU, rope = flags & HQQ_BLOCK_ROPE;
  if ((flags & ~(HQQ_BLOCK_NORM | HQQ_BLOCK_RESID | HQQ_BLOCK_SILU | HQQ_BLOCK_ROPE)) || !(norm || resid) || (resid && (norm || silu || rope)) || ((silu || rope) && !norm) || (silu && rope)) {
    set_error("hqq_hip_gemv_block: flags 0x%x: HQQ_BLOCK_NORM, HQQ_BLOCK_NORM | HQQ_BLOCK_ROPE, HQQ_BLOCK_NORM | HQQ_BLOCK_SILU or HQQ_BLOCK_RESID", flags);
    return HQQ_ERR_SHAPE;
  }
  if (n_layers < 1 || n_layers > HQQ_GEMV_MAX_GROUP || ((resid || silu) && n_layers != 1) || (rope && n_layers != 3)) {
    set_error("hqq_hip_gemv_block: n_layers=%d (the residual / SiLU epilogues serve ONE layer, the rotary one q | k | v)", n_layers);
    return HQQ_ERR_SHAPE;
  }
  GbRope gr{nullptr, nullptr, nullptr, 0, 0};
  if (rope) {
    if (!rope_args || !rope_args->cos || !rope_args->sin || !rope_args->pos || rope_args->head_dim < 2 || rope_args->head_dim % 2 || rope_args->head_dim > 4096 || rope_args->cache_len < 1 ||
        rope_args->cache_len > INT32_MAX) { set_error("hqq_hip_gemv_block: HQQ_BLOCK_ROPE needs cos / sin / pos, an even head_dim and a cache length"); return HQQ_ERR_SHAPE; }
    if (!N || N[0] % rope_args->head_dim || N[1] % rope_args->head_dim || N[1] != N[2]) { set_error("hqq_hip_gemv_block: q / k / v widths must be whole heads, k and v alike"); return HQQ_ERR_SHAPE; }
    gr = GbRope{static_cast<const uint16_t*>(rope_args->cos), static_cast<const uint16_t*>(rope_args->sin), rope_args->pos, static_cast<int>(rope_args->head_dim), static_cast<int>(rope_args->cache_len)};
  }
  if (!x || !Wq || !scale || !zero || !y || !N || (norm && !norm_weight)) { set_error("hqq_hip_gemv_block: null argument"); return HQQ_ERR_SHAPE; }
  if (dtype != HQQ_F16 && dtype != HQQ_BF16) { set_error("hqq_hip_gemv_block: dtype %d not covered (fp16 / bf16)", dtype); return HQQ_ERR_UNSUPPORTED; }
  const bool w3s = nbits == 3 && (opts & HQQ_OPT_W3S);
  if ((nbits != 4 && nbits != 2 && !w3s) || group_size != 64 || K <= 0 || K % 64 || (opts & HQQ_OPT_FACTORED)) {
    set_error("hqq_hip_gemv_block: covers 4- / 2-bit and the 3-bit stream layout, group_size 64, exact weights (nbits=%d gs=%lld)", nbits, (long long)group_size);
    return HQQ_ERR_UNSUPPORTED;
  }
  if (K > INT32_MAX / 2 || K * 2 > GV_LDS_MAX - 4096) { set_error("hqq_hip_gemv_block: K=%lld too large to stage one row in LDS", (long long)K); return HQQ_ERR_UNSUPPORTED; }
  if (!aligned16(x) || (norm && !aligned16(norm_weight))) { set_error("hqq_hip_gemv_block: pointers must be 16-byte aligned"); return HQQ_ERR_ALIGN; }
  const int per = w3s ? 2 : 8 / nbits;
  GvArgs a;
  int64_t total = 0;
  for (int i = 0; i < n_layers; ++i) {
    if (N[i] <= 0 || N[i] % per || (silu && N[i] % (2 * per))) { set_error("hqq_hip_gemv_block: N=%lld does not divide into %d slabs%s", (long long)N[i], per, silu ? " of a paired layer" : ""); return N[i] <= 0 ? HQQ_ERR_SHAPE : HQQ_ERR_UNSUPPORTED; }
    if (N[i] * (K / 64) > INT32_MAX || (N[i] / per) * K > static_cast<int64_t>(UINT32_MAX)) { set_error("hqq_hip_gemv_block: size overflow"); return HQQ_ERR_SHAPE; }
    if (!Wq[i] || !scale[i] || !zero[i] || !y[i]) { set_error("hqq_hip_gemv_block: null layer pointer"); return HQQ_ERR_SHAPE; }
    if (!aligned16(Wq[i])) { set_error("hqq_hip_gemv_block: pointers must be 16-byte aligned"); return HQQ_ERR_ALIGN; }
    total += N[i] / per;
    if (total > INT32_MAX) { set_error("hqq_hip_gemv_block: size overflow"); return HQQ_ERR_SHAPE; }
    a.Wq[i] = static_cast<const uint8_t*>(Wq[i]);
    a.scale[i] = static_cast<const half_t*>(scale[i]);
    a.zero[i] = static_cast<const half_t*>(zero[i]);
    a.bias[i] = nullptr;
    a.y[i] = static_cast<half_t*>(y[i]);
    a.N[i] = static_cast<int>(N[i]);
    a.prow_end[i] = static_cast<int>(total);
  }
  for (int i = n_layers; i < GV_MAXL; ++i) {
    a.Wq[i] = a.Wq[n_layers - 1]; a.scale[i] = a.scale[n_layers - 1]; a.zero[i] = a.zero[n_layers - 1]; a.bias[i] = nullptr;
    a.y[i] = a.y[n_layers - 1]; a.N[i] = a.N[n_layers - 1]; a.prow_end[i] = a.prow_end[n_layers - 1];
  }
  a.x = static_cast<const half_t*>(x);
  a.K = static_cast<int>(K);
  a.gs = 64;
  a.G = static_cast<int>(K / 64);
  a.total_prow = static_cast<int>(total);
  const int kind = resid ? 2 : (silu ? 1 : (rope ? 3 : 0));
  const half_t* nw = static_cast<const half_t*>(norm_weight);
  hipStream_t st = as_stream(stream);
  switch (nbits) {
    case 4: return gemv_block_run_4(kind, a, nw, eps, gr, dtype, opts, st);
    case 3: return gemv_block_run_3(kind, a, nw, eps, gr, dtype, opts, st);
    default: return gemv_block_run_2(kind, a, nw, eps, gr, dtype, opts, st);
  }
}
#endif
