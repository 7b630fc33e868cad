// quantize.hip — Quantizer.quantize with the half-quadratic proximal solver, fused with bit-packing (gfx950).
//
// Reference: hqq/core/quantize.py:75-180 (min/max init, scale inversion, meta), hqq/core/optimize.py:96-108
// (shrink_lp_op), :201-206 (one step), :208-255 (optimize_weights_proximal_legacy: fixed beta, <= `iters`
// iterations, layer-global early stop), hqq/core/bitpack.py (pack).  axis=1, channel_wise=True.
//
// In the reference this is ~20 iterations x ~12 eager kernels, each a full pass over 4*N*K bytes plus a
// host sync per iteration.  Here the whole iteration loop runs in registers:
//
//   solve_kernel   8 lanes own one group of `gs` weights (gs/8 per lane, element c = 8*v + lane), a wave
//                  owns 8 groups.  No MFMA, no LDS traffic in the loop: the row mean is a sub-wave
//                  reduction.  Writes zero after every iteration (zero_hist) and per-lane |W_f - W_r|
//                  partials; the layer-global stop iteration cannot be known before all groups finish.
//   reduce_err     deterministic double-precision column sums of the per-block error partials.
//   finalize_pack  picks the stop iteration T exactly like optimize.py:244-247 (zero is overwritten
//                  BEFORE the test, so the returned zero is the one computed in the breaking iteration),
//                  recomputes W_q = clamp(rint(W*scale + zero)) from the float32 weights (optimize.py:254)
//                  and packs straight into the reference layout; emits meta scale = 1/scale and zero.
//
// Arithmetic is float32 with the reference's exact op sequence (this file is compiled with
// -ffp-contract=off): mul then add, true division, rint half-to-even.  The row mean reproduces ATen's
// float summation order for a contiguous inner reduction (SumKernel.cpp: 8-wide vectors, 4 interleaved
// accumulators, sequential lane sum), which is why the lane<->element map is c = 8*v + lane.
// |e|^(p-1) is evaluated in double and rounded once (ATen uses Sleef's <=1ulp powf).  The global error is
// summed in double.  See oracle/hqq_oracle.c, which restates the same sequence on the CPU.
#include <algorithm>

#include "hqq_common.h"

namespace hqq {

constexpr int SOLVE_THREADS = 256;          // 4 waves, 32 groups per workgroup
constexpr int SOLVE_GROUPS_PER_BLOCK = SOLVE_THREADS / 8;
constexpr int MAX_ITERS = 64;

template <typename T> __device__ __forceinline__ float load_f32(const T* p, int64_t i);
template <> __device__ __forceinline__ float load_f32<float>(const float* p, int64_t i) { return p[i]; }
template <> __device__ __forceinline__ float load_f32<half_t>(const half_t* p, int64_t i) { return static_cast<float>(p[i]); }
template <> __device__ __forceinline__ float load_f32<bf16_t>(const bf16_t* p, int64_t i) { return bf16_to_f32(p[i].v); }

__device__ __forceinline__ float sgnf(float e) { return static_cast<float>((e > 0.f) - (e < 0.f)); }

// |e|^(p-1) evaluated in double and rounded once to float (ATen's Sleef powf is <= 1 ulp; this matched it on every fixture)
__device__ __forceinline__ float pow_lp(float a, double pexp) {
  return static_cast<float>(pow(static_cast<double>(a), pexp));   // a = 0 -> +inf
}


struct SolveParams {
  int64_t R;          // number of groups
  int gs;
  float maxv;
  int round_zero;
  int iters;          // 0 when optimize == false
  float inv_beta;     // (float)(1.0 / beta)
  double pexp;        // (double)(float)(lp_norm - 1)
  int lp_is_one;
  float a_skip;       // |e| below this: the shrinkage is provably clamped to 0, no pow needed (0: always evaluate)
  const float* scale_in;   // hqq_hip_optimize: start from the caller's scale / zero instead of the group's min / max (else nullptr)
  const float* zero_in;
};

// shrink_lp_op (optimize.py:96-108): W_e = sign(e) max(|e| - (1/beta) |e|^(p-1), 0).  The double-precision pow was 84 % of the
// solver's time, and for 0 < p < 1 it is only needed near and above a* = (1/beta)^(1/(2-p)), where the argument changes sign: with
// a = c a*, the argument is a* (c - c^(p-1)) — increasing in c, and <= -0.1 a* for c <= 0.9, far outside any rounding.  Below
// a_skip = 0.9 a* the caller's clamp gives exactly 0 either way, so the wave skips the pow unless one of its lanes needs it (errors of
// a quantised layer are ~1e-3 against a* = 0.17 at the reference's beta = 10, p = 0.7: practically every wave skips).  Bit-identical.
// returns u = (W_f - W_e) * scale of optimize.py:204-205 with W_e = shrink(e), e = W_f - W_r, a = |e|
__device__ __forceinline__ float shrink_u(float wf, float e, float a, float sc, const SolveParams& p) {
  float t;
  if (p.lp_is_one) {
    t = a - p.inv_beta;
  } else {
    const bool need = !(a < p.a_skip);                          // (NaN: evaluate, it must stay NaN)
    // nobody in the wave can get a non-zero W_e: it is +-0, and (W_f - (+-0)) * scale = W_f * scale (the sign of a zero is dropped by
    // the subtraction from the level that follows) — no pow, no clamp, no sign
    if (__builtin_amdgcn_ballot_w64(need) == 0) return wf * sc;
    const float pw = pow_lp(a, p.pexp);                         // a = 0 -> +inf
    t = p.inv_beta * pw;
    t = a - t;                                                  // 0 - inf = -inf -> clamped below
  }
  t = (t < 0.f) ? 0.f : t;                                      // clamp_min_(0); NaN stays NaN
  const float we = t * sgnf(e);
  const float u = wf - we;                                      // :205
  return u * sc;
}

// workspace: s_ws[R] | zero_hist[(iters+1)][R] | err_part[nblocks][iters] (double) | err_mean[iters] (double)
template <typename WT, int EPL>
__global__ __launch_bounds__(SOLVE_THREADS) void solve_kernel(const WT* __restrict__ W, SolveParams p,
                                                             float* __restrict__ s_ws, float* __restrict__ zero_hist,
                                                             double* __restrict__ err_part) {
  extern __shared__ __attribute__((aligned(16))) float err_lds[];   // [iters][SOLVE_THREADS]
  const int tid = threadIdx.x, lane = tid & 63, j = tid & 7;
  const int64_t r = static_cast<int64_t>(blockIdx.x) * SOLVE_GROUPS_PER_BLOCK + (tid >> 3);
  const bool live = r < p.R;
  const int cl = lane & ~7;   // first lane of this group's 8-lane cluster

  float w[EPL];
#pragma unroll
  for (int v = 0; v < EPL; ++v) w[v] = live ? load_f32<WT>(W, r * p.gs + 8 * v + j) : 0.f;

  // ---- min/max init (quantize.py:118-134) ----
  float mn = w[0], mx = w[0];
#pragma unroll
  for (int v = 1; v < EPL; ++v) { mn = fminf(mn, w[v]); mx = fmaxf(mx, w[v]); }
#pragma unroll
  for (int off = 1; off < 8; off <<= 1) { mn = fminf(mn, __shfl_xor(mn, off, 64)); mx = fmaxf(mx, __shfl_xor(mx, off, 64)); }
  const float denom = mx - mn;
  float sc = (1.0f / denom) * p.maxv;          // Tensor.__rtruediv__: reciprocal() * max_v, two roundings
  if (fabsf(denom) <= 1e-4f) sc = 1.0f;
  sc = (sc > 2e4f) ? 2e4f : sc;   // clamp(max=2e4) keeps a NaN, as torch.clamp does
  float ze = (-mn) * sc;
  if (p.round_zero) ze = rintf(ze);
  if (p.scale_in) { sc = p.scale_in[live ? r : 0]; ze = p.zero_in[live ? r : 0]; }   // optimize_weights_proximal_legacy called on its own
  if (live && j == 0) { s_ws[r] = sc; zero_hist[r] = ze; }

  // ---- proximal iterations (optimize.py:237-247), all `iters` of them; the stop index is chosen later ----
  for (int it = 0; it < p.iters; ++it) {
    float t3[EPL];
    float eabs = 0.f;
#pragma unroll
    for (int v = 0; v < EPL; ++v) {
      const float wf = w[v];
      float q = wf * sc;                       // optimize.py:202
      q = q + ze;
      q = rintf(q);
      q = fminf(fmaxf(q, 0.f), p.maxv);
      const float wr = (q - ze) / sc;          // :203
      const float e = wf - wr;                 // :204
      const float a = fabsf(e);
      eabs += a;                               // :239 (partial of the layer-global mean)
      const float u = shrink_u(wf, e, a, sc, p);   // shrink_lp_op, optimize.py:96-108, :205
      t3[v] = q - u;
    }
    // row sum in ATen order: 4 interleaved accumulators over the 8-wide vectors, leftovers to acc 0,
    // then acc0 += acc1, acc2, acc3; finally the 8 lanes in order starting from 0.
    constexpr int ILP = EPL / 4;
    float a0, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if constexpr (ILP >= 1) {
      a0 = t3[0]; a1 = t3[1]; a2 = t3[2]; a3 = t3[3];
#pragma unroll
      for (int i = 1; i < ILP; ++i) { a0 += t3[4 * i]; a1 += t3[4 * i + 1]; a2 += t3[4 * i + 2]; a3 += t3[4 * i + 3]; }
    } else {
      a0 = 0.f;
    }
#pragma unroll
    for (int v = ILP * 4; v < EPL; ++v) a0 += t3[v];
    a0 += a1; a0 += a2; a0 += a3;
    float fin = 0.f;
#pragma unroll
    for (int l = 0; l < 8; ++l) fin += __shfl(a0, cl | l, 64);
    ze = fin / static_cast<float>(p.gs);       // torch.mean = sum / n
    if (live && j == 0) zero_hist[static_cast<int64_t>(it + 1) * p.R + r] = ze;
    err_lds[it * SOLVE_THREADS + tid] = live ? eabs : 0.f;
  }

  // ---- per-block error partials, fixed summation order ----
  __syncthreads();
  if (tid < p.iters) {
    double sacc = 0.0;
    for (int t = 0; t < SOLVE_THREADS; ++t) sacc += static_cast<double>(err_lds[tid * SOLVE_THREADS + ((t + tid) & (SOLVE_THREADS - 1))]);
    // (rotated start only de-conflicts LDS banks across the `iters` threads; the order is still fixed per thread)
    err_part[static_cast<int64_t>(blockIdx.x) * p.iters + tid] = sacc;
  }
}

// Any other group size (multiples of 8: 512, 1024, a whole row ...): the same 8 lanes per group, the group's weights re-read in
// every iteration instead of held in registers, and ATen's full row-sum order — four interleaved accumulators over the 8-wide
// vectors, each a cascade (16 vectors into level 0, level 0 into level 1, ...: it matters from 512 elements on), leftover vectors
// to accumulator 0, accumulators 1..3 added to 0, the 8 lanes in order (oracle/hqq_oracle.c aten_row_sum_f32).
struct CascadeF {   // one accumulator of multi_row_sum, fed one element at a time (level_step 16)
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int i = 0;
  __device__ __forceinline__ void add(float x) {
    a0 += x;
    ++i;
    if ((i & 15) == 0) {
      a1 += a0; a0 = 0.f;
      if ((i & 0xF0) == 0) {
        a2 += a1; a1 = 0.f;
        if ((i & 0xF00) == 0) { a3 += a2; a2 = 0.f; }
      }
    }
  }
  __device__ __forceinline__ float total() const { float t = a0; t += a1; t += a2; t += a3; return t; }
};

template <typename WT>
__global__ __launch_bounds__(SOLVE_THREADS) void solve_generic_kernel(const WT* __restrict__ W, SolveParams p,
                                                                     float* __restrict__ s_ws, float* __restrict__ zero_hist,
                                                                     double* __restrict__ err_part) {
  extern __shared__ __attribute__((aligned(16))) float err_lds[];   // [iters][SOLVE_THREADS]
  const int tid = threadIdx.x, lane = tid & 63, j = tid & 7;
  const int64_t r = static_cast<int64_t>(blockIdx.x) * SOLVE_GROUPS_PER_BLOCK + (tid >> 3);
  const bool live = r < p.R;
  const int cl = lane & ~7;
  const int gs = p.gs, nvec = gs / 8;        // gs % 8 == 0
  const int size_ilp = nvec / 4;             // rows of the (-1, 4)-shaped vector view
  const WT* wg = W + (live ? r : 0) * gs;
  float mn = load_f32<WT>(wg, j), mx = mn;
  for (int v = 1; v < nvec; ++v) { const float w = load_f32<WT>(wg, 8 * v + j); mn = fminf(mn, w); mx = fmaxf(mx, w); }
#pragma unroll
  for (int off = 1; off < 8; off <<= 1) { mn = fminf(mn, __shfl_xor(mn, off, 64)); mx = fmaxf(mx, __shfl_xor(mx, off, 64)); }
  const float denom = mx - mn;
  float sc = (1.0f / denom) * p.maxv;
  if (fabsf(denom) <= 1e-4f) sc = 1.0f;
  sc = (sc > 2e4f) ? 2e4f : sc;   // clamp(max=2e4) keeps a NaN, as torch.clamp does
  float ze = (-mn) * sc;
  if (p.round_zero) ze = rintf(ze);
  if (p.scale_in) { sc = p.scale_in[live ? r : 0]; ze = p.zero_in[live ? r : 0]; }   // optimize_weights_proximal_legacy called on its own
  if (live && j == 0) { s_ws[r] = sc; zero_hist[r] = ze; }
  for (int it = 0; it < p.iters; ++it) {
    double eabs = 0.0;   // (up to 2^16 elements per lane: keep the per-lane partial of the layer-global error exact enough)
    CascadeF c0, c1, c2, c3;
    float left = 0.f;   // leftover vectors (beyond 4 * size_ilp) are added to accumulator 0 AFTER its cascade is totalled
    float a0 = 0.f;
    bool totalled = false;
    for (int v = 0; v < nvec; ++v) {
      const float wf = load_f32<WT>(wg, 8 * v + j);
      float q = wf * sc;
      q = q + ze;
      q = rintf(q);
      q = fminf(fmaxf(q, 0.f), p.maxv);
      const float wr = (q - ze) / sc;
      const float e = wf - wr;
      const float a = fabsf(e);
      eabs += static_cast<double>(a);
      const float u = shrink_u(wf, e, a, sc, p);
      const float t3 = q - u;
      if (v < size_ilp * 4) {
        const int k = v & 3;
        if (k == 0) c0.add(t3); else if (k == 1) c1.add(t3); else if (k == 2) c2.add(t3); else c3.add(t3);
      } else {
        if (!totalled) { a0 = c0.total(); totalled = true; }
        a0 += t3;
      }
    }
    if (!totalled) a0 = c0.total();
    (void)left;
    a0 += c1.total(); a0 += c2.total(); a0 += c3.total();
    float fin = 0.f;
#pragma unroll
    for (int l = 0; l < 8; ++l) fin += __shfl(a0, cl | l, 64);
    ze = fin / static_cast<float>(gs);
    if (live && j == 0) zero_hist[static_cast<int64_t>(it + 1) * p.R + r] = ze;
    err_lds[it * SOLVE_THREADS + tid] = live ? static_cast<float>(eabs) : 0.f;
  }
  __syncthreads();
  if (tid < p.iters) {
    double sacc = 0.0;
    for (int t = 0; t < SOLVE_THREADS; ++t) sacc += static_cast<double>(err_lds[tid * SOLVE_THREADS + ((t + tid) & (SOLVE_THREADS - 1))]);
    err_part[static_cast<int64_t>(blockIdx.x) * p.iters + tid] = sacc;
  }
}

// err_mean[it] = sum_b err_part[b][it] / numel, deterministic tree; one workgroup per iteration
__global__ __launch_bounds__(256) void reduce_err_kernel(const double* __restrict__ err_part, double* __restrict__ err_mean,
                                                         int64_t nblocks, int iters, double inv_numel) {
  __shared__ double red[256];
  const int it = blockIdx.x, tid = threadIdx.x;
  double s = 0.0;
  for (int64_t b = tid; b < nblocks; b += 256) s += err_part[b * iters + it];
  red[tid] = s;
  __syncthreads();
  for (int w = 128; w >= 1; w >>= 1) {
    if (tid < w) red[tid] += red[tid + w];
    __syncthreads();
  }
  if (tid == 0) err_mean[it] = red[0] * inv_numel;
}

// stop rule of optimize.py:237-247 on float32 errors; returns the zero_hist slot to use
__device__ __forceinline__ int pick_slot(const double* err_mean, int iters, int* ran) {
  if (iters == 0) { *ran = 0; return 0; }
  float best = __builtin_inff();
  int i = 0;
  for (; i < iters; ++i) {
    const float cur = static_cast<float>(err_mean[i]);
    if (cur < best) best = cur; else break;
  }
  const int T = (i < iters) ? i : iters - 1;   // breaking iteration, or the last one
  *ran = T + 1;
  return T + 1;                                // zero after iteration T
}

// W_q = clamp(rint(W*scale + zero), 0, maxv) packed into the reference layout; VEC containers per thread
template <typename WT, int NBITS, int VEC>
__global__ __launch_bounds__(256) void finalize_pack_kernel(const WT* __restrict__ W, const float* __restrict__ s_ws,
                                                            const float* __restrict__ zero_hist, const double* __restrict__ err_mean,
                                                            void* __restrict__ Wq_out, float* __restrict__ scale_out,
                                                            float* __restrict__ zero_out, int32_t* __restrict__ info_out,
                                                            int64_t n, int64_t total, int64_t R, int gs, float maxv, int iters) {
  constexpr int PER = (NBITS == 3) ? 10 : 8 / NBITS;
  constexpr int SUB = (VEC >= 8) ? 8 : VEC;    // gs is a multiple of 8: a SUB-chunk never straddles a group
  int ran;
  const int slot = pick_slot(err_mean, iters, &ran);
  const float* zsel = zero_hist + static_cast<int64_t>(slot) * R;
  const int64_t i0 = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) * VEC;
  if (i0 == 0 && info_out) { info_out[0] = ran; info_out[1] = slot - 1; }
  if (i0 >= n) return;
  uint32_t acc[VEC];
#pragma unroll
  for (int c = 0; c < VEC; ++c) acc[c] = 0;
#pragma unroll
  for (int s = 0; s < PER; ++s) {
    const int64_t e0 = static_cast<int64_t>(s) * n + i0;
    if (e0 >= total) continue;   // 3-bit zero padding rows
#pragma unroll
    for (int c = 0; c < VEC; c += SUB) {
      const int64_t r = (e0 + c) / gs;
      const float sc = s_ws[r], ze = zsel[r];
      if ((e0 + c) % gs == 0) { scale_out[r] = 1.0f / sc; zero_out[r] = ze; }   // quantize.py:154
#pragma unroll
      for (int k = 0; k < SUB; ++k) {
        float q = load_f32<WT>(W, e0 + c + k) * sc;
        q = q + ze;
        q = rintf(q);
        q = fminf(fmaxf(q, 0.f), maxv);
        const int sh = (NBITS == 3) ? (27 - 3 * s) : NBITS * (PER - 1 - s);
        acc[c + k] |= static_cast<uint32_t>(q) << sh;
      }
    }
  }
  if constexpr (NBITS == 3) {
#pragma unroll
    for (int c = 0; c < VEC; ++c) static_cast<int32_t*>(Wq_out)[i0 + c] = static_cast<int32_t>(acc[c]);
  } else {
    uint8_t b[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) b[c] = static_cast<uint8_t>(acc[c]);
    if constexpr (VEC == 16) *reinterpret_cast<u32x4*>(static_cast<uint8_t*>(Wq_out) + i0) = *reinterpret_cast<u32x4*>(b);
    else if constexpr (VEC == 8) *reinterpret_cast<u32x2*>(static_cast<uint8_t*>(Wq_out) + i0) = *reinterpret_cast<u32x2*>(b);
    else static_cast<uint8_t*>(Wq_out)[i0] = b[0];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// axis = 0 (quantize.py:104-116 with axis=0): W viewed as [gs, C], C = numel / gs; group j is COLUMN j, statistics and the
// solver's mean run down the rows.  One thread per group, the group's weights re-read (coalesced across the threads of a
// wave) in every iteration; the mean restates ATen's float sum over the OUTER dimension (SumKernel.cpp vectorized_outer_sum):
// columns below 32 * floor(C / 32) a cascade over the rows (16 rows into level 0, level 0 into level 1, ...), the remaining
// columns the row_sum order (four interleaved cascades over rows i % 4, leftover rows, partials added in order) — see
// oracle/hqq_oracle.c aten_col_sum_f32, which is pinned to the reference.
// ---------------------------------------------------------------------------------------------------------------------
struct Cascade {   // multi_row_sum's accumulation for one output, fed one row at a time (level_step 16: row counts below 2^16)
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int i = 0;
  __device__ __forceinline__ void add(float x) {
    a0 += x;
    ++i;
    if ((i & 15) == 0) {
      a1 += a0; a0 = 0.f;
      if ((i & 0xF0) == 0) {
        a2 += a1; a1 = 0.f;
        if ((i & 0xF00) == 0) { a3 += a2; a2 = 0.f; }
      }
    }
  }
  __device__ __forceinline__ float total() const { float t = a0; t += a1; t += a2; t += a3; return t; }
};

template <typename WT>
__global__ __launch_bounds__(256) void solve0_kernel(const WT* __restrict__ W, SolveParams p, int64_t C,
                                                     float* __restrict__ s_ws, float* __restrict__ zero_hist, double* __restrict__ err_part) {
  extern __shared__ __attribute__((aligned(16))) double err_lds0[];   // [iters][256]
  const int tid = threadIdx.x;
  const int64_t j = static_cast<int64_t>(blockIdx.x) * 256 + tid;
  const bool live = j < C;
  const int gs = p.gs;
  const bool casc_col = j < (C / 32) * 32;
  float mn = 0.f, mx = 0.f;
  if (live) {
    mn = mx = load_f32<WT>(W, j);
    for (int i = 1; i < gs; ++i) { const float w = load_f32<WT>(W, static_cast<int64_t>(i) * C + j); mn = fminf(mn, w); mx = fmaxf(mx, w); }
  }
  const float denom = mx - mn;
  float sc = (1.0f / denom) * p.maxv;
  if (fabsf(denom) <= 1e-4f) sc = 1.0f;
  sc = (sc > 2e4f) ? 2e4f : sc;   // clamp(max=2e4) keeps a NaN, as torch.clamp does
  float ze = (-mn) * sc;
  if (p.round_zero) ze = rintf(ze);
  if (p.scale_in) { sc = p.scale_in[live ? j : 0]; ze = p.zero_in[live ? j : 0]; }
  if (live) { s_ws[j] = sc; zero_hist[j] = ze; }
  const int size4 = (gs / 4) * 4;
  for (int it = 0; it < p.iters; ++it) {
    double eabs = 0.0;
    Cascade c0, c1, c2, c3;
    float tail = 0.f;   // row_sum: rows past 4 * floor(gs / 4), added to partial 0 after its cascade
    if (live) {
      for (int i = 0; i < gs; ++i) {
        const float wf = load_f32<WT>(W, static_cast<int64_t>(i) * C + j);
        float q = wf * sc;
        q = q + ze;
        q = rintf(q);
        q = fminf(fmaxf(q, 0.f), p.maxv);
        const float wr = (q - ze) / sc;
        const float e = wf - wr;
        const float aa = fabsf(e);
        eabs += static_cast<double>(aa);
        const float u = shrink_u(wf, e, aa, sc, p);
        const float t3 = q - u;
        if (casc_col) {
          c0.add(t3);
        } else if (i < size4) {
          const int k = i & 3;
          if (k == 0) c0.add(t3); else if (k == 1) c1.add(t3); else if (k == 2) c2.add(t3); else c3.add(t3);
        } else {
          tail += t3;   // (at most three rows, added to partial 0 in order below: 0 + x is exact, so summing them first is the same)
        }
      }
      float sum;
      if (casc_col) {
        sum = c0.total();
      } else {
        float p0 = c0.total();
        // leftover rows go to partial 0 one by one (row_sum); `tail` holds them already added in order only when there is one —
        // keep the reference's association: re-read is avoided by never having more than 3 and adding them individually
        sum = p0;
        for (int i = size4; i < gs; ++i) {
          const float wf = load_f32<WT>(W, static_cast<int64_t>(i) * C + j);
          float q = wf * sc; q = q + ze; q = rintf(q); q = fminf(fmaxf(q, 0.f), p.maxv);
          const float wr = (q - ze) / sc;
          const float e = wf - wr;
          const float aa = fabsf(e);
          const float u = shrink_u(wf, e, aa, sc, p);
          sum += q - u;
        }
        sum += c1.total();
        sum += c2.total();
        sum += c3.total();
        (void)tail;
      }
      ze = sum / static_cast<float>(gs);
      zero_hist[static_cast<int64_t>(it + 1) * C + j] = ze;
    }
    err_lds0[it * 256 + tid] = live ? eabs : 0.0;
  }
  __syncthreads();
  if (tid < p.iters) {
    double sacc = 0.0;
    for (int t = 0; t < 256; ++t) sacc += err_lds0[tid * 256 + ((t + tid) & 255)];
    err_part[static_cast<int64_t>(blockIdx.x) * p.iters + tid] = sacc;
  }
}

// W_q[i, j] = clamp(rint(W*scale_j + zero_j)) packed row slab by row slab: packed row pr holds rows s * step + pr of the [gs, C] view
template <typename WT, int NBITS>
__global__ __launch_bounds__(256) void finalize_pack0_kernel(const WT* __restrict__ W, const float* __restrict__ s_ws, const float* __restrict__ zero_hist,
                                                             const double* __restrict__ err_mean, void* __restrict__ Wq_out, float* __restrict__ scale_out,
                                                             float* __restrict__ zero_out, int32_t* __restrict__ info_out, int64_t C, int gs, int step,
                                                             float maxv, int iters) {
  constexpr int PER = (NBITS == 3) ? 10 : 8 / NBITS;
  int ran;
  const int slot = pick_slot(err_mean, iters, &ran);
  const float* zsel = zero_hist + static_cast<int64_t>(slot) * C;
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;   // packed element (pr, j)
  if (idx == 0 && info_out) { info_out[0] = ran; info_out[1] = slot - 1; }
  if (idx >= static_cast<int64_t>(step) * C) return;
  const int pr = static_cast<int>(idx / C);
  const int64_t j = idx - static_cast<int64_t>(pr) * C;
  const float sc = s_ws[j], ze = zsel[j];
  if (pr == 0) { scale_out[j] = 1.0f / sc; zero_out[j] = ze; }
  uint32_t acc = 0;
#pragma unroll
  for (int s = 0; s < PER; ++s) {
    const int i = s * step + pr;
    if (i >= gs) continue;   // 3-bit zero padding rows
    float q = load_f32<WT>(W, static_cast<int64_t>(i) * C + j) * sc;
    q = q + ze;
    q = rintf(q);
    q = fminf(fmaxf(q, 0.f), maxv);
    const int sh = (NBITS == 3) ? (27 - 3 * s) : NBITS * (PER - 1 - s);
    acc |= static_cast<uint32_t>(q) << sh;
  }
  if constexpr (NBITS == 3) static_cast<int32_t*>(Wq_out)[idx] = static_cast<int32_t>(acc);
  else static_cast<uint8_t*>(Wq_out)[idx] = static_cast<uint8_t>(acc);
}

static inline size_t align256(size_t v) { return (v + 255) & ~static_cast<size_t>(255); }

struct WsLayout {
  size_t s_off, zh_off, ep_off, em_off, total;
  int64_t nblocks;
};
static WsLayout ws_layout(int64_t numel, int64_t gs, int iters) {
  WsLayout L;
  const int64_t R = numel / gs;
  L.nblocks = (R + SOLVE_GROUPS_PER_BLOCK - 1) / SOLVE_GROUPS_PER_BLOCK;
  L.s_off = 0;
  L.zh_off = align256(sizeof(float) * static_cast<size_t>(R));
  L.ep_off = L.zh_off + align256(sizeof(float) * static_cast<size_t>(R) * static_cast<size_t>(iters + 1));
  L.em_off = L.ep_off + align256(sizeof(double) * static_cast<size_t>(L.nblocks) * static_cast<size_t>(iters > 0 ? iters : 1));
  L.total = L.em_off + align256(sizeof(double) * static_cast<size_t>(iters > 0 ? iters : 1));
  return L;
}

template <typename WT, int EPL>
static void launch_solve(const void* W, const SolveParams& p, float* s_ws, float* zh, double* ep, int64_t nblocks, hipStream_t st) {
  hipLaunchKernelGGL((solve_kernel<WT, EPL>), dim3(static_cast<unsigned>(nblocks)), dim3(SOLVE_THREADS), sizeof(float) * SOLVE_THREADS * (p.iters > 0 ? p.iters : 1), st,
                     static_cast<const WT*>(W), p, s_ws, zh, ep);
}

template <typename WT>
static int dispatch_solve(const void* W, const SolveParams& p, float* s_ws, float* zh, double* ep, int64_t nblocks, hipStream_t st) {
  switch (p.gs) {
    case 8: launch_solve<WT, 1>(W, p, s_ws, zh, ep, nblocks, st); break;
    case 16: launch_solve<WT, 2>(W, p, s_ws, zh, ep, nblocks, st); break;
    case 32: launch_solve<WT, 4>(W, p, s_ws, zh, ep, nblocks, st); break;
    case 64: launch_solve<WT, 8>(W, p, s_ws, zh, ep, nblocks, st); break;
    case 128: launch_solve<WT, 16>(W, p, s_ws, zh, ep, nblocks, st); break;
    case 256: launch_solve<WT, 32>(W, p, s_ws, zh, ep, nblocks, st); break;
    default:
      if (p.gs % 8 || p.gs >= (1 << 19)) {   // (the reference's configuration asserts multiples of 8, quantize.py:1088-1091)
        set_error("hqq_hip_quantize: group_size=%d not covered (multiples of 8 below 2^19)", p.gs);
        return HQQ_ERR_UNSUPPORTED;
      }
      hipLaunchKernelGGL((solve_generic_kernel<WT>), dim3(static_cast<unsigned>(nblocks)), dim3(SOLVE_THREADS), sizeof(float) * SOLVE_THREADS * (p.iters > 0 ? p.iters : 1), st,
                         static_cast<const WT*>(W), p, s_ws, zh, ep);
      break;
  }
  return check_launch("hqq_hip_quantize(solve)");
}

template <typename WT, int NBITS>
static int launch_finalize(const void* W, const float* s_ws, const float* zh, const double* em, void* Wq, float* so, float* zo,
                           int32_t* info, int64_t n, int64_t total, int64_t R, int gs, float maxv, int iters, hipStream_t st) {
  constexpr int V = (NBITS == 3) ? 4 : 16;
  const dim3 blk(256);
  if (n % V == 0) {
    const dim3 grid(static_cast<unsigned>((n / V + 255) / 256));
    hipLaunchKernelGGL((finalize_pack_kernel<WT, NBITS, V>), grid, blk, 0, st, static_cast<const WT*>(W), s_ws, zh, em, Wq, so, zo, info, n, total, R, gs, maxv, iters);
  } else if (NBITS != 3 && n % 8 == 0) {
    const dim3 grid(static_cast<unsigned>((n / 8 + 255) / 256));
    hipLaunchKernelGGL((finalize_pack_kernel<WT, NBITS, 8>), grid, blk, 0, st, static_cast<const WT*>(W), s_ws, zh, em, Wq, so, zo, info, n, total, R, gs, maxv, iters);
  } else {
    set_error("hqq_hip_quantize: packed element count %lld not a multiple of %d", (long long)n, (NBITS == 3) ? 4 : 8);
    return HQQ_ERR_UNSUPPORTED;
  }
  return check_launch("hqq_hip_quantize(finalize)");
}

template <typename WT>
static int dispatch_finalize(int pack_bits, const void* W, const float* s_ws, const float* zh, const double* em, void* Wq, float* so,
                             float* zo, int32_t* info, int64_t n, int64_t total, int64_t R, int gs, float maxv, int iters, hipStream_t st) {
  switch (pack_bits) {
    case 8: return launch_finalize<WT, 8>(W, s_ws, zh, em, Wq, so, zo, info, n, total, R, gs, maxv, iters, st);
    case 4: return launch_finalize<WT, 4>(W, s_ws, zh, em, Wq, so, zo, info, n, total, R, gs, maxv, iters, st);
    case 3: return launch_finalize<WT, 3>(W, s_ws, zh, em, Wq, so, zo, info, n, total, R, gs, maxv, iters, st);
    case 2: return launch_finalize<WT, 2>(W, s_ws, zh, em, Wq, so, zo, info, n, total, R, gs, maxv, iters, st);
    case 1: return launch_finalize<WT, 1>(W, s_ws, zh, em, Wq, so, zo, info, n, total, R, gs, maxv, iters, st);
  }
  return HQQ_ERR_NBITS;
}

template <typename WT>
static int run_quantize(const void* W, int64_t numel, int64_t gs, int max_v, int pack_bits, int round_zero, int iters,
                        float beta, float lp_norm, void* Wq_out, float* scale_out, float* zero_out, int32_t* info_out,
                        void* ws, hipStream_t st, const float* scale_in = nullptr, const float* zero_in = nullptr) {
  const WsLayout L = ws_layout(numel, gs, iters);
  char* base = static_cast<char*>(ws);
  float* s_ws = reinterpret_cast<float*>(base + L.s_off);
  float* zh = reinterpret_cast<float*>(base + L.zh_off);
  double* ep = reinterpret_cast<double*>(base + L.ep_off);
  double* em = reinterpret_cast<double*>(base + L.em_off);
  const int64_t R = numel / gs;
  SolveParams p;
  p.R = R; p.gs = static_cast<int>(gs); p.maxv = static_cast<float>(max_v); p.round_zero = round_zero; p.iters = iters;
  p.inv_beta = static_cast<float>(1.0 / static_cast<double>(beta));
  p.pexp = static_cast<double>(static_cast<float>(static_cast<double>(lp_norm) - 1.0));
  p.lp_is_one = (lp_norm == 1.0f);
  p.scale_in = scale_in; p.zero_in = zero_in;
  p.a_skip = (p.pexp < 0.0 && p.pexp > -1.0 && p.inv_beta > 0.f && true) ? static_cast<float>(0.9 * pow(static_cast<double>(p.inv_beta), 1.0 / (1.0 - p.pexp))) : 0.f;
  int rc = dispatch_solve<WT>(W, p, s_ws, zh, ep, L.nblocks, st);
  if (rc) return rc;
  if (iters > 0) {
    hipLaunchKernelGGL(reduce_err_kernel, dim3(iters), dim3(256), 0, st, ep, em, L.nblocks, iters, 1.0 / static_cast<double>(numel));
    rc = check_launch("hqq_hip_quantize(reduce_err)");
    if (rc) return rc;
  }
  const int64_t prow = hqq_hip_packed_rows(pack_bits, R);
  const int64_t n = prow * gs;
  return dispatch_finalize<WT>(pack_bits, W, s_ws, zh, em, Wq_out, scale_out, zero_out, info_out, n, numel, R,
                               static_cast<int>(gs), static_cast<float>(max_v), iters, st);
}

template <typename WT>
static int run_quantize_axis0(const void* W, int64_t numel, int64_t gs, int max_v, int pack_bits, int round_zero, int iters,
                              float beta, float lp_norm, void* Wq_out, float* scale_out, float* zero_out, int32_t* info_out,
                              void* ws, hipStream_t st, const float* scale_in = nullptr, const float* zero_in = nullptr) {
  const WsLayout L = ws_layout(numel, gs, iters);
  char* base = static_cast<char*>(ws);
  float* s_ws = reinterpret_cast<float*>(base + L.s_off);
  float* zh = reinterpret_cast<float*>(base + L.zh_off);
  double* ep = reinterpret_cast<double*>(base + L.ep_off);
  double* em = reinterpret_cast<double*>(base + L.em_off);
  const int64_t C = numel / gs;
  SolveParams p;
  p.R = C; p.gs = static_cast<int>(gs); p.maxv = static_cast<float>(max_v); p.round_zero = round_zero; p.iters = iters;
  p.inv_beta = static_cast<float>(1.0 / static_cast<double>(beta));
  p.pexp = static_cast<double>(static_cast<float>(static_cast<double>(lp_norm) - 1.0));
  p.lp_is_one = (lp_norm == 1.0f);
  p.scale_in = scale_in; p.zero_in = zero_in;
  p.a_skip = (p.pexp < 0.0 && p.pexp > -1.0 && p.inv_beta > 0.f && true) ? static_cast<float>(0.9 * pow(static_cast<double>(p.inv_beta), 1.0 / (1.0 - p.pexp))) : 0.f;
  const int64_t nblocks = (C + 255) / 256;   // (<= the axis-1 block count the workspace was sized for)
  hipLaunchKernelGGL((solve0_kernel<WT>), dim3(static_cast<unsigned>(nblocks)), dim3(256), sizeof(double) * 256 * (iters > 0 ? iters : 1), st,
                     static_cast<const WT*>(W), p, C, s_ws, zh, ep);
  int rc = check_launch("hqq_hip_quantize(axis 0 solve)");
  if (rc) return rc;
  if (iters > 0) {
    hipLaunchKernelGGL(reduce_err_kernel, dim3(iters), dim3(256), 0, st, ep, em, nblocks, iters, 1.0 / static_cast<double>(numel));
    rc = check_launch("hqq_hip_quantize(reduce_err)");
    if (rc) return rc;
  }
  const int per = per_of(pack_bits);
  const int step = static_cast<int>((gs + per - 1) / per);
  const int64_t n = static_cast<int64_t>(step) * C;
  const dim3 grid(static_cast<unsigned>((n + 255) / 256)), blk(256);
#define HQQ_FIN0(NB) hipLaunchKernelGGL((finalize_pack0_kernel<WT, NB>), grid, blk, 0, st, static_cast<const WT*>(W), s_ws, zh, em, Wq_out, scale_out, zero_out, info_out, C, static_cast<int>(gs), step, static_cast<float>(max_v), iters)
  switch (pack_bits) {
    case 8: HQQ_FIN0(8); break;
    case 4: HQQ_FIN0(4); break;
    case 3: HQQ_FIN0(3); break;
    case 2: HQQ_FIN0(2); break;
    case 1: HQQ_FIN0(1); break;
    default: return HQQ_ERR_NBITS;
  }
#undef HQQ_FIN0
  return check_launch("hqq_hip_quantize(axis 0 finalize)");
}

// ---------------------------------------------------------------------------------------------------------------------
// channel_wise = False (quantize.py:114-116): ONE scale / zero for the whole tensor from its min and max, no solver; the levels are
// packed in the tensor's own [rows, cols] shape.  min / max are order-free, so a two-level reduction is exact.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int TW_MAX_BLOCKS = 1024;

template <typename WT>
__global__ __launch_bounds__(256) void tensor_minmax_kernel(const WT* __restrict__ W, int64_t numel, float* __restrict__ part) {
  __shared__ float lmn[4], lmx[4];
  float mn = INFINITY, mx = -INFINITY;
  bool nan = false;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < numel; i += static_cast<int64_t>(gridDim.x) * 256) {
    const float v = load_f32<WT>(W, i);
    nan |= (v != v);
    mn = fminf(mn, v); mx = fmaxf(mx, v);
  }
  if (nan) { mn = NAN; mx = NAN; }             // Tensor.min()/max() propagate NaN
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const float a = __shfl_xor(mn, off, 64), b = __shfl_xor(mx, off, 64);
    mn = (a != a || mn != mn) ? NAN : fminf(mn, a);
    mx = (b != b || mx != mx) ? NAN : fmaxf(mx, b);
  }
  if ((threadIdx.x & 63) == 0) { lmn[threadIdx.x >> 6] = mn; lmx[threadIdx.x >> 6] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) {
      mn = (lmn[w] != lmn[w] || mn != mn) ? NAN : fminf(mn, lmn[w]);
      mx = (lmx[w] != lmx[w] || mx != mx) ? NAN : fmaxf(mx, lmx[w]);
    }
    part[2 * blockIdx.x] = mn; part[2 * blockIdx.x + 1] = mx;
  }
}

__global__ __launch_bounds__(64) void tensor_init_kernel(const float* __restrict__ part, int nparts, float maxv, int round_zero,
                                                         float* __restrict__ s_ws, float* __restrict__ zero_hist) {
  float mn = INFINITY, mx = -INFINITY;
  for (int i = threadIdx.x; i < nparts; i += 64) {
    const float a = part[2 * i], b = part[2 * i + 1];
    mn = (a != a || mn != mn) ? NAN : fminf(mn, a);
    mx = (b != b || mx != mx) ? NAN : fmaxf(mx, b);
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const float a = __shfl_xor(mn, off, 64), b = __shfl_xor(mx, off, 64);
    mn = (a != a || mn != mn) ? NAN : fminf(mn, a);
    mx = (b != b || mx != mx) ? NAN : fmaxf(mx, b);
  }
  if (threadIdx.x == 0) {
    const float denom = mx - mn;               // quantize.py:126-134 on 0-d tensors: the same op sequence as per group
    float sc = (1.0f / denom) * maxv;
    if (fabsf(denom) <= 1e-4f) sc = 1.0f;
    sc = (sc > 2e4f) ? 2e4f : sc;   // clamp(max=2e4) keeps a NaN, as torch.clamp does
    float ze = (-mn) * sc;
    if (round_zero) ze = rintf(ze);
    s_ws[0] = sc; zero_hist[0] = ze;
  }
}

template <typename WT>
static int run_quantize_tensor(const void* W, int64_t rows, int64_t cols, int max_v, int pack_bits, int round_zero,
                               void* Wq_out, float* scale_out, float* zero_out, void* ws, hipStream_t st) {
  const int64_t numel = rows * cols;
  float* s_ws = static_cast<float*>(ws);       // [0] scale, [1] zero, [2] unused, [4..] per-block (min, max)
  float* zh = s_ws + 1;
  float* part = s_ws + 4;
  const int nb = static_cast<int>(std::min<int64_t>(TW_MAX_BLOCKS, (numel + 2047) / 2048));
  hipLaunchKernelGGL((tensor_minmax_kernel<WT>), dim3(nb), dim3(256), 0, st, static_cast<const WT*>(W), numel, part);
  int rc = check_launch("hqq_hip_quantize_tensor(min/max)");
  if (rc) return rc;
  hipLaunchKernelGGL(tensor_init_kernel, dim3(1), dim3(64), 0, st, part, nb, static_cast<float>(max_v), round_zero, s_ws, zh);
  rc = check_launch("hqq_hip_quantize_tensor(init)");
  if (rc) return rc;
  // the packing kernel of the grouped path with the whole tensor as its one group: rows of the [rows, cols] level matrix share a container
  const int64_t n = hqq_hip_packed_rows(pack_bits, rows) * cols;
  return dispatch_finalize<WT>(pack_bits, W, s_ws, zh, nullptr, Wq_out, scale_out, zero_out, nullptr, n, numel, 1,
                               static_cast<int>(numel), static_cast<float>(max_v), 0, st);
}

}  // namespace hqq

using namespace hqq;

extern "C" {

int hqq_hip_quantize_axis0(const void* W, int w_dtype, int64_t numel, int64_t group_size, int max_v, int pack_bits,
                           int round_zero, int optimize, int iters, float beta, float lp_norm,
                           void* Wq_out, float* scale_out, float* zero_out, int32_t* info_out,
                           void* workspace, size_t workspace_bytes, void* stream) {
  clear_stale_error();
  if (numel <= 0 || group_size <= 0 || numel % group_size) {
    set_error("hqq_hip_quantize_axis0: group_size should divide the tensor size (numel=%lld, group_size=%lld)", (long long)numel, (long long)group_size);
    return HQQ_ERR_SHAPE;
  }
  if (!per_of(pack_bits)) { set_error("hqq_hip_quantize_axis0: pack_bits=%d not in {8,4,3,2,1}", pack_bits); return HQQ_ERR_NBITS; }
  if (max_v < 1 || max_v > 255) { set_error("hqq_hip_quantize_axis0: max_v=%d out of range", max_v); return HQQ_ERR_SHAPE; }
  if (group_size >= 65536) { set_error("hqq_hip_quantize_axis0: group_size %lld (the row cascade is restated for < 2^16 rows)", (long long)group_size); return HQQ_ERR_UNSUPPORTED; }
  if (numel / group_size < 8) { set_error("hqq_hip_quantize_axis0: fewer than 8 groups (ATen's scalar outer-sum order is not restated)"); return HQQ_ERR_UNSUPPORTED; }
  if (pack_bits != 3 && group_size % per_of(pack_bits)) {
    set_error("hqq_hip_quantize_axis0: group_size %lld rows cannot be packed at %d bits (must divide by %d)", (long long)group_size, pack_bits, per_of(pack_bits));
    return HQQ_ERR_SHAPE;
  }
  if (!optimize) iters = 0;
  if (iters < 0 || iters > MAX_ITERS) { set_error("hqq_hip_quantize_axis0: iters=%d outside [0,%d]", iters, MAX_ITERS); return HQQ_ERR_SHAPE; }
  const size_t need = ws_layout(numel, group_size, iters).total;
  if (!workspace || workspace_bytes < need) { set_error("hqq_hip_quantize_axis0: workspace %zu < %zu bytes", workspace_bytes, need); return HQQ_ERR_WORKSPACE; }
  if (!aligned16(W) || !aligned16(Wq_out) || !aligned16(workspace)) { set_error("hqq_hip_quantize_axis0: pointers must be 16-byte aligned"); return HQQ_ERR_ALIGN; }
  hipStream_t st = as_stream(stream);
  switch (w_dtype) {
    case HQQ_F32: return run_quantize_axis0<float>(W, numel, group_size, max_v, pack_bits, round_zero, iters, beta, lp_norm, Wq_out, scale_out, zero_out, info_out, workspace, st);
    case HQQ_F16: return run_quantize_axis0<half_t>(W, numel, group_size, max_v, pack_bits, round_zero, iters, beta, lp_norm, Wq_out, scale_out, zero_out, info_out, workspace, st);
    case HQQ_BF16: return run_quantize_axis0<bf16_t>(W, numel, group_size, max_v, pack_bits, round_zero, iters, beta, lp_norm, Wq_out, scale_out, zero_out, info_out, workspace, st);
  }
  set_error("hqq_hip_quantize_axis0: bad w_dtype %d", w_dtype);
  return HQQ_ERR_DTYPE;
}

int hqq_hip_quantize_tensor(const void* W, int w_dtype, int64_t rows, int64_t cols, int max_v, int pack_bits, int round_zero,
                            void* Wq_out, float* scale_out, float* zero_out, void* workspace, size_t workspace_bytes, void* stream) {
  clear_stale_error();
  if (rows <= 0 || cols <= 0 || rows * cols >= (int64_t(1) << 31)) { set_error("hqq_hip_quantize_tensor: shape [%lld, %lld] (below 2^31 elements)", (long long)rows, (long long)cols); return HQQ_ERR_SHAPE; }
  if (!per_of(pack_bits)) { set_error("hqq_hip_quantize_tensor: pack_bits=%d not in {8,4,3,2,1}", pack_bits); return HQQ_ERR_NBITS; }
  if (max_v < 1 || max_v > 255) { set_error("hqq_hip_quantize_tensor: max_v=%d out of range", max_v); return HQQ_ERR_SHAPE; }
  if (hqq_hip_packed_rows(pack_bits, rows) < 0) {
    set_error("hqq_hip_quantize_tensor: %lld rows cannot be packed at %d bits (row count must divide by %d)", (long long)rows, pack_bits, per_of(pack_bits));
    return HQQ_ERR_SHAPE;
  }
  if (cols % 8) { set_error("hqq_hip_quantize_tensor: cols=%lld must be a multiple of 8", (long long)cols); return HQQ_ERR_UNSUPPORTED; }
  if (!workspace || workspace_bytes < HQQ_QUANTIZE_TENSOR_WS_BYTES) { set_error("hqq_hip_quantize_tensor: workspace %zu < %d bytes", workspace_bytes, HQQ_QUANTIZE_TENSOR_WS_BYTES); return HQQ_ERR_WORKSPACE; }
  if (!aligned16(W) || !aligned16(Wq_out) || !aligned16(workspace)) { set_error("hqq_hip_quantize_tensor: pointers must be 16-byte aligned"); return HQQ_ERR_ALIGN; }
  hipStream_t st = as_stream(stream);
  switch (w_dtype) {
    case HQQ_F32: return run_quantize_tensor<float>(W, rows, cols, max_v, pack_bits, round_zero, Wq_out, scale_out, zero_out, workspace, st);
    case HQQ_F16: return run_quantize_tensor<half_t>(W, rows, cols, max_v, pack_bits, round_zero, Wq_out, scale_out, zero_out, workspace, st);
    case HQQ_BF16: return run_quantize_tensor<bf16_t>(W, rows, cols, max_v, pack_bits, round_zero, Wq_out, scale_out, zero_out, workspace, st);
  }
  set_error("hqq_hip_quantize_tensor: bad w_dtype %d", w_dtype);
  return HQQ_ERR_DTYPE;
}

int hqq_hip_optimize(const void* W, int w_dtype, int64_t numel, int64_t group_size, int axis, int max_v, const float* scale_in, const float* zero_in,
                     int iters, float beta, float lp_norm, void* levels_out, float* zero_out, int32_t* info_out,
                     void* workspace, size_t workspace_bytes, void* stream) {
  clear_stale_error();
  if (numel <= 0 || group_size <= 0 || numel % group_size || (axis != 0 && axis != 1)) { set_error("hqq_hip_optimize: bad numel / group_size / axis"); return HQQ_ERR_SHAPE; }
  if (!scale_in || !zero_in || !levels_out || !zero_out) { set_error("hqq_hip_optimize: null argument"); return HQQ_ERR_SHAPE; }
  if (max_v < 1 || max_v > 255) { set_error("hqq_hip_optimize: max_v=%d out of range", max_v); return HQQ_ERR_SHAPE; }
  if (iters < 0 || iters > MAX_ITERS) { set_error("hqq_hip_optimize: iters=%d outside [0,%d]", iters, MAX_ITERS); return HQQ_ERR_SHAPE; }
  if (axis == 0 && (group_size >= 65536 || numel / group_size < 8)) { set_error("hqq_hip_optimize: axis 0 needs group_size < 2^16 and at least 8 groups"); return HQQ_ERR_UNSUPPORTED; }
  const int64_t R = numel / group_size;
  const size_t need = ws_layout(numel, group_size, iters).total + static_cast<size_t>(R) * sizeof(float);   // (+ a scratch row for the inverted scale the packing step writes)
  if (!workspace || workspace_bytes < need) { set_error("hqq_hip_optimize: workspace %zu < %zu bytes (hqq_hip_quantize_workspace_bytes + 4 bytes per group)", workspace_bytes, need); return HQQ_ERR_WORKSPACE; }
  if (!aligned16(W) || !aligned16(levels_out) || !aligned16(workspace)) { set_error("hqq_hip_optimize: pointers must be 16-byte aligned"); return HQQ_ERR_ALIGN; }
  float* inv_scale_scratch = reinterpret_cast<float*>(static_cast<char*>(workspace) + ws_layout(numel, group_size, iters).total);
  hipStream_t st = as_stream(stream);
#define HQQ_OPT_GO(T) (axis == 1 ? run_quantize<T>(W, numel, group_size, max_v, 8, 0, iters, beta, lp_norm, levels_out, inv_scale_scratch, zero_out, info_out, workspace, st, scale_in, zero_in) \
                                 : run_quantize_axis0<T>(W, numel, group_size, max_v, 8, 0, iters, beta, lp_norm, levels_out, inv_scale_scratch, zero_out, info_out, workspace, st, scale_in, zero_in))
  switch (w_dtype) {
    case HQQ_F32: return HQQ_OPT_GO(float);
    case HQQ_F16: return HQQ_OPT_GO(half_t);
    case HQQ_BF16: return HQQ_OPT_GO(bf16_t);
  }
#undef HQQ_OPT_GO
  set_error("hqq_hip_optimize: bad w_dtype %d", w_dtype);
  return HQQ_ERR_DTYPE;
}

size_t hqq_hip_quantize_workspace_bytes(int64_t numel, int64_t group_size, int iters) {
  if (numel <= 0 || group_size <= 0 || numel % group_size || iters < 0 || iters > MAX_ITERS) return 0;
  return ws_layout(numel, group_size, iters).total;
}

int hqq_hip_quantize(const void* W, int w_dtype, int64_t numel, int64_t group_size, int max_v, int pack_bits,
                     int round_zero, int optimize, int iters, float beta, float lp_norm,
                     void* Wq_out, float* scale_out, float* zero_out, int32_t* info_out,
                     void* workspace, size_t workspace_bytes, void* stream) {
  clear_stale_error();
  if (numel <= 0 || group_size <= 0 || numel % group_size) {   // quantize.py:94-100
    set_error("hqq_hip_quantize: group_size should divide the tensor size (numel=%lld, group_size=%lld)", (long long)numel, (long long)group_size);
    return HQQ_ERR_SHAPE;
  }
  if (!per_of(pack_bits)) { set_error("hqq_hip_quantize: pack_bits=%d not in {8,4,3,2,1}", pack_bits); return HQQ_ERR_NBITS; }
  if (max_v < 1 || max_v > 255) { set_error("hqq_hip_quantize: max_v=%d out of range", max_v); return HQQ_ERR_SHAPE; }
  if (!optimize) iters = 0;
  if (iters < 0 || iters > MAX_ITERS) { set_error("hqq_hip_quantize: iters=%d outside [0,%d]", iters, MAX_ITERS); return HQQ_ERR_SHAPE; }
  const int64_t R = numel / group_size;
  if (hqq_hip_packed_rows(pack_bits, R) < 0) {
    set_error("hqq_hip_quantize: %lld groups cannot be packed at %d bits (row count must divide by %d)", (long long)R, pack_bits, per_of(pack_bits));
    return HQQ_ERR_SHAPE;
  }
  const size_t need = ws_layout(numel, group_size, iters).total;
  if (!workspace || workspace_bytes < need) { set_error("hqq_hip_quantize: workspace %zu < %zu bytes", workspace_bytes, need); return HQQ_ERR_WORKSPACE; }
  if (!aligned16(W) || !aligned16(Wq_out) || !aligned16(workspace)) { set_error("hqq_hip_quantize: pointers must be 16-byte aligned"); return HQQ_ERR_ALIGN; }
  hipStream_t st = as_stream(stream);
  switch (w_dtype) {
    case HQQ_F32: return run_quantize<float>(W, numel, group_size, max_v, pack_bits, round_zero, iters, beta, lp_norm, Wq_out, scale_out, zero_out, info_out, workspace, st);
    case HQQ_F16: return run_quantize<half_t>(W, numel, group_size, max_v, pack_bits, round_zero, iters, beta, lp_norm, Wq_out, scale_out, zero_out, info_out, workspace, st);
    case HQQ_BF16: return run_quantize<bf16_t>(W, numel, group_size, max_v, pack_bits, round_zero, iters, beta, lp_norm, Wq_out, scale_out, zero_out, info_out, workspace, st);
  }
  set_error("hqq_hip_quantize: bad w_dtype %d", w_dtype);
  return HQQ_ERR_DTYPE;
}

}  // extern "C"
