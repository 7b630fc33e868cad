// gemm_pipe.hip — fused unpack -> dequantize -> MFMA GEMM for the rows between decode and long prefill (65 <= M <= ~1024: batched
// decode, speculative verification, short prompts), gfx950, fp16, 4-/2-bit.
//
// Reference chain replaced (axis=1): BitPack.unpack_* -> (W_r - zero) * scale -> torch.matmul(x, W.t()) (+ bias)
//   hqq/core/bitpack.py:31-64, hqq/core/quantize.py:183-199, :880-898.
// Why a kernel of its own: in this range the layer is neither weight-streaming (skinny.hip: one pass over the packed bytes, M <= 64)
// nor big enough for a plain output-tile grid — a 4096 x 4096 layer at M = 128 has 32 tiles of 128 x 128 for 256 CUs, and the
// composition "dequantise kernel + library GEMM" pays 13-38 us for writing and re-reading the fp16 matrix around a GEMM of 20-35 us
// (tools/sweep_prefill.py).  Here the packed weights are the only weight bytes that leave HBM, and K is split across workgroups
// until the chip is full:
//   tile      128 output features (PER slabs x 128/PER packed rows) x BM tokens (64 or 128) x 64 k per step, four waves as
//             2 (features) x 2 (tokens); W is the MFMA A operand, so a lane ends up with 4 consecutive features of one token.
//   pipeline  global -> registers three steps ahead (three register sets, used round-robin), dequantise + write to LDS one step
//             ahead of the MFMAs (two LDS stages), ONE workgroup barrier per step; two workgroups per CU overlap each other's
//             barriers and memory latency.
//   rebuild   the three-op exact sequence of decode_common.h where the layer's (zero, scale) allow it (HQQ_OPT_META_SCALABLE),
//             else the four-op one: the same two fp16 roundings as Quantizer.dequantize either way.  The packed dword's middle
//             bytes are swapped first (one v_perm per 4 bytes), so that the masked pairs come out in natural k order and x goes to
//             LDS unpermuted.
//   split-K   grid = tiles x KS; every split parks its fp32 tile in the caller's workspace, and a second small launch adds the KS
//             tiles in split order, rounds, adds the bias and stores: fixed order, reproducible bits.
#include "decode_common.h"

namespace hqq {

constexpr int GP_N = 128, GP_K = 64, GP_T = 256, GP_NS = 3;
constexpr int GP_MAX_KS = 16;

struct GpArgs {
  const half_t* x;
  const uint8_t* Wq;
  const half_t* scale;
  const half_t* zero;
  const half_t* bias;
  half_t* y;
  float* part;     // [KS][tiles][4 MT accumulator quads][256 threads] x 4 fp32 (KS > 1 only)
  int M, N, K, gs_shift, G, n_tiles, m_tiles, KS, kps;
};

__device__ __forceinline__ int gp_off(int row, int c) { return row * 128 + ((c ^ (row & 7)) << 4); }   // 16-byte chunk c of a [rows][64] fp16 tile

template <int NBITS, int S, int PER, int DW, bool SUB>
struct GpSlab {   // the 4 DW k-values of slab S in `w` (middle bytes of every dword already swapped) -> 2 DW fp16 pairs in natural k order
  static __device__ __forceinline__ void run(const uint32_t (&w)[DW], const half_t (&z)[PER], const half_t (&s)[PER], uint32_t (&out)[PER][2 * DW]) {
    constexpr int sh = NBITS * (PER - 1 - S);
    constexpr uint32_t m1 = ((NBITS == 8) ? 0xFFu : ((1u << NBITS) - 1u)) << sh;
    constexpr uint32_t m = m1 | (m1 << 16);
    half2_t q[2 * DW];
    uint32_t o[2 * DW];
    if constexpr (SUB) {
      constexpr int J = 9 - sh;
      const half_t zj = z[S] * static_cast<half_t>(1.0f / static_cast<float>(1 << J));   // exact (hqq_hip_meta_check)
      const half_t sj = s[S] * static_cast<half_t>(static_cast<float>(1 << J));
      const half2_t nz = {-zj, -zj}, ss = {sj, sj};
      const half2_t lift = {static_cast<half_t>(32768.0f), static_cast<half_t>(32768.0f)};
#pragma unroll
      for (int d = 0; d < DW; ++d) {
        q[2 * d] = as_h2(w[d] & m);              // (k 4d, 4d+1): q * 2^(sh-24), a subnormal pair
        q[2 * d + 1] = as_h2((w[d] >> 8) & m);   // (k 4d+2, 4d+3)
      }
#pragma unroll
      for (int i = 0; i < 2 * DW; ++i) q[i] = __builtin_elementwise_fma(q[i], lift, nz);   // rounding 1
#pragma unroll
      for (int i = 0; i < 2 * DW; ++i) o[i] = __builtin_bit_cast(uint32_t, q[i] * ss);      // rounding 2
    } else {
      constexpr float inv = 1.0f / static_cast<float>(1 << sh);
      const half2_t k1 = {static_cast<half_t>(inv), static_cast<half_t>(inv)};
      const half2_t k2 = {static_cast<half_t>(-1024.0f * inv), static_cast<half_t>(-1024.0f * inv)};
      const half2_t zz = {z[S], z[S]}, ss = {s[S], s[S]};
#pragma unroll
      for (int d = 0; d < DW; ++d) {
        q[2 * d] = as_h2((w[d] & m) | 0x64006400u);
        q[2 * d + 1] = as_h2(((w[d] >> 8) & m) | 0x64006400u);
      }
#pragma unroll
      for (int i = 0; i < 2 * DW; ++i) q[i] = __builtin_elementwise_fma(q[i], k1, k2);   // exact integer level
#pragma unroll
      for (int i = 0; i < 2 * DW; ++i) q[i] = q[i] - zz;                                  // rounding 1
#pragma unroll
      for (int i = 0; i < 2 * DW; ++i) o[i] = __builtin_bit_cast(uint32_t, q[i] * ss);    // rounding 2
    }
#pragma unroll
    for (int i = 0; i < 2 * DW; ++i) out[S][i] = o[i];
    if constexpr (S + 1 < PER) GpSlab<NBITS, S + 1, PER, DW, SUB>::run(w, z, s, out);
  }
};

template <int NBITS, int BM, bool SUB>
__global__ __launch_bounds__(GP_T, 2) void gemm_pipe_f16_kernel(const GpArgs a) {
  constexpr int PER = 8 / NBITS;
  constexpr int PROWS = GP_N / PER;            // packed rows per tile
  constexpr int KPT = PROWS / 4;               // packed bytes (= k) per thread and step: every thread carries a piece (16 at 4 bits, 8 at 2, 32 at 8)
  constexpr int DW = KPT / 4, TPRW = GP_K / KPT;   // dwords per thread, threads per packed row
  constexpr int CH = BM / 32;                  // x chunks per thread and step
  constexpr int TPR = 8 / CH;                  // threads per x row
  constexpr int MT = BM / 32;                  // token MFMA tiles per wave (a wave covers BM / 2 tokens)
  constexpr int WBYTES = GP_N * GP_K * 2, STAGE = WBYTES + BM * GP_K * 2;
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];   // [2 stages][W tile | x tile]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 1, wm = wave & 1;
  const int b = blockIdx.x;
  const int nt = b % a.n_tiles, rest = b / a.n_tiles, mt = rest % a.m_tiles, ks = rest / a.m_tiles;
  const int N = a.N, K = a.K, M = a.M, G = a.G;
  const int rows_per_slab = N / PER;
  const int p0 = nt * PROWS, m0 = mt * BM;
  const int nk = K / GP_K;
  const int kt0 = ks * a.kps;
  const int nsteps = (kt0 + a.kps < nk ? kt0 + a.kps : nk) - kt0;

  // The loop body below is branch-free (one basic block per step, so that the scheduler can put the rebuild's VALU work and the LDS
  // traffic between the MFMAs): rows past the end of the slab read the last row with scale 0 — exact zeros —, token rows past M
  // read row 0: their accumulator columns are never stored, and a column depends on its own x row only.
  const int wp = tid / TPRW, wq = tid % TPRW;
  const bool w_active = (p0 + wp) < rows_per_slab;
  const int wrow = w_active ? p0 + wp : rows_per_slab - 1;
  const int xr = tid / TPR, xc0 = (tid % TPR) * CH;
  const uint8_t* wsrc = a.Wq + static_cast<int64_t>(wrow) * K + wq * KPT;
  const half_t* xsrc = a.x + static_cast<int64_t>(m0 + xr < M ? m0 + xr : 0) * K + xc0 * 8;
  const int64_t mrow = static_cast<int64_t>(wrow) * G;
  const half_t smask = w_active ? static_cast<half_t>(1.0f) : static_cast<half_t>(0.0f);

  struct Set { uint32_t w[DW]; half_t z[PER], s[PER]; u32x4 x[CH]; };
  auto load = [&](Set& st, int step) {
    step = step < nsteps ? step : nsteps - 1;   // past the range: the last step again (cached), never staged into a buffer that is read
    const int k0 = (kt0 + step) * GP_K;
    if constexpr (DW == 2) {
      const u32x2 v = *reinterpret_cast<const u32x2*>(wsrc + k0);
      st.w[0] = v.x; st.w[1] = v.y;
    } else {
#pragma unroll
      for (int d4 = 0; d4 < DW / 4; ++d4) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(wsrc + k0 + 16 * d4);
        st.w[4 * d4] = v.x; st.w[4 * d4 + 1] = v.y; st.w[4 * d4 + 2] = v.z; st.w[4 * d4 + 3] = v.w;
      }
    }
    const int g = (k0 + wq * KPT) >> a.gs_shift;
#pragma unroll
    for (int s = 0; s < PER; ++s) {
      const int64_t q = mrow + static_cast<int64_t>(s) * rows_per_slab * G + g;
      st.z[s] = a.zero[q];
      st.s[s] = a.scale[q];
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) st.x[i] = *reinterpret_cast<const u32x4*>(xsrc + k0 + i * 8);
  };
  auto rebuild = [&](const Set& st, uint32_t (&o)[PER][2 * DW]) {
    uint32_t w[DW];
#pragma unroll
    for (int d = 0; d < DW; ++d) w[d] = __builtin_amdgcn_perm(st.w[d], st.w[d], 0x03010200u);   // bytes (b0,b1,b2,b3) -> (b0,b2,b1,b3)
    half_t sm[PER];
#pragma unroll
    for (int s = 0; s < PER; ++s) sm[s] = st.s[s] * smask;
    GpSlab<NBITS, 0, PER, DW, SUB>::run(w, st.z, sm, o);
  };
  auto put = [&](const uint32_t (&o)[PER][2 * DW], const Set& st, int buf) {   // rebuilt weights + the step's x -> LDS stage `buf`
    uint8_t* ldsW = lds + buf * STAGE;
    uint8_t* ldsX = ldsW + WBYTES;
#pragma unroll
    for (int s = 0; s < PER; ++s)
#pragma unroll
      for (int c = 0; c < DW / 2; ++c)
        *reinterpret_cast<u32x4*>(ldsW + gp_off(s * PROWS + wp, wq * (KPT / 8) + c)) = u32x4{o[s][4 * c], o[s][4 * c + 1], o[s][4 * c + 2], o[s][4 * c + 3]};
#pragma unroll
    for (int i = 0; i < CH; ++i) *reinterpret_cast<u32x4*>(ldsX + gp_off(xr, xc0 + i)) = st.x[i];
  };

  f32x4 acc[4][MT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fq = lane >> 4;

  // ---- software pipeline, three stages deep in registers + two in LDS.  Iteration i:
  //        LDS      fragment reads of step i (stage i & 1); writes of step i + 1 (rebuilt during iteration i - 1) into the other stage
  //        MFMA     step i, and dealt out under the MFMAs (a 16x16x32 MFMA holds the matrix pipe for 8 passes; the wave can issue
  //                 ~3 other instructions under it): the VALU rebuild of step i + 2 into registers, the global loads of step i + 4
  //        one workgroup barrier.
  //      Source order inside an iteration = the order the memory model lets the scheduler keep (reads of one stage, then writes of
  //      the other); the sched_group_barrier sequence does the dealing.  Set (step % 3) holds the global data of `step`. ----
  Set st[GP_NS];
  uint32_t o[PER][2 * DW];
  load(st[0], 0);
  load(st[1], 1);
  load(st[2], 2);
  rebuild(st[0], o);
  put(o, st[0], 0);
  load(st[0], 3);
  rebuild(st[1], o);
  __syncthreads();
  auto iter = [&](int i, Set& n1, Set& n2) {   // n1 / n2: the sets that hold steps i + 1 / i + 2 (past the last step: copies of it)
    const int buf = i & 1;
    const uint8_t* ldsW = lds + buf * STAGE;
    const uint8_t* ldsX = ldsW + WBYTES;
    h8_t fa[2][4], fb[2][MT];
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
#pragma unroll
      for (int q = 0; q < 4; ++q) fa[k2][q] = *reinterpret_cast<const h8_t*>(ldsW + gp_off(wn * 64 + q * 16 + fr, k2 * 4 + fq));
#pragma unroll
      for (int j = 0; j < MT; ++j) fb[k2][j] = *reinterpret_cast<const h8_t*>(ldsX + gp_off(wm * (BM / 2) + j * 16 + fr, k2 * 4 + fq));
    }
#ifndef GP_LAB_NOPUT
    put(o, n1, buf ^ 1);
#endif
#ifndef GP_LAB_NOLOAD
    load(n1, i + 1 + GP_NS);
#endif
#ifndef GP_LAB_NOVALU
    rebuild(n2, o);
#endif
#ifndef GP_LAB_NOMFMA
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[q][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[k2][q], fb[k2][j], acc[q][j], 0, 0, 0);
#else
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
      for (int q = 0; q < 4; ++q) { acc[q][0][0] += static_cast<float>(fa[k2][q][0]); acc[q][1][0] += static_cast<float>(fb[k2][q % MT][0]); }
#endif
    constexpr int NMF = 8 * MT;                       // MFMAs per step
    constexpr int VPM = (96 + NMF - 1) / NMF;         // VALU instructions dealt under each MFMA (rebuild + addressing: ~90 per step)
    __builtin_amdgcn_sched_group_barrier(0x100, 4 + MT, 0);                    // fragment reads of the first k half
    __builtin_amdgcn_sched_group_barrier(0x200, PER * (DW / 2) + CH, 0);       // the LDS writes of step i + 1: their values are ready
    __builtin_amdgcn_sched_group_barrier(0x100, 4 + MT, 0);                    // fragment reads of the second half
#pragma unroll
    for (int t = 0; t < NMF; ++t) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
      if (t % (NMF / 8) == NMF / 8 - 1) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);   // global loads of step i + 4
    }
    __syncthreads();
  };
  for (int i0 = 0; i0 < nsteps; i0 += GP_NS) {
    iter(i0, st[1], st[2]);
    if (i0 + 1 < nsteps) iter(i0 + 1, st[2], st[0]);
    if (i0 + 2 < nsteps) iter(i0 + 2, st[0], st[1]);
  }

  const int tile = mt * a.n_tiles + nt;
  if (a.KS > 1) {   // park the split's fp32 tile in accumulator order (a wave writes 1 KiB of consecutive bytes per instruction); gemm_pipe_reduce_kernel adds them up
    const int64_t tiles = static_cast<int64_t>(a.n_tiles) * a.m_tiles;
    f32x4* mine = reinterpret_cast<f32x4*>(a.part) + (static_cast<int64_t>(ks) * tiles + tile) * (4 * MT * GP_T) + tid;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) __builtin_nontemporal_store(acc[i][j], mine + (i * MT + j) * GP_T);
    return;
  }

  // ---- epilogue: lane holds features (fq * 4 .. + 3) of feature block i, token fr of token block j ----
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int trow = wn * 64 + i * 16 + fq * 4;
    const int slab = trow / PROWS, pin = trow % PROWS;
    const int prow = p0 + pin;
    if (prow >= rows_per_slab) continue;
    const int n = slab * rows_per_slab + prow;
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const int m = m0 + wm * (BM / 2) + j * 16 + fr;
      if (m >= M) continue;
      half_t o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        o[r] = static_cast<half_t>(acc[i][j][r]);
        if (a.bias && prow + r < rows_per_slab) o[r] = o[r] + a.bias[n + r];   // `out += bias` on the rounded matmul result (quantize.py:896-897)
      }
      half_t* dst = a.y + static_cast<int64_t>(m) * N + n;
      if (prow + 3 < rows_per_slab) {
        *reinterpret_cast<u32x2*>(dst) = *reinterpret_cast<u32x2*>(o);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (prow + r < rows_per_slab) dst[r] = o[r];
      }
    }
  }
}

// Second launch of a split-K call: output quad (tile, feature block q, token block j, thread) = the sum of the KS parked tiles in
// split order (all 4 KS loads of a thread in flight at once, every CU takes part), rounded once, + bias.  One finishing workgroup per
// tile inside the first kernel (ticket scheme) read its KS x 64 KiB alone and cost 2-3 us per split.
template <int NBITS, int BM>
__global__ __launch_bounds__(GP_T) void gemm_pipe_reduce_kernel(const GpArgs a) {
  constexpr int PER = 8 / NBITS, PROWS = GP_N / PER, MT = BM / 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wm = wave & 1, fr = lane & 15, fq = lane >> 4;
  const int qj = blockIdx.x % (4 * MT), tile = blockIdx.x / (4 * MT);
  const int q = qj / MT, j = qj % MT;
  const int nt = tile % a.n_tiles, mt = tile / a.n_tiles;
  const int64_t tiles = static_cast<int64_t>(a.n_tiles) * a.m_tiles;
  const f32x4* src = reinterpret_cast<const f32x4*>(a.part) + (static_cast<int64_t>(tile) * (4 * MT) + qj) * GP_T + tid;
  const int64_t kstride = tiles * (4 * MT * GP_T);
  f32x4 sum = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < a.KS; k0 += 4) {
    f32x4 t[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) t[kk] = __builtin_nontemporal_load(src + (k0 + kk < a.KS ? k0 + kk : a.KS - 1) * kstride);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
      if (k0 + kk < a.KS) { sum[0] += t[kk][0]; sum[1] += t[kk][1]; sum[2] += t[kk][2]; sum[3] += t[kk][3]; }
  }
  const int rows_per_slab = a.N / PER;
  const int trow = wn * 64 + q * 16 + fq * 4;
  const int slab = trow / PROWS, pin = trow % PROWS;
  const int prow = nt * PROWS + pin;
  const int m = mt * BM + wm * (BM / 2) + j * 16 + fr;
  if (prow >= rows_per_slab || m >= a.M) return;
  const int n = slab * rows_per_slab + prow;
  half_t o[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    o[r] = static_cast<half_t>(sum[r]);
    if (a.bias && prow + r < rows_per_slab) o[r] = o[r] + a.bias[n + r];
  }
  half_t* dst = a.y + static_cast<int64_t>(m) * a.N + n;
  if (prow + 3 < rows_per_slab) {
    *reinterpret_cast<u32x2*>(dst) = *reinterpret_cast<u32x2*>(o);
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (prow + r < rows_per_slab) dst[r] = o[r];
  }
}

// ---- host side --------------------------------------------------------------------------------------------------------------------
struct GpPlan { int BM, n_tiles, m_tiles, KS, kps; };

// Shapes only (never the data): the split depends on (M, N, K), so a row of y can differ in the last bit between batch sizes that
// choose different splits — as with any split-K GEMM — but is reproducible run to run.
GpPlan gp_plan(int nbits, int64_t M, int64_t N, int64_t K, uint32_t opts) {
  const int per = 8 / nbits;
  GpPlan p;
  p.BM = (M <= 64 || (M > 128 && M <= 192)) ? 64 : 128;
  const int64_t rows_per_slab = N / per;
  p.n_tiles = static_cast<int>((rows_per_slab + GP_N / per - 1) / (GP_N / per));
  p.m_tiles = static_cast<int>((M + p.BM - 1) / p.BM);
  const int64_t tiles = static_cast<int64_t>(p.n_tiles) * p.m_tiles;
  const int nk = static_cast<int>(K / GP_K);
  int ks = 1;
  const int forced = static_cast<int>(opts >> 24);
  if (forced) {
    ks = forced;
  } else if (tiles < 384) {
    ks = static_cast<int>((512 + tiles - 1) / tiles);
  }
  if (ks > GP_MAX_KS) ks = GP_MAX_KS;
  if (ks > nk / 4) ks = nk / 4 > 0 ? nk / 4 : 1;      // at least four steps (256 k) per split
  p.kps = (nk + ks - 1) / ks;
  p.KS = (nk + p.kps - 1) / p.kps;                     // no empty split
  return p;
}

size_t gemm_pipe_workspace_bytes(int nbits, int64_t M, int64_t N, int64_t K, uint32_t opts) {
  const GpPlan p = gp_plan(nbits, M, N, K, opts);
  if (p.KS <= 1) return 0;
  const int64_t tiles = static_cast<int64_t>(p.n_tiles) * p.m_tiles;
  return static_cast<size_t>(p.KS) * tiles * p.BM * GP_N * sizeof(float);
}

bool gemm_pipe_covers(int nbits, int64_t M, int64_t N, int64_t K, int64_t gs, int dtype) {
  if (dtype != HQQ_F16 || (nbits != 8 && nbits != 4 && nbits != 2)) return false;
  const int per = 8 / nbits;
  return N % per == 0 && (N / per) % 4 == 0 && gs >= 32 && (gs & (gs - 1)) == 0 && K % GP_K == 0 && K % gs == 0 && M >= 1;   // (a thread's k piece stays inside one group)
}

template <int NBITS, int BM, bool SUB>
static int gp_launch(const GpArgs& a, int64_t blocks, hipStream_t st) {
  constexpr int lds_bytes = 2 * (GP_N * GP_K * 2 + BM * GP_K * 2);
  hipLaunchKernelGGL((gemm_pipe_f16_kernel<NBITS, BM, SUB>), dim3(static_cast<unsigned>(blocks)), dim3(GP_T), lds_bytes, st, a);
  int rc = check_launch("hqq_hip_gemm(pipelined)");
  if (rc || a.KS <= 1) return rc;
  const int64_t rblocks = static_cast<int64_t>(a.n_tiles) * a.m_tiles * (4 * (BM / 32));
  hipLaunchKernelGGL((gemm_pipe_reduce_kernel<NBITS, BM>), dim3(static_cast<unsigned>(rblocks)), dim3(GP_T), 0, st, a);
  return check_launch("hqq_hip_gemm(split-K reduce)");
}

int gemm_pipe_run(int nbits, const void* x, const void* Wq, const void* scale, const void* zero, const void* bias, void* y,
                  int64_t M, int64_t N, int64_t K, int64_t gs, uint32_t opts, void* workspace, size_t workspace_bytes, hipStream_t st) {
  const GpPlan p = gp_plan(nbits, M, N, K, opts);
  const int64_t tiles = static_cast<int64_t>(p.n_tiles) * p.m_tiles;
  const int64_t blocks = tiles * p.KS;
  if (blocks > INT32_MAX) { set_error("hqq_hip_gemm: grid too large"); return HQQ_ERR_SHAPE; }
  GpArgs a;
  a.x = static_cast<const half_t*>(x); a.Wq = static_cast<const uint8_t*>(Wq); a.scale = static_cast<const half_t*>(scale);
  a.zero = static_cast<const half_t*>(zero); a.bias = static_cast<const half_t*>(bias); a.y = static_cast<half_t*>(y);
  a.part = nullptr;
  a.M = static_cast<int>(M); a.N = static_cast<int>(N); a.K = static_cast<int>(K); a.gs_shift = __builtin_ctzll(static_cast<unsigned long long>(gs)); a.G = static_cast<int>(K / gs);
  a.n_tiles = p.n_tiles; a.m_tiles = p.m_tiles; a.KS = p.KS; a.kps = p.kps;
  if (p.KS > 1) {
    const size_t need = gemm_pipe_workspace_bytes(nbits, M, N, K, opts);
    if (!workspace || workspace_bytes < need || !aligned16(workspace)) {
      set_error("hqq_hip_gemm: workspace %zu < %zu bytes (hqq_hip_forward_workspace_bytes)", workspace_bytes, need);
      return HQQ_ERR_WORKSPACE;
    }
    a.part = static_cast<float*>(workspace);
  }
  const bool sub = (opts & HQQ_OPT_META_SCALABLE) != 0;
#define GP_GO(NB, BM_) (sub ? gp_launch<NB, BM_, true>(a, blocks, st) : gp_launch<NB, BM_, false>(a, blocks, st))
  if (nbits == 8) return p.BM == 64 ? GP_GO(8, 64) : GP_GO(8, 128);
  if (nbits == 4) return p.BM == 64 ? GP_GO(4, 64) : GP_GO(4, 128);
  return p.BM == 64 ? GP_GO(2, 64) : GP_GO(2, 128);
#undef GP_GO
}

}  // namespace hqq
