/*
 * oracle/hqq_oracle.c — CPU restatement of the HQQ hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle for the MI355X kernels in hqq_amd/csrc.  It is NOT part of
 * the product: only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load
 * it, and only as the checker / reported CPU baseline.  Nothing under hqq_amd/ imports it.
 *
 * It restates, in plain C with explicitly ordered IEEE arithmetic, what the reference
 * (mobiusml/hqq v0.2.8.post1, pure PyTorch) computes on its CPU path:
 *
 *   hqq_oracle_pack / hqq_oracle_unpack   hqq/core/bitpack.py:14-144   (BitPack.pack_* / unpack_*)
 *   hqq_oracle_quantize                   hqq/core/quantize.py:75-180  (Quantizer.quantize, axis=1)
 *                                         hqq/core/optimize.py:96-108  (shrink_lp_op)
 *                                         hqq/core/optimize.py:201-255 (optimize_weights_proximal_legacy)
 *   hqq_oracle_quantize_axis0             the same with axis=0 (groups run down the rows of the [gs, numel/gs] view; the group mean
 *                                         is ATen's OUTER-dimension float sum: SumKernel.cpp vectorized_outer_sum)
 *   hqq_oracle_dequantize                 hqq/core/quantize.py:183-199 (Quantizer.dequantize)
 *   hqq_oracle_matmul / _forward          hqq/core/quantize.py:880-898 (HQQLinear.matmul / forward_pytorch)
 *
 * Pinning (see tests/golden/make_golden.py and tests/test_oracle_golden.py): the restatement is
 * checked against outputs of the imported reference itself.  Integer paths (pack/unpack) and the
 * dequantised weights are bit-exact, and so is the solver: on all 46 fixtures — BASELINE configs[0] (1 M weights) and configs[1]
 * (16.7 M weights, sha256 of the packed W_q / zero / scale) included — 0 levels, 0 zero-point bits and 0 scale bits differ from the
 * reference (tests/test_oracle_golden.py asserts array_equal).  Two places where this restatement is NOT ATen's instruction sequence,
 * kept here because they are where a future torch could differ; neither has shown a single differing bit on torch 2.10:
 *   (1) |e|^(p-1): evaluated as (float)pow((double)a, (double)(float)(p-1)), the correctly rounded value — which is what ATen's CPU
 *       kernel returns on every element of every fixture (a float powf differs on 26 % of them: tools/solver_probe.py);
 *   (2) the layer-global early-stop error is summed here in double; ATen sums in float with a thread-count dependent cascade.  It
 *       decides only which iteration stops the loop, and would matter only if two successive errors tied to ~1e-7.
 * The 64-element row mean follows ATen's exact float summation order (aten/src/ATen/native/cpu/
 * SumKernel.cpp: vectorized_inner_sum -> row_sum -> multi_row_sum, 8-float vectors, ilp 4), which was
 * verified bit-exact against torch 2.10 in this image.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define HQQ_ORACLE_VERSION 1

/* ------------------------------------------------------------------------------------------ */
/* half / bfloat16 <-> float, round-to-nearest-even, no hardware dependence                    */
/* ------------------------------------------------------------------------------------------ */
static inline uint32_t f32_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float bits_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static inline float f16_to_f32(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1Fu;
  uint32_t man = h & 0x3FFu;
  if (exp == 0) {
    if (man == 0) return bits_f32(sign);
    /* subnormal: value = man * 2^-24 */
    float v = (float)man * 5.9604644775390625e-08f;
    return sign ? -v : v;
  }
  if (exp == 31) return bits_f32(sign | 0x7F800000u | (man << 13));
  return bits_f32(sign | ((exp + 112u) << 23) | (man << 13));
}

static inline uint16_t f32_to_f16(float f) {
  uint32_t x = f32_bits(f);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t ax = x & 0x7FFFFFFFu;
  if (ax >= 0x7F800000u) { /* inf / nan */
    return (uint16_t)(sign | 0x7C00u | ((ax > 0x7F800000u) ? 0x200u : 0u));
  }
  if (ax >= 0x477FF000u) { /* >= 65520 rounds to inf */
    return (uint16_t)(sign | 0x7C00u);
  }
  if (ax < 0x38800000u) { /* < 2^-14: subnormal half or zero */
    if (ax < 0x33000000u) return (uint16_t)sign; /* < 2^-25 -> 0 (ties at exactly 2^-25 go to even=0) */
    /* use float add trick: adding 0.5 (exponent 2^-1) aligns to 2^-24 ulp */
    float a = bits_f32(ax);
    float r = a + 0.5f; /* a < 2^-14 so ulp(0.5..1) = 2^-24 after ... */
    /* 0.5f has ulp 2^-24; r = 0.5 + round_to_2^-24(a) with RNE */
    uint32_t rb = f32_bits(r) - f32_bits(0.5f);
    return (uint16_t)(sign | rb);
  }
  /* normal */
  uint32_t mant_odd = (ax >> 13) & 1u;
  ax += 0xFFFu + mant_odd; /* RNE on the 13 dropped bits */
  return (uint16_t)(sign | ((ax - 0x38000000u) >> 13));
}

static inline float bf16_to_f32(uint16_t h) { return bits_f32((uint32_t)h << 16); }

static inline uint16_t f32_to_bf16(float f) {
  uint32_t x = f32_bits(f);
  if ((x & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((x >> 16) | 0x40u); /* nan */
  uint32_t lsb = (x >> 16) & 1u;
  x += 0x7FFFu + lsb;
  return (uint16_t)(x >> 16);
}

/* dtype codes shared with include/hqq_hip.h */
enum { HQQ_F32 = 0, HQQ_F16 = 1, HQQ_BF16 = 2 };

static inline float load_cd(const void* p, int64_t i, int dtype) {
  switch (dtype) {
    case HQQ_F16: return f16_to_f32(((const uint16_t*)p)[i]);
    case HQQ_BF16: return bf16_to_f32(((const uint16_t*)p)[i]);
    default: return ((const float*)p)[i];
  }
}
/* round a float to the compute dtype and come back (one rounding of that dtype) */
static inline float round_cd(float v, int dtype) {
  switch (dtype) {
    case HQQ_F16: return f16_to_f32(f32_to_f16(v));
    case HQQ_BF16: return bf16_to_f32(f32_to_bf16(v));
    default: return v;
  }
}
static inline void store_cd(void* p, int64_t i, float v, int dtype) {
  switch (dtype) {
    case HQQ_F16: ((uint16_t*)p)[i] = f32_to_f16(v); break;
    case HQQ_BF16: ((uint16_t*)p)[i] = f32_to_bf16(v); break;
    default: ((float*)p)[i] = v;
  }
}

int hqq_oracle_version(void) { return HQQ_ORACLE_VERSION; }

/* exported for the conversion unit tests */
void hqq_oracle_f32_to_f16(const float* in, uint16_t* out, int64_t n) { for (int64_t i = 0; i < n; ++i) out[i] = f32_to_f16(in[i]); }
void hqq_oracle_f16_to_f32(const uint16_t* in, float* out, int64_t n) { for (int64_t i = 0; i < n; ++i) out[i] = f16_to_f32(in[i]); }
void hqq_oracle_f32_to_bf16(const float* in, uint16_t* out, int64_t n) { for (int64_t i = 0; i < n; ++i) out[i] = f32_to_bf16(in[i]); }

/* ------------------------------------------------------------------------------------------ */
/* BitPack  (hqq/core/bitpack.py)                                                              */
/*   values of `per` row-slabs, `step` rows apart, share one packed element; slab 0 is most    */
/*   significant.  4-bit per=2 (:24-28), 2-bit per=4 (:43-52), 1-bit per=8 (:115-128),          */
/*   3-bit per=10 into int32, rows zero padded to 10*ceil(R/10) (:69-91), 8-bit identity (:14). */
/* ------------------------------------------------------------------------------------------ */
static int per_of(int nbits) {
  switch (nbits) { case 8: return 1; case 4: return 2; case 2: return 4; case 1: return 8; case 3: return 10; default: return 0; }
}

int64_t hqq_oracle_packed_rows(int nbits, int64_t R) {
  int per = per_of(nbits);
  if (!per) return -1;
  if (nbits == 3) return (R + 9) / 10;
  /* bitpack.py: _step = int(len(W_q)/per); the OR of unequal slabs raises in torch -> reject */
  if (R % per) return -2;
  return R / per;
}

/* U: [R, C] uint8 values in [0, 2^nbits).  out: [step, C] uint8, or int32 for 3-bit. */
int hqq_oracle_pack(int nbits, const uint8_t* U, int64_t R, int64_t C, void* out) {
  int per = per_of(nbits);
  int64_t step = hqq_oracle_packed_rows(nbits, R);
  if (step < 0) return (int)step;
  if (nbits == 3) {
    int32_t* o = (int32_t*)out;
    for (int64_t p = 0; p < step; ++p)
      for (int64_t c = 0; c < C; ++c) {
        uint32_t w = 0;
        for (int s = 0; s < 10; ++s) {
          int64_t r = (int64_t)s * step + p;
          uint32_t v = (r < R) ? U[r * C + c] : 0u; /* zero padded rows, bitpack.py:71-76 */
          w |= v << (27 - 3 * s);
        }
        o[p * C + c] = (int32_t)w;
      }
    return 0;
  }
  uint8_t* o = (uint8_t*)out;
  for (int64_t p = 0; p < step; ++p)
    for (int64_t c = 0; c < C; ++c) {
      uint32_t w = 0;
      for (int s = 0; s < per; ++s) {
        /* torch uint8 `<<` wraps modulo 256, so out-of-range values lose their high bits exactly like this cast */
        w |= (uint32_t)(uint8_t)(U[((int64_t)s * step + p) * C + c] << (nbits * (per - 1 - s)));
      }
      o[p * C + c] = (uint8_t)w;
    }
  return 0;
}

/* packed: [P, C];  out: [per*P, C] uint8  (bitpack.py:31-38, 55-64, 95-110, 130-144) */
int hqq_oracle_unpack(int nbits, const void* packed, int64_t P, int64_t C, uint8_t* out) {
  int per = per_of(nbits);
  if (!per) return -1;
  if (nbits == 3) {
    const int32_t* in = (const int32_t*)packed;
    for (int s = 0; s < 10; ++s)
      for (int64_t p = 0; p < P; ++p)
        for (int64_t c = 0; c < C; ++c)
          out[((int64_t)s * P + p) * C + c] = (uint8_t)(((uint32_t)in[p * C + c] >> (27 - 3 * s)) & 7u);
    return 0;
  }
  const uint8_t* in = (const uint8_t*)packed;
  uint32_t mask = (1u << nbits) - 1u;
  for (int s = 0; s < per; ++s)
    for (int64_t p = 0; p < P; ++p)
      for (int64_t c = 0; c < C; ++c)
        out[((int64_t)s * P + p) * C + c] = (uint8_t)((in[p * C + c] >> (nbits * (per - 1 - s))) & mask);
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* ATen float row sum order (SumKernel.cpp), for a contiguous row of n floats, n >= 8           */
/* ------------------------------------------------------------------------------------------ */
static int ceil_log2_i64(int64_t x) {
  if (x <= 2) return 1;
  int b = 0; uint64_t v = (uint64_t)x - 1;
  while (v) { ++b; v >>= 1; }
  return b; /* findLastSet(x-1)+1 */
}

static float aten_row_sum_f32(const float* x, int64_t n) {
  enum { V = 8, ILP = 4, LEVELS = 4 };
  const int64_t vec_size = n / V;          /* number of 8-float vectors              */
  const int64_t size_ilp = vec_size / ILP; /* rows of the (-1, 4)-shaped vector view */
  float acc[LEVELS][ILP][V];
  memset(acc, 0, sizeof(acc));
  int level_power = ceil_log2_i64(size_ilp) / LEVELS; if (level_power < 4) level_power = 4;
  const int64_t level_step = (int64_t)1 << level_power, level_mask = level_step - 1;
  int64_t i = 0;
  for (; i + level_step <= size_ilp;) {
    for (int64_t j = 0; j < level_step; ++j, ++i)
      for (int k = 0; k < ILP; ++k)
        for (int l = 0; l < V; ++l) acc[0][k][l] += x[(i * ILP + k) * V + l];
    for (int j = 1; j < LEVELS; ++j) {
      for (int k = 0; k < ILP; ++k)
        for (int l = 0; l < V; ++l) { acc[j][k][l] += acc[j - 1][k][l]; acc[j - 1][k][l] = 0.f; }
      const int64_t mask = level_mask << (j * level_power);
      if ((i & mask) != 0) break;
    }
  }
  for (; i < size_ilp; ++i)
    for (int k = 0; k < ILP; ++k)
      for (int l = 0; l < V; ++l) acc[0][k][l] += x[(i * ILP + k) * V + l];
  for (int j = 1; j < LEVELS; ++j)
    for (int k = 0; k < ILP; ++k)
      for (int l = 0; l < V; ++l) acc[0][k][l] += acc[j][k][l];
  /* row_sum: leftover vectors go to partial 0, then partials 1..3 are added to partial 0 in order */
  for (int64_t v = size_ilp * ILP; v < vec_size; ++v)
    for (int l = 0; l < V; ++l) acc[0][0][l] += x[v * V + l];
  for (int k = 1; k < ILP; ++k)
    for (int l = 0; l < V; ++l) acc[0][0][l] += acc[0][k][l];
  /* vectorized_inner_sum: scalar tail first, then the 8 lanes in order */
  float fin = 0.f;
  for (int64_t t = vec_size * V; t < n; ++t) fin += x[t];
  for (int l = 0; l < V; ++l) fin += acc[0][0][l];
  return fin;
}

float hqq_oracle_row_sum_f32(const float* x, int64_t n) { return aten_row_sum_f32(x, n); }

/* ------------------------------------------------------------------------------------------ */
/* Quantizer.quantize, axis=1, channel_wise=True  (quantize.py:75-180) with the legacy HQ        */
/* proximal solver (optimize.py:201-255) in float32 (the reference's CPU precision, :231).        */
/*                                                                                              */
/*  W        [numel] float32 (already `tensor.float()`, quantize.py:102), viewed as [R, gs]      */
/*  Wq_out   [R*gs]  uint8 quantised levels (before packing)                                     */
/*  scale_out[R]     float32 = 1/scale   (quantize.py:154: meta stores the inverse)              */
/*  zero_out [R]     float32                                                                     */
/*  err_hist [iters] double: layer mean |W_f - W_r| of every iteration that ran (may be NULL)    */
/*  returns the number of solver iterations executed (<= iters), or < 0 on bad arguments.        */
/* ------------------------------------------------------------------------------------------ */
int hqq_oracle_quantize(const float* W, int64_t numel, int gs, int max_v, int round_zero, int optimize,
                        int iters, float beta, float lp_norm,
                        uint8_t* Wq_out, float* scale_out, float* zero_out, double* err_hist) {
  if (gs <= 0 || numel % gs) return -1;   /* quantize.py:94-100 */
  if (gs < 8) return -2;                  /* row-sum restatement assumes ATen's vectorised path */
  const int64_t R = numel / gs;
  float* s = (float*)malloc(sizeof(float) * (size_t)R);
  float* z = (float*)malloc(sizeof(float) * (size_t)R);
  if (!s || !z) { free(s); free(z); return -3; }
  const float maxv = (float)max_v;

  /* min/max init, quantize.py:118-134 */
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < R; ++r) {
    const float* w = W + r * gs;
    float mn = w[0], mx = w[0];
    for (int c = 1; c < gs; ++c) { if (w[c] < mn) mn = w[c]; if (w[c] > mx) mx = w[c]; }
    float denom = mx - mn;
    /* `max_v / denom` is Tensor.__rtruediv__ = denom.reciprocal() * max_v : two roundings */
    float sc = (1.0f / denom) * maxv;
    if (fabsf(denom) <= 1e-4f) sc = 1.0f; /* :128 (threshold is compared in float32) */
    if (sc > 2e4f) sc = 2e4f;             /* :129 clamp(max=2e4) */
    float ze = (-mn) * sc;                /* :130 */
    if (round_zero) ze = rintf(ze);       /* :133-134, half-to-even */
    s[r] = sc; z[r] = ze;
  }

  int ran = 0;
  if (optimize) {
    /* optimize.py:208-255.  kappa is read but never applied (:219-224 vs :237-247). */
    const float inv_beta = (float)(1.0 / (double)beta);    /* python double 1.0/beta, cast to float by the mul kernel */
    const float pexp_f = (float)((double)lp_norm - 1.0);    /* pow exponent cast to the tensor dtype */
    const double pexp = (double)pexp_f;
    float* znew = (float*)malloc(sizeof(float) * (size_t)R);
    if (!znew) { free(s); free(z); return -3; }
    float best = INFINITY;
    for (int it = 0; it < iters; ++it) {
      double err_sum = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : err_sum)
      for (int64_t r = 0; r < R; ++r) {
        const float* w = W + r * gs;
        const float sc = s[r], ze = z[r];
        float t3[256];
        float* buf = (gs <= 256) ? t3 : (float*)malloc(sizeof(float) * (size_t)gs);
        double e_acc = 0.0;
        for (int c = 0; c < gs; ++c) {
          const float wf = w[c];
          float q = wf * sc;        /* optimize.py:202  round(W_f*scale + zero): mul then add, unfused */
          q = q + ze;
          q = rintf(q);
          q = fminf(fmaxf(q, 0.f), maxv);
          float wr = (q - ze) / sc; /* :203 */
          float e = wf - wr;        /* :204 */
          float a = fabsf(e);
          e_acc += (double)a;       /* :239 |W_f - W_r| */
          /* shrink_lp_op, optimize.py:96-108 */
          float we;
          if (lp_norm == 1.0f) {
            float t = a - inv_beta;
            t = (t < 0.f) ? 0.f : t;
            we = t * ((e > 0.f) - (e < 0.f));
          } else {
            float pw = (float)pow((double)a, pexp);   /* a=0 -> +inf */
            float t = inv_beta * pw;
            t = a - t;                                /* 0 - inf = -inf */
            t = (t < 0.f || t != t) ? ((t != t) ? t : 0.f) : t; /* clamp_min_(0) (NaN propagates) */
            we = t * (float)((e > 0.f) - (e < 0.f));  /* mul_(sign(x)); -inf path already clamped */
          }
          float u = wf - we;        /* :205 */
          u = u * sc;
          buf[c] = q - u;
        }
        znew[r] = aten_row_sum_f32(buf, gs) / (float)gs; /* mean = sum / n */
        if (buf != t3) free(buf);
        err_sum += e_acc;
      }
      /* zero is overwritten before the early-stop test (:238, :244-247) */
      memcpy(z, znew, sizeof(float) * (size_t)R);
      float cur = (float)(err_sum / (double)numel);
      if (err_hist) err_hist[it] = err_sum / (double)numel;
      ran = it + 1;
      if (cur < best) best = cur; else break;
    }
    free(znew);
  }

  /* final W_q from the float32 tensor with the returned scale/zero (optimize.py:254 / quantize.py:147) */
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < R; ++r) {
    const float* w = W + r * gs;
    for (int c = 0; c < gs; ++c) {
      float q = w[c] * s[r];
      q = q + z[r];
      q = rintf(q);
      q = fminf(fmaxf(q, 0.f), maxv);
      Wq_out[r * gs + c] = (uint8_t)q;
    }
    scale_out[r] = 1.0f / s[r]; /* quantize.py:154 */
    zero_out[r] = z[r];
  }
  free(s); free(z);
  return ran;
}

/* ------------------------------------------------------------------------------------------ */
/* ATen float sum over the OUTER dimension of a contiguous [n, C] tensor (torch.mean(dim=0)):     */
/* SumKernel.cpp vectorized_outer_sum.  Columns are taken 32 at a time (four 8-float vectors)     */
/* through multi_row_sum — a cascade over the rows: 16 rows into level 0, level 0 into level 1    */
/* ... —, a remaining run of 8 columns and the last C % 8 columns through row_sum (four           */
/* interleaved partial sums over rows i % 4, each a cascade; leftover rows; partials 1..3 added   */
/* to partial 0 in order).  Verified bit-exact against torch 2.10 in this image for n = 8..256,   */
/* C = 7..65536.  x(i) = base[i * stride].                                                        */
/* ------------------------------------------------------------------------------------------ */
static float cascade_sum_strided(const float* base, int64_t stride, int64_t n) {
  enum { LEVELS = 4 };
  float acc[LEVELS] = {0.f, 0.f, 0.f, 0.f};
  int level_power = ceil_log2_i64(n) / LEVELS; if (level_power < 4) level_power = 4;
  const int64_t level_step = (int64_t)1 << level_power, level_mask = level_step - 1;
  int64_t i = 0;
  for (; i + level_step <= n;) {
    for (int64_t j = 0; j < level_step; ++j, ++i) acc[0] += base[i * stride];
    for (int j = 1; j < LEVELS; ++j) {
      acc[j] += acc[j - 1]; acc[j - 1] = 0.f;
      if ((i & (level_mask << (j * level_power))) != 0) break;
    }
  }
  for (; i < n; ++i) acc[0] += base[i * stride];
  for (int j = 1; j < LEVELS; ++j) acc[0] += acc[j];
  return acc[0];
}
static float aten_col_sum_f32(const float* col, int64_t stride, int64_t n, int64_t j, int64_t C) {
  if (j < (C / 32) * 32) return cascade_sum_strided(col, stride, n);
  /* row_sum: the column read as a (-1, 4) array -> four partial sums, each a cascade over every fourth row */
  const int64_t size_ilp = n / 4;
  float part[4];
  for (int k = 0; k < 4; ++k) part[k] = cascade_sum_strided(col + k * stride, 4 * stride, size_ilp);
  for (int64_t i = size_ilp * 4; i < n; ++i) part[0] += col[i * stride];
  for (int k = 1; k < 4; ++k) part[0] += part[k];
  return part[0];
}
float hqq_oracle_col_sum_f32(const float* col, int64_t stride, int64_t n, int64_t j, int64_t C) { return aten_col_sum_f32(col, stride, n, j, C); }

/* ------------------------------------------------------------------------------------------ */
/* Quantizer.quantize, axis=0, channel_wise=True (quantize.py:75-180): W viewed as [gs, C],      */
/* C = numel / gs; group j is column j.  Same solver (optimize.py:201-255), statistics and mean   */
/* along axis 0.  Wq_out [gs*C] uint8 in the [gs, C] layout (what BitPack.pack_* then packs row   */
/* slab by row slab), scale_out [C] = 1/scale, zero_out [C].                                      */
/* ------------------------------------------------------------------------------------------ */
int hqq_oracle_quantize_axis0(const float* W, int64_t numel, int gs, int max_v, int round_zero, int optimize,
                              int iters, float beta, float lp_norm,
                              uint8_t* Wq_out, float* scale_out, float* zero_out, double* err_hist) {
  if (gs <= 0 || numel % gs) return -1;
  const int64_t C = numel / gs;
  if (C < 8) return -2;                   /* ATen sums fewer than 8 columns on its scalar outer path: another order, not restated */
  float* s = (float*)malloc(sizeof(float) * (size_t)C);
  float* z = (float*)malloc(sizeof(float) * (size_t)C);
  float* t3 = (float*)malloc(sizeof(float) * (size_t)numel);
  if (!s || !z || !t3) { free(s); free(z); free(t3); return -3; }
  const float maxv = (float)max_v;
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < C; ++j) {
    float mn = W[j], mx = W[j];
    for (int i = 1; i < gs; ++i) { const float w = W[(int64_t)i * C + j]; if (w < mn) mn = w; if (w > mx) mx = w; }
    float denom = mx - mn;
    float sc = (1.0f / denom) * maxv;
    if (fabsf(denom) <= 1e-4f) sc = 1.0f;
    if (sc > 2e4f) sc = 2e4f;
    float ze = (-mn) * sc;
    if (round_zero) ze = rintf(ze);
    s[j] = sc; z[j] = ze;
  }
  int ran = 0;
  if (optimize) {
    const float inv_beta = (float)(1.0 / (double)beta);
    const double pexp = (double)(float)((double)lp_norm - 1.0);
    float best = INFINITY;
    for (int it = 0; it < iters; ++it) {
      double err_sum = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : err_sum)
      for (int64_t j = 0; j < C; ++j) {
        const float sc = s[j], ze = z[j];
        double e_acc = 0.0;
        for (int i = 0; i < gs; ++i) {
          const int64_t idx = (int64_t)i * C + j;
          const float wf = W[idx];
          float q = wf * sc; q = q + ze; q = rintf(q); q = fminf(fmaxf(q, 0.f), maxv);
          float wr = (q - ze) / sc;
          float e = wf - wr;
          float a = fabsf(e);
          e_acc += (double)a;
          float we;
          if (lp_norm == 1.0f) {
            float t = a - inv_beta; t = (t < 0.f) ? 0.f : t;
            we = t * (float)((e > 0.f) - (e < 0.f));
          } else {
            float pw = (float)pow((double)a, pexp);
            float t = inv_beta * pw; t = a - t;
            t = (t < 0.f || t != t) ? ((t != t) ? t : 0.f) : t;
            we = t * (float)((e > 0.f) - (e < 0.f));
          }
          float u = wf - we; u = u * sc;
          t3[idx] = q - u;
        }
        err_sum += e_acc;
      }
#pragma omp parallel for schedule(static)
      for (int64_t j = 0; j < C; ++j) z[j] = aten_col_sum_f32(t3 + j, C, gs, j, C) / (float)gs;
      float cur = (float)(err_sum / (double)numel);
      if (err_hist) err_hist[it] = err_sum / (double)numel;
      ran = it + 1;
      if (cur < best) best = cur; else break;
    }
  }
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < C; ++j) {
    for (int i = 0; i < gs; ++i) {
      const int64_t idx = (int64_t)i * C + j;
      float q = W[idx] * s[j]; q = q + z[j]; q = rintf(q); q = fminf(fmaxf(q, 0.f), maxv);
      Wq_out[idx] = (uint8_t)q;
    }
    scale_out[j] = 1.0f / s[j];
    zero_out[j] = z[j];
  }
  free(s); free(z); free(t3);
  return ran;
}

/* ------------------------------------------------------------------------------------------ */
/* Quantizer.dequantize (quantize.py:183-199), axis=1:                                          */
/*   W_r = unpack(W_q).to(cd)[:R];  W = ((W_r - zero) * scale).reshape(N, K)                    */
/* each of the two ops rounds once to the compute dtype `cd`.  scale/zero are given in `cd`.     */
/* packed: [P, gs] ; out: [N*K] in cd ; R = N*K/gs.                                              */
/* ------------------------------------------------------------------------------------------ */
int hqq_oracle_dequantize(int nbits, const void* packed, const void* scale, const void* zero, void* out,
                          int64_t N, int64_t K, int gs, int dtype) {
  const int per = per_of(nbits);
  if (!per || gs <= 0 || (N * K) % gs) return -1;
  const int64_t R = N * K / gs;
  const int64_t P = hqq_oracle_packed_rows(nbits, R);
  if (P < 0) return -2;
  const uint32_t mask = (nbits == 8) ? 0xFFu : ((1u << nbits) - 1u);
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < R; ++r) {
    const int64_t slot = r / P, p = r % P;
    const float ze = load_cd(zero, r, dtype), sc = load_cd(scale, r, dtype);
    for (int c = 0; c < gs; ++c) {
      uint32_t q;
      if (nbits == 3) q = ((uint32_t)((const int32_t*)packed)[p * gs + c] >> (27 - 3 * (int)slot)) & 7u;
      else q = ((uint32_t)((const uint8_t*)packed)[p * gs + c] >> (nbits * (per - 1 - (int)slot))) & mask;
      float d = round_cd((float)q - ze, dtype);  /* exact in float, one rounding to cd */
      float w = round_cd(d * sc, dtype);         /* product exact in float for f16/bf16 operands */
      store_cd(out, r * gs + c, w, dtype);
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* HQQLinear.matmul / forward_pytorch (quantize.py:880-898): y = x @ W^T (+ bias).             */
/* x [M,K], Wd [N,K] (dequantised, compute dtype), bias [N] or NULL, y [M,N] compute dtype.     */
/* BLAS accumulation order is unspecified in the reference; the oracle accumulates in double    */
/* (the value every fp32-accumulating order approximates) and rounds once to the compute dtype. */
/* y32 (optional, may be NULL) receives the un-rounded float value for tolerance checks.        */
/* ------------------------------------------------------------------------------------------ */
int hqq_oracle_matmul(const void* x, const void* Wd, const void* bias, void* y, float* y32,
                      int64_t M, int64_t N, int64_t K, int dtype) {
  float* xf = (float*)malloc(sizeof(float) * (size_t)(M * K));
  if (!xf) return -3;
  for (int64_t i = 0; i < M * K; ++i) xf[i] = load_cd(x, i, dtype);
#pragma omp parallel for schedule(static)
  for (int64_t n = 0; n < N; ++n) {
    float* wrow = (float*)malloc(sizeof(float) * (size_t)K);
    for (int64_t k = 0; k < K; ++k) wrow[k] = load_cd(Wd, n * K + k, dtype);
    for (int64_t m = 0; m < M; ++m) {
      double acc = 0.0;
      const float* xr = xf + m * K;
      for (int64_t k = 0; k < K; ++k) acc += (double)xr[k] * (double)wrow[k];
      float v = (float)acc;
      if (bias) v = round_cd(v, dtype) + load_cd(bias, n, dtype); /* `out += bias` on the rounded matmul result */
      if (y32) y32[m * N + n] = v;
      store_cd(y, m * N + n, v, dtype);
    }
    free(wrow);
  }
  free(xf);
  return 0;
}

/* dequantize + matmul in one call (what HQQBackend.PYTORCH does per forward); used as the CPU baseline */
int hqq_oracle_forward(int nbits, const void* packed, const void* scale, const void* zero, const void* bias,
                       const void* x, void* y, int64_t M, int64_t N, int64_t K, int gs, int dtype) {
  const size_t esz = (dtype == HQQ_F32) ? 4 : 2;
  void* Wd = malloc(esz * (size_t)(N * K));
  if (!Wd) return -3;
  int rc = hqq_oracle_dequantize(nbits, packed, scale, zero, Wd, N, K, gs, dtype);
  if (!rc) rc = hqq_oracle_matmul(x, Wd, bias, y, NULL, M, N, K, dtype);
  free(Wd);
  return rc;
}
