// block.hip — the steps either side of the fused GEMVs in a decode step, gfx950, fp16 and bf16.
//
// The reference's headline number is the tok/s of its generate loop (hqq/utils/generation_hf.py:117-540, Readme.md:153): HF's decoder
// block around HQQLinear.forward — RMSNorm, rotary embedding, KV-cache update, SiLU(gate) * up, the residual adds — is ~25 small
// eager kernels per block, 79 % of a bs = 1 token once the linears are fused.  Three kernels replace twenty of them;
// each restates the HF module's arithmetic rounding for rounding, so that the fused loop emits the same tokens (T = the model's dtype):
//   add_rmsnorm   h += delta (one rounding: `residual + hidden_states`), then LlamaRMSNorm: fp32 x * rsqrt(mean(x^2) + eps) -> T -> weight * (a product in T)
//   rope_cache    apply_rotary_pos_emb: (q * cos) + (rotate_half(q) * sin) with three roundings to T, k likewise, and the StaticCache
//                 update (k_rot / v written at cache_position, read from device memory: graph-replay safe)
//   silu_mul      LlamaMLP: act_fn(gate) * up — silu in fp32 (x / (1 + exp(-x))), rounded to T, then the product in T
// T = fp16: native half arithmetic.  T = bf16: float arithmetic + one round-to-nearest-even per op, which is how torch evaluates bf16 elementwise ops.
// And one that does NOT restate a kernel bit for bit (opt-in, FusedLlamaStep(attention="hip")):
//   attn_decode   softmax(q K^T * scaling) V for ONE query per head over the static KV cache's first pos + 1 positions, fp32 scores / softmax /
//                 accumulation, one rounding of the output: what F.scaled_dot_product_attention computes for a decode step, within
//                 rounding of it (SDPA's flash kernel blocks the keys and rounds P to T; this one does neither) — 8 us instead of the
//                 12-15 us the library's prefill-shaped kernel takes for a single query (profiles/r04_e2e_kernel_times.txt)
// Compiled with -ffp-contract=off: a fused multiply-add would remove a rounding HF's separate ops make.
#include "hqq_common.h"
#include "block_math.h"

namespace hqq {

// ---- h (+= delta), xn = weight * T(float(h) * rsqrt(mean(float(h)^2) + eps)): one workgroup of 512 threads per row.
//      Rows of up to 512 * 8 * RPT elements stay in registers between the two passes (every load is issued before the reduction: the kernel is
//      one memory round trip + one barrier long); longer rows take the generic two-pass path ----
template <int RPT, bool BF>   // RPT: 16-byte chunks per thread held in registers; 0: re-read
__global__ __launch_bounds__(512) void add_rmsnorm_kernel(uint16_t* __restrict__ h, const uint16_t* __restrict__ delta, const uint16_t* __restrict__ weight, float eps,
                                                          uint16_t* __restrict__ xn, int H) {
  using E = El<BF>;
  __shared__ float part[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint16_t* hr = h + static_cast<int64_t>(blockIdx.x) * H;
  const uint16_t* dr = delta ? delta + static_cast<int64_t>(blockIdx.x) * H : nullptr;
  uint16_t* xr = xn + static_cast<int64_t>(blockIdx.x) * H;
  constexpr int NR = RPT > 0 ? RPT : 1;
  u32x4 hv[NR], wv[NR];
  float sum = 0.f;
  if constexpr (RPT > 0) {
    u32x4 dv[NR];
#pragma unroll
    for (int c = 0; c < NR; ++c) {
      const int i = (c * 512 + tid) * 8;
      if (i < H) {
        hv[c] = *reinterpret_cast<const u32x4*>(hr + i);
        if (dr) dv[c] = *reinterpret_cast<const u32x4*>(dr + i);
        wv[c] = *reinterpret_cast<const u32x4*>(weight + i);
      }
    }
#pragma unroll
    for (int c = 0; c < NR; ++c) {
      const int i = (c * 512 + tid) * 8;
      if (i < H) {
        uint16_t* hp = reinterpret_cast<uint16_t*>(&hv[c]);
        if (dr) {
          const uint16_t* dp = reinterpret_cast<const uint16_t*>(&dv[c]);
#pragma unroll
          for (int j = 0; j < 8; ++j) hp[j] = E::add(hp[j], dp[j]);   // residual + hidden_states, one rounding
          *reinterpret_cast<u32x4*>(hr + i) = hv[c];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float f = E::f(hp[j]); sum += f * f; }
      }
    }
  } else {
    for (int i = tid * 8; i < H; i += 512 * 8) {
      u32x4 v = *reinterpret_cast<const u32x4*>(hr + i);
      uint16_t* hp = reinterpret_cast<uint16_t*>(&v);
      if (dr) {
        const u32x4 dv = *reinterpret_cast<const u32x4*>(dr + i);
        const uint16_t* dp = reinterpret_cast<const uint16_t*>(&dv);
#pragma unroll
        for (int j = 0; j < 8; ++j) hp[j] = E::add(hp[j], dp[j]);
        *reinterpret_cast<u32x4*>(hr + i) = v;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float f = E::f(hp[j]); sum += f * f; }
    }
  }
  // wave sum (shuffles), then the eight waves through LDS, fixed order
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
  if (lane == 0) part[wave] = sum;
  __syncthreads();
  const float total = ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
  const float r = rsqrtf(total / static_cast<float>(H) + eps);
  if constexpr (RPT > 0) {
#pragma unroll
    for (int c = 0; c < NR; ++c) {
      const int i = (c * 512 + tid) * 8;
      if (i < H) {
        const uint16_t* hp = reinterpret_cast<const uint16_t*>(&hv[c]);
        const uint16_t* wp = reinterpret_cast<const uint16_t*>(&wv[c]);
        u32x4 ov;
        uint16_t* op = reinterpret_cast<uint16_t*>(&ov);
#pragma unroll
        for (int j = 0; j < 8; ++j) op[j] = E::mul(wp[j], E::r_prod(E::f(hp[j]), r));
        *reinterpret_cast<u32x4*>(xr + i) = ov;
      }
    }
  } else {
    for (int i = tid * 8; i < H; i += 512 * 8) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(hr + i);
      const u32x4 w = *reinterpret_cast<const u32x4*>(weight + i);
      const uint16_t* hp = reinterpret_cast<const uint16_t*>(&v);
      const uint16_t* wp = reinterpret_cast<const uint16_t*>(&w);
      u32x4 ov;
      uint16_t* op = reinterpret_cast<uint16_t*>(&ov);
#pragma unroll
      for (int j = 0; j < 8; ++j) op[j] = E::mul(wp[j], E::r_prod(E::f(hp[j]), r));
      *reinterpret_cast<u32x4*>(xr + i) = ov;
    }
  }
}

// ---- rotary embedding of q and k, KV-cache write.  One thread per (head, i < hd / 2): elements i and i + hd / 2 of a head ----
template <bool BF>
__global__ __launch_bounds__(256) void rope_cache_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ k, const uint16_t* __restrict__ v,
                                                         const uint16_t* __restrict__ cosv, const uint16_t* __restrict__ sinv, const int64_t* __restrict__ pos,
                                                         uint16_t* __restrict__ q_out, uint16_t* __restrict__ k_cache, uint16_t* __restrict__ v_cache,
                                                         int n_heads, int n_kv, int hd, int cache_len) {
  const int half = hd / 2;
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = (n_heads + n_kv) * half;
  if (id >= total) return;
  const int head = id / half, i = id - head * half;
  const bool is_k = head >= n_heads;
  const uint16_t* src = is_k ? k + static_cast<int64_t>(head - n_heads) * hd : q + static_cast<int64_t>(head) * hd;
  uint16_t o1, o2;
  rope_pair<BF>(src[i], src[i + half], cosv[i], cosv[i + half], sinv[i], sinv[i + half], o1, o2);
  if (!is_k) {
    q_out[static_cast<int64_t>(head) * hd + i] = o1;
    q_out[static_cast<int64_t>(head) * hd + i + half] = o2;
  } else {
    const int64_t p = pos[0];
    if (p < 0 || p >= cache_len) return;   // a position outside the cache writes nothing (HF's StaticCache index_copy_ raises a device assert there)
    const int kh = head - n_heads;
    uint16_t* kd = k_cache + (static_cast<int64_t>(kh) * cache_len + p) * hd;
    uint16_t* vd = v_cache + (static_cast<int64_t>(kh) * cache_len + p) * hd;
    kd[i] = o1;
    kd[i + half] = o2;
    const uint16_t* vs = v + static_cast<int64_t>(kh) * hd;
    vd[i] = vs[i];
    vd[i + half] = vs[i + half];
  }
}

// ---- out = T(silu(gate)) * up ----
template <bool BF>
__global__ __launch_bounds__(256) void silu_mul_kernel(const uint16_t* __restrict__ g, const uint16_t* __restrict__ u, uint16_t* __restrict__ out, int64_t n) {
  const int64_t i = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) * 8;
  if (i >= n) return;
  const u32x4 gv = *reinterpret_cast<const u32x4*>(g + i);
  const u32x4 uv = *reinterpret_cast<const u32x4*>(u + i);
  const uint16_t* gp = reinterpret_cast<const uint16_t*>(&gv);
  const uint16_t* up = reinterpret_cast<const uint16_t*>(&uv);
  u32x4 ov;
  uint16_t* op = reinterpret_cast<uint16_t*>(&ov);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    op[j] = silu_mul_el<BF>(gp[j], up[j]);
  }
  *reinterpret_cast<u32x4*>(out + i) = ov;
}

// ---- decode attention: one workgroup of 512 threads per query head.  Phase 1: a LANE per key (its 2 HD bytes in 16-byte loads, q broadcast
//      from LDS, v_dot2 into fp32), scores into LDS, workgroup maximum.  Phase 2: exp(s - max) in place, workgroup sum.  Phase 3: a wave
//      per key (keys dealt round-robin to the 8 waves), a lane per pair of dims: o += p V[j]; the 8 partial vectors are added in wave order.
//      Deterministic: no atomics, fixed orders.  Keys beyond pos are never read.
//      ROPE = true (hqq_hip_rope_attn_decode): q, k, v are the RAW projections; the workgroup applies the rotary embedding to its query and to its
//      KV head's new key itself (rope_cache_kernel's arithmetic, rounding for rounding), uses the new key / value from LDS for position pos — the
//      cache is only read below pos, so no workgroup depends on another's write — and the first query head of each KV head writes them to the cache ----
template <int HD, bool ROPE, bool BF>
__global__ __launch_bounds__(512) void attn_decode_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ kc_in, const uint16_t* __restrict__ vc_in,
                                                          const int64_t* __restrict__ pos, uint16_t* __restrict__ out, int n_heads, int n_kv, int L, float scaling,
                                                          const uint16_t* __restrict__ k_raw, const uint16_t* __restrict__ v_raw, const uint16_t* __restrict__ cosv,
                                                          const uint16_t* __restrict__ sinv, uint16_t* __restrict__ kc_out, uint16_t* __restrict__ vc_out,
                                                          int S, float* __restrict__ ws) {
  using E = El<BF>;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  float* red = reinterpret_cast<float*>(smem);                 // [16] reduction scratch
  uint16_t* qs = reinterpret_cast<uint16_t*>(smem + 64);        // [HD] the query; ROPE: + [HD] the new key, [HD] the new value
  float* part = reinterpret_cast<float*>(smem + 64 + HD * 6);   // [8][HD]
  float* sc = part + 8 * HD;                                    // [n] scores, then probabilities
  uint16_t* knew = qs + HD;
  uint16_t* vnew = qs + 2 * HD;
  const int h = blockIdx.x, sp = blockIdx.y, rep = n_heads / n_kv, kvh = h / rep;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // a position outside [0, L) must not index the cache or the score buffer (sized from L): attend as if at the last slot, write nothing
  const int64_t p_raw = pos[0];
  const bool p_ok = p_raw >= 0 && p_raw < L;
  const int p0 = p_ok ? static_cast<int>(p_raw) : L - 1;
  const int n = p0 + 1;
  // S > 1 (long caches): workgroup (h, sp) attends over keys [k0, k1), a 1 / S share of the pos + 1 visible ones, and parks (max, sum,
  // unnormalised output) in the workspace; attn_combine_kernel merges the S shares in split order.  One workgroup per head streams 1.3 TB/s:
  // 48 us at 4096 keys; eight per head 19 (the 64 MB of cache at 3.4 TB/s + the two launches)
  const int chunk = (n + S - 1) / S;
  const int k0 = sp * chunk, k1 = (k0 + chunk < n) ? k0 + chunk : n;
  const uint16_t* K = kc_in + static_cast<int64_t>(kvh) * L * HD;
  const uint16_t* V = vc_in + static_cast<int64_t>(kvh) * L * HD;
  if constexpr (ROPE) {
    // thread t < HD / 2: elements t and t + HD / 2 of the query; HD / 2 <= t < HD: of the new key; HD <= t < HD + HD / 8: a 16-byte chunk of the new value
    constexpr int half = HD / 2;
    if (tid < HD) {
      const bool is_k = tid >= half;
      const int i = is_k ? tid - half : tid;
      const uint16_t* src = is_k ? k_raw + static_cast<int64_t>(kvh) * HD : q + static_cast<int64_t>(h) * HD;
      uint16_t o1, o2;
      rope_pair<BF>(src[i], src[i + half], cosv[i], cosv[i + half], sinv[i], sinv[i + half], o1, o2);
      uint16_t* dst = is_k ? knew : qs;
      dst[i] = o1;
      dst[i + half] = o2;
      if (is_k && h % rep == 0 && sp == 0 && p_ok) {
        uint16_t* kd = kc_out + (static_cast<int64_t>(kvh) * L + p0) * HD;
        kd[i] = o1;
        kd[i + half] = o2;
      }
    } else if (tid < HD + HD / 8) {
      const int c = tid - HD;
      const u32x4 vv = reinterpret_cast<const u32x4*>(v_raw + static_cast<int64_t>(kvh) * HD)[c];
      reinterpret_cast<u32x4*>(vnew)[c] = vv;
      if (h % rep == 0 && sp == 0 && p_ok) reinterpret_cast<u32x4*>(vc_out + (static_cast<int64_t>(kvh) * L + p0) * HD)[c] = vv;
    }
  } else {
    if (tid < HD / 8) reinterpret_cast<u32x4*>(qs)[tid] = reinterpret_cast<const u32x4*>(q + static_cast<int64_t>(h) * HD)[tid];
  }
  __syncthreads();
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  typedef __bf16 b2 __attribute__((ext_vector_type(2)));
  // phase 1: like phase 3 below, a wave instruction covers KPW = 512 / HD whole key rows (LPK = HD / 8 lanes per row, 16 bytes per lane:
  //          1 KiB of consecutive memory) — a lane per key read 64 different lines per instruction and thrashed the L1;
  //          the LPK partial dot products of a row are added by shuffles
  constexpr int LPK = HD / 8, KPW = 64 / LPK, STEP = 8 * KPW;
  const int sub = lane / LPK, ch = lane - sub * LPK;
  const u32x4 qf = reinterpret_cast<const u32x4*>(qs)[ch];
  auto krow = [&](int j) -> u32x4 {
    const uint16_t* kr = (ROPE && j == p0) ? knew : K + static_cast<int64_t>(j) * HD;   // the new key: from LDS, its cache row is being written by another workgroup
    return reinterpret_cast<const u32x4*>(kr)[ch];
  };
  auto score = [&](const u32x4& kv) -> float {
    float acc = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t qe = qf[e], ke = kv[e];   // (a bit_cast of an ext-vector ELEMENT reads element 0: copy to a scalar first)
      if constexpr (BF) acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b2, qe), __builtin_bit_cast(b2, ke), acc, false);
      else acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, qe), __builtin_bit_cast(h2, ke), acc, false);
    }
#pragma unroll
    for (int off = 1; off < LPK; off <<= 1) acc += __shfl_xor(acc, off, 64);
    return acc * scaling;
  };
  float mx = -INFINITY;
  {
    int j = k0 + wave * KPW + sub;
    for (; j + 3 * STEP < k1; j += 4 * STEP) {
      u32x4 k4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) k4[u] = krow(j + u * STEP);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float sv = score(k4[u]);
        if (ch == 0) sc[j + u * STEP - k0] = sv;
        mx = fmaxf(mx, sv);
      }
    }
    // (the shuffles of score() need every lane of a row group: rows past the end are computed on row k1 - 1 and dropped)
    for (; j - sub < k1; j += STEP) {
      const bool live = j < k1;
      const float sv = score(krow(live ? j : k1 - 1));
      if (live) {
        if (ch == 0) sc[j - k0] = sv;
        mx = fmaxf(mx, sv);
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = red[0];
#pragma unroll
  for (int w = 1; w < 8; ++w) mx = fmaxf(mx, red[w]);
  // phase 2
  float sum = 0.f;
  for (int j = tid; j < k1 - k0; j += 512) {
    const float pj = __expf(sc[j] - mx);
    sc[j] = pj;
    sum += pj;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
  if (lane == 0) red[8 + wave] = sum;
  __syncthreads();
  sum = ((red[8] + red[9]) + (red[10] + red[11])) + ((red[12] + red[13]) + (red[14] + red[15]));
  // phase 3: a wave instruction covers KPW = 512 / HD value rows: LPK = HD / 8 lanes per row, 16 bytes (8 dims) per lane; a wave takes rows
  //          k0 + KPW wave + sub, stepping 8 KPW, four instructions in flight; the KPW row groups of a wave are added by shuffles
  float o8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o8[e] = 0.f;
  auto vrow = [&](int j) -> u32x4 {
    const uint16_t* vr = (ROPE && j == p0) ? vnew : V + static_cast<int64_t>(j) * HD;
    return reinterpret_cast<const u32x4*>(vr)[ch];
  };
  auto fold = [&](const u32x4& v, float pj) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t w = v[e];   // (element copied to a scalar before its halves are taken)
      o8[2 * e] = fmaf(pj, E::f(static_cast<uint16_t>(w & 0xFFFFu)), o8[2 * e]);
      o8[2 * e + 1] = fmaf(pj, E::f(static_cast<uint16_t>(w >> 16)), o8[2 * e + 1]);
    }
  };
  int j = k0 + wave * KPW + sub;
  for (; j + 3 * STEP < k1; j += 4 * STEP) {
    u32x4 v4[4];
    float p4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { v4[u] = vrow(j + u * STEP); p4[u] = sc[j + u * STEP - k0]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) fold(v4[u], p4[u]);
  }
  for (; j < k1; j += STEP) fold(vrow(j), sc[j - k0]);
#pragma unroll
  for (int off = LPK; off < 64; off <<= 1)
#pragma unroll
    for (int e = 0; e < 8; ++e) o8[e] += __shfl_xor(o8[e], off, 64);
  if (sub == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) part[wave * HD + ch * 8 + e] = o8[e];
  }
  __syncthreads();
  if (tid < HD) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) t += part[w * HD + tid];
    if (S == 1) {
      out[static_cast<int64_t>(h) * HD + tid] = E::r(t / sum);
    } else {
      float* rec = ws + (static_cast<int64_t>(h) * S + sp) * (HD + 2);   // (an empty share leaves max = -inf, sum = 0, output 0)
      rec[2 + tid] = t;
      if (tid == 0) { rec[0] = mx; rec[1] = sum; }
    }
  }
}

// ---- the S shares of a head, merged in split order: out = sum_s e^(m_s - m) o_s / sum_s e^(m_s - m) l_s ----
template <bool BF>
__global__ __launch_bounds__(256) void attn_combine_kernel(const float* __restrict__ ws, uint16_t* __restrict__ out, int S, int HD) {
  using E = El<BF>;
  const int h = blockIdx.x, d = threadIdx.x;
  if (d >= HD) return;
  const float* rec = ws + static_cast<int64_t>(h) * S * (HD + 2);
  float m = -INFINITY;
  for (int s_ = 0; s_ < S; ++s_) m = fmaxf(m, rec[s_ * (HD + 2)]);
  float num = 0.f, den = 0.f;
  for (int s_ = 0; s_ < S; ++s_) {
    const float* r_ = rec + s_ * (HD + 2);
    const float w = r_[1] > 0.f ? __expf(r_[0] - m) : 0.f;
    num = fmaf(w, r_[2 + d], num);
    den = fmaf(w, r_[1], den);
  }
  out[static_cast<int64_t>(h) * HD + d] = E::r(num / den);
}

// ---- the per-token work either side of the decoder blocks (hqq/utils/generation_hf.py:405-540: the embedding lookup, the rotary table row and the causal mask of one
//      query in front; argmax, token hand-over and position increment behind), one launch each instead of nine small torch kernels.  Pure copies and compares:
//      bit-identical to the torch ops they replace.
//      token_prologue: h = embed[tok]; cos = cos_tab[pos]; sin = sin_tab[pos]; mask[i] = i <= pos ? 0 : -inf (mask == null: the caller's attention needs none).
//      A token / position outside the tables reads the last row (the torch ops would trap; the host checks positions, utils/generation.py) ----
__global__ __launch_bounds__(256) void token_prologue_kernel(const int64_t* __restrict__ tok, const int64_t* __restrict__ pos, const uint16_t* __restrict__ embed, int64_t vocab, int H,
                                                             const uint16_t* __restrict__ cos_tab, const uint16_t* __restrict__ sin_tab, int64_t L, int hd,
                                                             uint16_t* __restrict__ h, uint16_t* __restrict__ cos_o, uint16_t* __restrict__ sin_o, uint16_t* __restrict__ mask,
                                                             uint16_t zero_bits, uint16_t ninf_bits) {
  int64_t t = tok[0], p = pos[0];
  t = t < 0 ? 0 : (t >= vocab ? vocab - 1 : t);
  const int64_t pr = p < 0 ? 0 : (p >= L ? L - 1 : p);
  const int nb = static_cast<int>(gridDim.x), b = static_cast<int>(blockIdx.x), tid = static_cast<int>(threadIdx.x);
  const u32x4* src = reinterpret_cast<const u32x4*>(embed + t * H);
  for (int i = b * 256 + tid; i < H / 8; i += nb * 256) reinterpret_cast<u32x4*>(h)[i] = src[i];
  if (b == 0 && cos_tab) {
    for (int i = tid; i < hd; i += 256) { cos_o[i] = cos_tab[pr * hd + i]; sin_o[i] = sin_tab[pr * hd + i]; }
  }
  if (mask) {
    for (int64_t i = b * 256 + tid; i < L; i += static_cast<int64_t>(nb) * 256) mask[i] = i <= p ? zero_bits : ninf_bits;
  }
}

// argmax_advance: next = the FIRST index of the largest logit (torch.argmax's tie rule; logits finite), written to next_tok and tok, pos += 1.  One workgroup:
// 1024 threads keep (value, index) of their strided share in index order, then a fixed tree in LDS.
template <bool BF>
__global__ __launch_bounds__(1024) void argmax_advance_kernel(const uint16_t* __restrict__ logits, int n, int64_t* __restrict__ next_tok, int64_t* __restrict__ tok, int64_t* __restrict__ pos) {
  using E = El<BF>;
  __shared__ float bv[1024];
  __shared__ int bi[1024];
  const int tid = static_cast<int>(threadIdx.x);
  // torch.argmax's order: a NaN is the maximum (the FIRST NaN wins), otherwise the greatest value, the earliest index on a tie
  auto better = [](float v, int i, float bvv, int bii) {
    if (bii == 0x7fffffff) return true;
    if (v != v) return !(bvv != bvv) || i < bii;
    if (bvv != bvv) return false;
    return v > bvv || (v == bvv && i < bii);
  };
  float best = -INFINITY;
  int idx = 0x7fffffff;
  for (int i = tid; i < n; i += 1024) {
    const float v = E::f(logits[i]);
    if (better(v, i, best, idx)) { best = v; idx = i; }
  }
  bv[tid] = best; bi[tid] = idx;
  __syncthreads();
  for (int s_ = 512; s_ > 0; s_ >>= 1) {
    if (tid < s_) {
      const float v = bv[tid + s_];
      const int j = bi[tid + s_];
      if (j != 0x7fffffff && better(v, j, bv[tid], bi[tid])) { bv[tid] = v; bi[tid] = j; }
    }
    __syncthreads();
  }
  if (tid == 0) {
    const int64_t w = bi[0] == 0x7fffffff ? 0 : bi[0];
    next_tok[0] = w;
    if (tok) tok[0] = w;
    if (pos) pos[0] += 1;
  }
}

}  // namespace hqq

using namespace hqq;

static inline bool block_dtype_ok(int dtype, const char* who) {
  if (dtype == HQQ_F16 || dtype == HQQ_BF16) return true;
  set_error("%s: fp16 / bf16 only (dtype %d)", who, dtype);
  return false;
}
typedef const uint16_t* cu16;
typedef uint16_t* u16;

extern "C" {

int hqq_hip_add_rmsnorm(void* h, const void* delta, const void* weight, float eps, void* xn_out, int64_t rows, int64_t H, int dtype, void* stream) {
  clear_stale_error();
  if (!block_dtype_ok(dtype, "hqq_hip_add_rmsnorm")) return HQQ_ERR_UNSUPPORTED;
  if (!h || !weight || !xn_out || rows < 1 || H < 8 || H % 8 || rows > INT32_MAX || H > INT32_MAX) { set_error("hqq_hip_add_rmsnorm: bad arguments (H must be a multiple of 8)"); return HQQ_ERR_SHAPE; }
  if (!aligned16(h) || !aligned16(weight) || !aligned16(xn_out) || (delta && !aligned16(delta))) { set_error("hqq_hip_add_rmsnorm: pointers must be 16-byte aligned"); return HQQ_ERR_ALIGN; }
#define HQQ_NORM_GO(RPTV, BFV)                                                                                                                   \
  hipLaunchKernelGGL((add_rmsnorm_kernel<RPTV, BFV>), dim3(static_cast<unsigned>(rows)), dim3(512), 0, as_stream(stream), static_cast<u16>(h), static_cast<cu16>(delta), \
                     static_cast<cu16>(weight), eps, static_cast<u16>(xn_out), static_cast<int>(H))
  if (dtype == HQQ_BF16) {
    if (H <= 512 * 8) HQQ_NORM_GO(1, true);
    else if (H <= 512 * 8 * 2) HQQ_NORM_GO(2, true);
    else HQQ_NORM_GO(0, true);
  } else {
    if (H <= 512 * 8) HQQ_NORM_GO(1, false);
    else if (H <= 512 * 8 * 2) HQQ_NORM_GO(2, false);
    else HQQ_NORM_GO(0, false);
  }
#undef HQQ_NORM_GO
  return check_launch("hqq_hip_add_rmsnorm");
}

int hqq_hip_rope_cache(const void* q, const void* k, const void* v, const void* cos, const void* sin, const int64_t* pos_dev, void* q_out, void* k_cache,
                       void* v_cache, int64_t n_heads, int64_t n_kv_heads, int64_t head_dim, int64_t cache_len, int dtype, void* stream) {
  clear_stale_error();
  if (!block_dtype_ok(dtype, "hqq_hip_rope_cache")) return HQQ_ERR_UNSUPPORTED;
  if (!q || !k || !v || !cos || !sin || !pos_dev || !q_out || !k_cache || !v_cache || n_heads < 1 || n_kv_heads < 1 || head_dim < 2 || head_dim % 2 || cache_len < 1 ||
      (n_heads + n_kv_heads) * head_dim > INT32_MAX || cache_len > INT32_MAX) { set_error("hqq_hip_rope_cache: bad arguments"); return HQQ_ERR_SHAPE; }
  const int64_t total = (n_heads + n_kv_heads) * (head_dim / 2);
#define HQQ_ROPE_GO(BFV)                                                                                                                          \
  hipLaunchKernelGGL(rope_cache_kernel<BFV>, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, as_stream(stream), static_cast<cu16>(q), static_cast<cu16>(k), \
                     static_cast<cu16>(v), static_cast<cu16>(cos), static_cast<cu16>(sin), pos_dev, static_cast<u16>(q_out), static_cast<u16>(k_cache),  \
                     static_cast<u16>(v_cache), static_cast<int>(n_heads), static_cast<int>(n_kv_heads), static_cast<int>(head_dim), static_cast<int>(cache_len))
  if (dtype == HQQ_BF16) HQQ_ROPE_GO(true);
  else HQQ_ROPE_GO(false);
#undef HQQ_ROPE_GO
  return check_launch("hqq_hip_rope_cache");
}

int hqq_hip_silu_mul(const void* gate, const void* up, void* out, int64_t n, int dtype, void* stream) {
  clear_stale_error();
  if (!block_dtype_ok(dtype, "hqq_hip_silu_mul")) return HQQ_ERR_UNSUPPORTED;
  if (!gate || !up || !out || n < 8 || n % 8) { set_error("hqq_hip_silu_mul: bad arguments (n must be a multiple of 8)"); return HQQ_ERR_SHAPE; }
  if (!aligned16(gate) || !aligned16(up) || !aligned16(out)) { set_error("hqq_hip_silu_mul: pointers must be 16-byte aligned"); return HQQ_ERR_ALIGN; }
  const dim3 grid(static_cast<unsigned>((n / 8 + 255) / 256));
  if (dtype == HQQ_BF16) hipLaunchKernelGGL(silu_mul_kernel<true>, grid, dim3(256), 0, as_stream(stream), static_cast<cu16>(gate), static_cast<cu16>(up), static_cast<u16>(out), n);
  else hipLaunchKernelGGL(silu_mul_kernel<false>, grid, dim3(256), 0, as_stream(stream), static_cast<cu16>(gate), static_cast<cu16>(up), static_cast<u16>(out), n);
  return check_launch("hqq_hip_silu_mul");
}

static int attn_decode_run(const char* who, bool rope, const void* q, const void* k_raw, const void* v_raw, const void* cosv, const void* sinv, const int64_t* pos_dev,
                           void* k_cache, void* v_cache, void* out, int64_t n_heads, int64_t n_kv_heads, int64_t head_dim, int64_t cache_len, float scaling, int dtype,
                           int64_t splits, void* workspace, size_t workspace_bytes, void* stream) {
  clear_stale_error();
  if (!block_dtype_ok(dtype, who)) return HQQ_ERR_UNSUPPORTED;
  if (!q || !k_cache || !v_cache || !pos_dev || !out || (rope && (!k_raw || !v_raw || !cosv || !sinv)) || n_heads < 1 || n_kv_heads < 1 || n_heads % n_kv_heads ||
      cache_len < 1 || cache_len > 30000 || n_heads > INT32_MAX) {
    set_error("%s: bad arguments (cache_len <= 30000, n_heads a multiple of n_kv_heads)", who);
    return HQQ_ERR_SHAPE;
  }
  if (head_dim != 64 && head_dim != 128 && head_dim != 256) { set_error("%s: head_dim %lld not covered (64 / 128 / 256)", who, (long long)head_dim); return HQQ_ERR_UNSUPPORTED; }
  if (!aligned16(q) || !aligned16(k_cache) || !aligned16(v_cache) || !aligned16(out) || (rope && (!aligned16(k_raw) || !aligned16(v_raw)))) {
    set_error("%s: pointers must be 16-byte aligned", who);
    return HQQ_ERR_ALIGN;
  }
  const int HD = static_cast<int>(head_dim);
  if (splits < 1 || splits > 64) { set_error("%s: splits must be 1..64 (got %lld)", who, (long long)splits); return HQQ_ERR_SHAPE; }
  const int S = static_cast<int>(splits);
  if (S > 1 && (!workspace || workspace_bytes < static_cast<size_t>(n_heads) * S * (HD + 2) * sizeof(float))) {
    set_error("%s: %lld splits need a workspace of hqq_hip_attn_decode_workspace_bytes(...) bytes", who, (long long)splits);
    return HQQ_ERR_SHAPE;
  }
  float* wsf = static_cast<float*>(workspace);
  const int lds = 64 + HD * 6 + 8 * HD * 4 + static_cast<int>((cache_len + S - 1) / S + 8) * 4;
  const dim3 grid(static_cast<unsigned>(n_heads), static_cast<unsigned>(S)), block(512);
  static LdsRaised raised[12];
  constexpr int LDS_MAX = 64 + 256 * 6 + 8 * 256 * 4 + 30000 * 4;
#define HQQ_ATTN_GO(HDV, RP, BFV, IDX)                                                                                                     \
  do {                                                                                                                                     \
    if (lds > 48 * 1024)                                                                                                                   \
      if (const int rc = raise_lds_limit(raised[IDX], reinterpret_cast<const void*>(&attn_decode_kernel<HDV, RP, BFV>), LDS_MAX, who)) return rc; \
    hipLaunchKernelGGL((attn_decode_kernel<HDV, RP, BFV>), grid, block, lds, as_stream(stream), static_cast<cu16>(q), static_cast<cu16>(k_cache), \
                       static_cast<cu16>(v_cache), pos_dev, static_cast<u16>(out), static_cast<int>(n_heads), static_cast<int>(n_kv_heads),  \
                       static_cast<int>(cache_len), scaling, static_cast<cu16>(k_raw), static_cast<cu16>(v_raw),                            \
                       static_cast<cu16>(cosv), static_cast<cu16>(sinv), static_cast<u16>(k_cache), static_cast<u16>(v_cache), S, wsf);    \
  } while (0)
#define HQQ_ATTN_HD(RP, BFV, BASE)                                                                                                         \
  do {                                                                                                                                     \
    if (HD == 64) HQQ_ATTN_GO(64, RP, BFV, BASE);                                                                                          \
    else if (HD == 128) HQQ_ATTN_GO(128, RP, BFV, BASE + 1);                                                                               \
    else HQQ_ATTN_GO(256, RP, BFV, BASE + 2);                                                                                              \
  } while (0)
  const bool bf = dtype == HQQ_BF16;
  if (rope && bf) HQQ_ATTN_HD(true, true, 0);
  else if (rope) HQQ_ATTN_HD(true, false, 3);
  else if (bf) HQQ_ATTN_HD(false, true, 6);
  else HQQ_ATTN_HD(false, false, 9);
#undef HQQ_ATTN_HD
#undef HQQ_ATTN_GO
  if (S > 1) {
    if (bf) hipLaunchKernelGGL(attn_combine_kernel<true>, dim3(static_cast<unsigned>(n_heads)), dim3(256), 0, as_stream(stream), wsf, static_cast<u16>(out), S, HD);
    else hipLaunchKernelGGL(attn_combine_kernel<false>, dim3(static_cast<unsigned>(n_heads)), dim3(256), 0, as_stream(stream), wsf, static_cast<u16>(out), S, HD);
  }
  return check_launch(who);
}

size_t hqq_hip_attn_decode_workspace_bytes(int64_t n_heads, int64_t head_dim, int64_t splits) {
  return splits > 1 ? static_cast<size_t>(n_heads) * static_cast<size_t>(splits) * static_cast<size_t>(head_dim + 2) * sizeof(float) : 0;
}

int hqq_hip_attn_decode(const void* q, const void* k_cache, const void* v_cache, const int64_t* pos_dev, void* out, int64_t n_heads, int64_t n_kv_heads,
                        int64_t head_dim, int64_t cache_len, float scaling, int dtype, int64_t splits, void* workspace, size_t workspace_bytes, void* stream) {
  return attn_decode_run("hqq_hip_attn_decode", false, q, nullptr, nullptr, nullptr, nullptr, pos_dev, const_cast<void*>(k_cache), const_cast<void*>(v_cache), out,
                         n_heads, n_kv_heads, head_dim, cache_len, scaling, dtype, splits, workspace, workspace_bytes, stream);
}

int hqq_hip_rope_attn_decode(const void* q, const void* k, const void* v, const void* cos, const void* sin, const int64_t* pos_dev, void* k_cache, void* v_cache, void* out,
                             int64_t n_heads, int64_t n_kv_heads, int64_t head_dim, int64_t cache_len, float scaling, int dtype, int64_t splits, void* workspace,
                             size_t workspace_bytes, void* stream) {
  return attn_decode_run("hqq_hip_rope_attn_decode", true, q, k, v, cos, sin, pos_dev, k_cache, v_cache, out, n_heads, n_kv_heads, head_dim, cache_len, scaling, dtype,
                         splits, workspace, workspace_bytes, stream);
}

int hqq_hip_token_prologue(const int64_t* tok_dev, const int64_t* pos_dev, const void* embed, int64_t vocab, int64_t H, const void* cos_tab, const void* sin_tab, int64_t L,
                           int64_t head_dim, void* h, void* cos, void* sin, void* mask, int dtype, void* stream) {
  clear_stale_error();
  if (!block_dtype_ok(dtype, "hqq_hip_token_prologue")) return HQQ_ERR_UNSUPPORTED;
  if (!tok_dev || !pos_dev || !embed || !h || vocab < 1 || H < 8 || H % 8 || H > INT32_MAX || L < 1 || (cos_tab && (!sin_tab || !cos || !sin || head_dim < 1 || head_dim > INT32_MAX))) {
    set_error("hqq_hip_token_prologue: bad arguments (H a multiple of 8; cos / sin tables and outputs come together)");
    return HQQ_ERR_SHAPE;
  }
  if (!aligned16(embed) || !aligned16(h)) { set_error("hqq_hip_token_prologue: embed / h must be 16-byte aligned"); return HQQ_ERR_ALIGN; }
  const uint16_t ninf = dtype == HQQ_BF16 ? 0xFF80u : 0xFC00u;
  const int64_t work = (mask ? L : 0) > H / 8 ? (mask ? L : 0) : H / 8;
  const unsigned blocks = static_cast<unsigned>(work / 256 < 1 ? 1 : (work / 256 > 64 ? 64 : work / 256));
  hipLaunchKernelGGL(token_prologue_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), tok_dev, pos_dev, static_cast<cu16>(embed), vocab, static_cast<int>(H), static_cast<cu16>(cos_tab),
                     static_cast<cu16>(sin_tab), L, static_cast<int>(head_dim), static_cast<u16>(h), static_cast<u16>(cos), static_cast<u16>(sin), static_cast<u16>(mask), static_cast<uint16_t>(0), ninf);
  return check_launch("hqq_hip_token_prologue");
}

int hqq_hip_argmax_advance(const void* logits, int64_t n, int dtype, int64_t* next_tok_dev, int64_t* tok_dev, int64_t* pos_dev, void* stream) {
  clear_stale_error();
  if (!block_dtype_ok(dtype, "hqq_hip_argmax_advance")) return HQQ_ERR_UNSUPPORTED;
  if (!logits || !next_tok_dev || n < 1 || n > INT32_MAX - 1) { set_error("hqq_hip_argmax_advance: bad arguments"); return HQQ_ERR_SHAPE; }
  if (dtype == HQQ_BF16) hipLaunchKernelGGL(argmax_advance_kernel<true>, dim3(1), dim3(1024), 0, as_stream(stream), static_cast<cu16>(logits), static_cast<int>(n), next_tok_dev, tok_dev, pos_dev);
  else hipLaunchKernelGGL(argmax_advance_kernel<false>, dim3(1), dim3(1024), 0, as_stream(stream), static_cast<cu16>(logits), static_cast<int>(n), next_tok_dev, tok_dev, pos_dev);
  return check_launch("hqq_hip_argmax_advance");
}

}  // extern "C"
