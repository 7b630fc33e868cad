// block.hip — the steps either side of the fused GEMVs in a decode step, gfx950, fp16.
//
// The reference's headline number is the tok/s of its generate loop (hqq/utils/generation_hf.py:117-540, Readme.md:153): HF's decoder
// block around HQQLinear.forward — RMSNorm, rotary embedding, KV-cache update, SiLU(gate) * up, the residual adds — is ~25 small
// eager kernels per block, 79 % of a bs = 1 token once the linears are fused.  Three kernels replace twenty of them;
// each restates the HF module's arithmetic rounding for rounding, so that the fused loop emits the same tokens:
//   add_rmsnorm   h += delta (fp16 add: `residual + hidden_states`), then LlamaRMSNorm: fp32 x * rsqrt(mean(x^2) + eps) -> fp16 -> weight * (fp16 mul)
//   rope_cache    apply_rotary_pos_emb: (q * cos) + (rotate_half(q) * sin) with three fp16 roundings, k likewise, and the StaticCache
//                 update (k_rot / v written at cache_position, read from device memory: graph-replay safe)
//   silu_mul      LlamaMLP: act_fn(gate) * up — silu in fp32 (x / (1 + exp(-x))), rounded to fp16, then the fp16 product
// Compiled with -ffp-contract=off: a fused multiply-add would remove a rounding HF's separate ops make.
#include "hqq_common.h"

namespace hqq {

// ---- h (+= delta), xn = weight * fp16(float(h) * rsqrt(mean(float(h)^2) + eps)): one workgroup of 256 threads per row ----
__global__ __launch_bounds__(256) void add_rmsnorm_kernel(half_t* __restrict__ h, const half_t* __restrict__ delta, const half_t* __restrict__ weight, float eps,
                                                          half_t* __restrict__ xn, int H) {
  __shared__ float part[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  half_t* hr = h + static_cast<int64_t>(blockIdx.x) * H;
  const half_t* dr = delta ? delta + static_cast<int64_t>(blockIdx.x) * H : nullptr;
  half_t* xr = xn + static_cast<int64_t>(blockIdx.x) * H;
  // 8 consecutive elements per thread and pass (16-byte accesses); H % 8 == 0
  float sum = 0.f;
  for (int i = tid * 8; i < H; i += 256 * 8) {
    u32x4 hv = *reinterpret_cast<const u32x4*>(hr + i);
    half_t* hp = reinterpret_cast<half_t*>(&hv);
    if (dr) {
      const u32x4 dv = *reinterpret_cast<const u32x4*>(dr + i);
      const half_t* dp = reinterpret_cast<const half_t*>(&dv);
#pragma unroll
      for (int j = 0; j < 8; ++j) hp[j] = hp[j] + dp[j];   // residual + hidden_states, one fp16 rounding
      *reinterpret_cast<u32x4*>(hr + i) = hv;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float f = static_cast<float>(hp[j]); sum += f * f; }
  }
  // wave sum (DPP-free: shuffles), then the four waves through LDS, fixed order
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
  if (lane == 0) part[wave] = sum;
  __syncthreads();
  const float total = (part[0] + part[1]) + (part[2] + part[3]);
  const float r = rsqrtf(total / static_cast<float>(H) + eps);
  for (int i = tid * 8; i < H; i += 256 * 8) {
    const u32x4 hv = *reinterpret_cast<const u32x4*>(hr + i);
    const u32x4 wv = *reinterpret_cast<const u32x4*>(weight + i);
    const half_t* hp = reinterpret_cast<const half_t*>(&hv);
    const half_t* wp = reinterpret_cast<const half_t*>(&wv);
    u32x4 ov;
    half_t* op = reinterpret_cast<half_t*>(&ov);
#pragma unroll
    for (int j = 0; j < 8; ++j) op[j] = wp[j] * static_cast<half_t>(static_cast<float>(hp[j]) * r);
    *reinterpret_cast<u32x4*>(xr + i) = ov;
  }
}

// ---- rotary embedding of q and k, KV-cache write.  One thread per (head, i < hd / 2): elements i and i + hd / 2 of a head ----
__global__ __launch_bounds__(256) void rope_cache_kernel(const half_t* __restrict__ q, const half_t* __restrict__ k, const half_t* __restrict__ v,
                                                         const half_t* __restrict__ cosv, const half_t* __restrict__ sinv, const int64_t* __restrict__ pos,
                                                         half_t* __restrict__ q_out, half_t* __restrict__ k_cache, half_t* __restrict__ v_cache,
                                                         int n_heads, int n_kv, int hd, int cache_len) {
  const int half = hd / 2;
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = (n_heads + n_kv) * half;
  if (id >= total) return;
  const int head = id / half, i = id - head * half;
  const bool is_k = head >= n_heads;
  const half_t* src = is_k ? k + static_cast<int64_t>(head - n_heads) * hd : q + static_cast<int64_t>(head) * hd;
  const half_t x1 = src[i], x2 = src[i + half];
  const half_t c1 = cosv[i], c2 = cosv[i + half], s1 = sinv[i], s2 = sinv[i + half];
  // q_embed = (q * cos) + (rotate_half(q) * sin), rotate_half = cat(-x2, x1): every product and the sum round to fp16
  const half_t o1 = (x1 * c1) + ((-x2) * s1);
  const half_t o2 = (x2 * c2) + (x1 * s2);
  if (!is_k) {
    q_out[static_cast<int64_t>(head) * hd + i] = o1;
    q_out[static_cast<int64_t>(head) * hd + i + half] = o2;
  } else {
    const int64_t p = pos[0];
    const int kh = head - n_heads;
    half_t* kd = k_cache + (static_cast<int64_t>(kh) * cache_len + p) * hd;
    half_t* vd = v_cache + (static_cast<int64_t>(kh) * cache_len + p) * hd;
    kd[i] = o1;
    kd[i + half] = o2;
    const half_t* vs = v + static_cast<int64_t>(kh) * hd;
    vd[i] = vs[i];
    vd[i + half] = vs[i + half];
  }
}

// ---- out = fp16(silu(gate)) * up ----
__global__ __launch_bounds__(256) void silu_mul_kernel(const half_t* __restrict__ g, const half_t* __restrict__ u, half_t* __restrict__ out, int64_t n) {
  const int64_t i = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) * 8;
  if (i >= n) return;
  const u32x4 gv = *reinterpret_cast<const u32x4*>(g + i);
  const u32x4 uv = *reinterpret_cast<const u32x4*>(u + i);
  const half_t* gp = reinterpret_cast<const half_t*>(&gv);
  const half_t* up = reinterpret_cast<const half_t*>(&uv);
  u32x4 ov;
  half_t* op = reinterpret_cast<half_t*>(&ov);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float x = static_cast<float>(gp[j]);
    const half_t s = static_cast<half_t>(x / (1.0f + expf(-x)));
    op[j] = s * up[j];
  }
  *reinterpret_cast<u32x4*>(out + i) = ov;
}

}  // namespace hqq

using namespace hqq;

extern "C" {

int hqq_hip_add_rmsnorm(void* h, const void* delta, const void* weight, float eps, void* xn_out, int64_t rows, int64_t H, int dtype, void* stream) {
  clear_stale_error();
  if (dtype != HQQ_F16) { set_error("hqq_hip_add_rmsnorm: fp16 only (dtype %d)", dtype); return HQQ_ERR_UNSUPPORTED; }
  if (!h || !weight || !xn_out || rows < 1 || H < 8 || H % 8 || rows > INT32_MAX || H > INT32_MAX) { set_error("hqq_hip_add_rmsnorm: bad arguments (H must be a multiple of 8)"); return HQQ_ERR_SHAPE; }
  if (!aligned16(h) || !aligned16(weight) || !aligned16(xn_out) || (delta && !aligned16(delta))) { set_error("hqq_hip_add_rmsnorm: pointers must be 16-byte aligned"); return HQQ_ERR_ALIGN; }
  hipLaunchKernelGGL(add_rmsnorm_kernel, dim3(static_cast<unsigned>(rows)), dim3(256), 0, as_stream(stream), static_cast<half_t*>(h), static_cast<const half_t*>(delta),
                     static_cast<const half_t*>(weight), eps, static_cast<half_t*>(xn_out), static_cast<int>(H));
  return check_launch("hqq_hip_add_rmsnorm");
}

int hqq_hip_rope_cache(const void* q, const void* k, const void* v, const void* cos, const void* sin, const int64_t* pos_dev, void* q_out, void* k_cache,
                       void* v_cache, int64_t n_heads, int64_t n_kv_heads, int64_t head_dim, int64_t cache_len, int dtype, void* stream) {
  clear_stale_error();
  if (dtype != HQQ_F16) { set_error("hqq_hip_rope_cache: fp16 only (dtype %d)", dtype); return HQQ_ERR_UNSUPPORTED; }
  if (!q || !k || !v || !cos || !sin || !pos_dev || !q_out || !k_cache || !v_cache || n_heads < 1 || n_kv_heads < 1 || head_dim < 2 || head_dim % 2 || cache_len < 1 ||
      (n_heads + n_kv_heads) * head_dim > INT32_MAX || cache_len > INT32_MAX) { set_error("hqq_hip_rope_cache: bad arguments"); return HQQ_ERR_SHAPE; }
  const int64_t total = (n_heads + n_kv_heads) * (head_dim / 2);
  hipLaunchKernelGGL(rope_cache_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, as_stream(stream), static_cast<const half_t*>(q), static_cast<const half_t*>(k),
                     static_cast<const half_t*>(v), static_cast<const half_t*>(cos), static_cast<const half_t*>(sin), pos_dev, static_cast<half_t*>(q_out),
                     static_cast<half_t*>(k_cache), static_cast<half_t*>(v_cache), static_cast<int>(n_heads), static_cast<int>(n_kv_heads), static_cast<int>(head_dim),
                     static_cast<int>(cache_len));
  return check_launch("hqq_hip_rope_cache");
}

int hqq_hip_silu_mul(const void* gate, const void* up, void* out, int64_t n, int dtype, void* stream) {
  clear_stale_error();
  if (dtype != HQQ_F16) { set_error("hqq_hip_silu_mul: fp16 only (dtype %d)", dtype); return HQQ_ERR_UNSUPPORTED; }
  if (!gate || !up || !out || n < 8 || n % 8) { set_error("hqq_hip_silu_mul: bad arguments (n must be a multiple of 8)"); return HQQ_ERR_SHAPE; }
  if (!aligned16(gate) || !aligned16(up) || !aligned16(out)) { set_error("hqq_hip_silu_mul: pointers must be 16-byte aligned"); return HQQ_ERR_ALIGN; }
  hipLaunchKernelGGL(silu_mul_kernel, dim3(static_cast<unsigned>((n / 8 + 255) / 256)), dim3(256), 0, as_stream(stream), static_cast<const half_t*>(gate),
                     static_cast<const half_t*>(up), static_cast<half_t*>(out), n);
  return check_launch("hqq_hip_silu_mul");
}

}  // extern "C"
