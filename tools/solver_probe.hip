// solver_probe.hip — lab only: one proximal iteration of quantize.hip's solve_kernel with every intermediate written out, so that
// the stage at which the device leaves the CPU oracle's float32 sequence can be named (tools/solver_probe.py compares on the box).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -shared tools/solver_probe.hip -o tools/libsolver_probe.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ float sgnf(float e) { return static_cast<float>((e > 0.f) - (e < 0.f)); }

// one thread per group (no lane tricks: this probes the arithmetic, not the reduction order)
extern "C" __global__ void probe_kernel(const float* W, int64_t R, int gs, float maxv, int round_zero, float inv_beta, double pexp,
                                        float* sc_o, float* ze_o, float* q_o, float* wr_o, float* e_o, float* pw_o, float* pwf_o, float* t_o,
                                        float* we_o, float* t3_o) {
  const int64_t r = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const float* w = W + r * gs;
  float mn = w[0], mx = w[0];
  for (int c = 1; c < gs; ++c) { mn = fminf(mn, w[c]); mx = fmaxf(mx, w[c]); }
  const float denom = mx - mn;
  float sc = (1.0f / denom) * maxv;
  if (fabsf(denom) <= 1e-4f) sc = 1.0f;
  sc = fminf(sc, 2e4f);
  float ze = (-mn) * sc;
  if (round_zero) ze = rintf(ze);
  sc_o[r] = sc; ze_o[r] = ze;
  for (int c = 0; c < gs; ++c) {
    const float wf = w[c];
    float q = wf * sc;
    q = q + ze;
    q = rintf(q);
    q = fminf(fmaxf(q, 0.f), maxv);
    const float wr = (q - ze) / sc;
    const float e = wf - wr;
    const float a = fabsf(e);
    const float pw = static_cast<float>(pow(static_cast<double>(a), pexp));
    const float pwf = powf(a, static_cast<float>(pexp));
    float t = inv_beta * pw;
    t = a - t;
    t = (t < 0.f) ? 0.f : t;
    const float we = t * sgnf(e);
    float u = wf - we;
    u = u * sc;
    const int64_t i = r * gs + c;
    q_o[i] = q; wr_o[i] = wr; e_o[i] = e; pw_o[i] = pw; pwf_o[i] = pwf; t_o[i] = t; we_o[i] = we; t3_o[i] = q - u;
  }
}

extern "C" int probe_run(const float* W, int64_t R, int gs, float maxv, int round_zero, float inv_beta, double pexp,
                         float* sc_o, float* ze_o, float* q_o, float* wr_o, float* e_o, float* pw_o, float* pwf_o, float* t_o, float* we_o, float* t3_o,
                         void* stream) {
  hipLaunchKernelGGL(probe_kernel, dim3((R + 63) / 64), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), W, R, gs, maxv, round_zero, inv_beta, pexp,
                     sc_o, ze_o, q_o, wr_o, e_o, pw_o, pwf_o, t_o, we_o, t3_o);
  return static_cast<int>(hipGetLastError());
}
