// mixlo_probe.hip — does hipcc's v_fma_mixlo_f16 for `half(float(h) * r)` round ONCE (exact product -> fp16) where the two-instruction form rounds twice (fp32, then fp16)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void k(const _Float16* h, const float* r, _Float16* fused, _Float16* two, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float a = static_cast<float>(h[i]);
  fused[i] = static_cast<_Float16>(a * r[i]);            // the compiler's choice (v_fma_mixlo_f16 on gfx950)
  float p = a * r[i];
  asm volatile("" : "+v"(p));                            // the product as an fp32 VALUE: v_mul_f32, then v_cvt_f16_f32
  two[i] = static_cast<_Float16>(p);
}
int main() {
  const int n = 1 << 24;
  std::vector<_Float16> h(n); std::vector<float> r(n);
  uint64_t s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  for (int i = 0; i < n; ++i) { h[i] = static_cast<_Float16>(((rnd() >> 11) * (1.0 / 9007199254740992.0) - 0.5) * 8.0); r[i] = 0.5f + static_cast<float>((rnd() >> 40) * (1.0 / 16777216.0)); }
  _Float16 *dh, *df, *dt; float* dr;
  (void)hipMalloc(&dh, n * 2); (void)hipMalloc(&df, n * 2); (void)hipMalloc(&dt, n * 2); (void)hipMalloc(&dr, n * 4);
  (void)hipMemcpy(dh, h.data(), n * 2, hipMemcpyHostToDevice); (void)hipMemcpy(dr, r.data(), n * 4, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dh, dr, df, dt, n);
  std::vector<_Float16> f(n), t(n);
  (void)hipMemcpy(f.data(), df, n * 2, hipMemcpyDeviceToHost); (void)hipMemcpy(t.data(), dt, n * 2, hipMemcpyDeviceToHost);
  long diff = 0, fused_is_single = 0, two_is_double = 0;
  for (int i = 0; i < n; ++i) {
    const double exact = static_cast<double>(static_cast<float>(h[i])) * static_cast<double>(r[i]);
    const _Float16 single = static_cast<_Float16>(exact);                                   // one rounding of the exact product (double holds it exactly: 11 + 24 bits)
    const _Float16 dbl = static_cast<_Float16>(static_cast<float>(exact));                 // fp32 first, then fp16
    if (f[i] != t[i]) ++diff;
    if (f[i] == single) ++fused_is_single;
    if (t[i] == dbl) ++two_is_double;
  }
  printf("n %d: fused != two-step in %ld cases; fused == single-rounding %ld; two-step == double-rounding %ld\n", n, diff, fused_is_single, two_is_double);
  return 0;
}
