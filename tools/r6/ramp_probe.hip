// ramp_probe.hip — how long the hardware takes to START a grid's waves (first wave start -> last wave start), by workgroups, waves per workgroup,
// registers per wave and LDS per workgroup.  Dependent launches (a predecessor kernel runs in front, as in the decode stack).   hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
template <int VG>
__global__ void k(unsigned long long* ts, int spin) {
  extern __shared__ float lds[];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (VG >= 64) asm volatile("v_mov_b32 v63, 0" ::: "v63");
  if (VG >= 96) asm volatile("v_mov_b32 v95, 0" ::: "v95");
  if (VG >= 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
  // live for a while, like a streaming wave (so that residency, not only dispatch, is exercised)
  unsigned long long t1 = t0;
  while (t1 - t0 < (unsigned long long)spin) t1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) { const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); ts[2 * w] = t0; ts[2 * w + 1] = t1; }
  if (spin < 0) lds[threadIdx.x] = 1.f;
}
__global__ void pred(float* p) { p[blockIdx.x * blockDim.x + threadIdx.x] = 1.f; }
// a predecessor like a decode launch: 1024 workgroups x 256 threads stream `n16` 16-byte vectors (non-temporal) and store 2 bytes per wave
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
__global__ void pred_stream(const u4* src, unsigned short* dst, long n16) {
  const long i0 = (long)blockIdx.x * 256 + threadIdx.x, stride = (long)gridDim.x * 256;
  unsigned acc = 0;
  for (long i = i0; i < n16; i += stride) { const u4 v = __builtin_nontemporal_load(src + i); acc += v.x ^ v.y ^ v.z ^ v.w; }
  for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) dst[blockIdx.x * 4 + (threadIdx.x >> 6)] = (unsigned short)acc;
}
struct Big { const void* p[20]; int v[16]; };
template <int VG>
__global__ void kbig(unsigned long long* ts, int spin, Big b) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (VG >= 96) asm volatile("v_mov_b32 v95, 0" ::: "v95");
  unsigned long long t1 = t0;
  while (t1 - t0 < (unsigned long long)spin) t1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) { const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); ts[2 * w] = t0; ts[2 * w + 1] = t1 + (b.v[blockIdx.x & 15] & 0); }
}
static const u4* g_src; static unsigned short* g_dst;
static int g_mode = 0;
template <int VG> static void run(int wgs, int threads, int lds, int spin, unsigned long long* dts, float* dp) {
  std::vector<unsigned long long> h(2 * wgs * (threads / 64));
  double spread = 0, total = 0; const int reps = 5;
  for (int r = 0; r < reps + 1; ++r) {
    (void)hipMemset(dts, 0, h.size() * 8);
    if (g_mode == 0) pred<<<1024, 256>>>(dp); else pred_stream<<<1024, 256>>>(g_src, g_dst, (28l << 20) / 16);
    if (g_mode == 2) { Big b{}; hipLaunchKernelGGL(kbig<96>, dim3(wgs), dim3(threads), lds, 0, dts, spin, b); }
    else hipLaunchKernelGGL(k<VG>, dim3(wgs), dim3(threads), lds, 0, dts, spin);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h.data(), dts, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long a = ~0ull, b = 0, e = 0;
    for (size_t i = 0; i < h.size(); i += 2) { a = std::min(a, h[i]); b = std::max(b, h[i]); e = std::max(e, h[i + 1]); }
    if (r) { spread += (b - a) / 100.0; total += (e - a) / 100.0; }
  }
  printf("VGPR>=%3d  %4d workgroups x %d waves  LDS %6d B  wave life %4.1f us:  first->last wave START %5.2f us   first start -> last exit %5.2f us\n", VG, wgs, threads / 64, lds, spin / 100.0, spread / reps, total / reps);
}
int main() {
  unsigned long long* dts; float* dp;
  (void)hipMalloc(&dts, 1 << 22); (void)hipMalloc(&dp, 1024 * 256 * 4);
  (void)hipFuncSetAttribute((const void*)k<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  (void)hipFuncSetAttribute((const void*)k<96>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  { u4* sp; (void)hipMalloc(&sp, 64 << 20); (void)hipMemset(sp, 1, 64 << 20); g_src = sp; (void)hipMalloc(&g_dst, 1 << 16); }
  for (int mode : {1, 2}) {
    g_mode = mode;
    printf("-- predecessor: a 28 MB streaming launch of 4096 waves%s\n", mode == 2 ? "; 224-byte kernel argument block" : "");
    run<96>(1024, 256, 8704, 300, dts, dp);
    run<96>(256, 512, 8704, 300, dts, dp);
    run<96>(1024, 256, 8704, 0, dts, dp);
  }
  g_mode = 0;
  printf("-- predecessor: a trivial launch\n");
  for (int spin : {0, 300}) {
    run<32>(1024, 256, 0, spin, dts, dp);
    run<32>(1024, 256, 8704, spin, dts, dp);
    run<32>(1024, 256, 33 * 1024, spin, dts, dp);
    run<96>(1024, 256, 8704, spin, dts, dp);
    run<128>(1024, 256, 8704, spin, dts, dp);
    run<96>(512, 512, 8704, spin, dts, dp);
    run<96>(256, 1024, 8704, spin, dts, dp);
    run<96>(256, 512, 8704, spin, dts, dp);
    run<96>(256, 512, 23 * 1024, spin, dts, dp);
    run<96>(512, 256, 8704, spin, dts, dp);
    run<96>(2048, 256, 8704, spin, dts, dp);
    run<96>(1536, 256, 8704, spin, dts, dp);
  }
  return 0;
}
