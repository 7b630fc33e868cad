// launch_probe.hip — fixed cost of one dependent launch inside a replayed hipGraph (development aid).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-kernarg-preload-count=16 tools/launch_probe.hip -o tools/lab_launch.bin
// Variants: empty kernel; kernarg -> one 16 B/lane load -> wave sum -> store, with the pointers inside a by-value struct
// (fetched with s_load) or as leading scalar arguments (preloaded into SGPRs by the command processor).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct S { const u32x4* src; float* dst; int n; int pad[9]; };

__global__ __launch_bounds__(256) void k_empty(S s) {}
__device__ __forceinline__ void body(const u32x4* src, float* dst, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const u32x4 v = __builtin_nontemporal_load(src + (i < n ? i : 0));
  float f = __uint_as_float(v.x) + __uint_as_float(v.y) + __uint_as_float(v.z) + __uint_as_float(v.w);
  for (int o = 32; o; o >>= 1) f += __shfl_xor(f, o);
  if ((threadIdx.x & 63) == 0) dst[i >> 6] = f;
}
__global__ __launch_bounds__(256) void k_struct(S s) { body(s.src, s.dst, s.n); }
__global__ __launch_bounds__(256) void k_scalar(const u32x4* src, float* dst, int n) { body(src, dst, n); }

// NL independent 16 B/lane loads per lane, all issued before the first use (what a streaming kernel with no dependent staging costs)
template <int NL> __global__ __launch_bounds__(256) void k_multi(const u32x4* src, float* dst, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  u32x4 v[NL];
#pragma unroll
  for (int l = 0; l < NL; ++l) v[l] = __builtin_nontemporal_load(src + (size_t)l * n + i);
  float f = 0.f;
#pragma unroll
  for (int l = 0; l < NL; ++l) f += __uint_as_float(v[l].x) + __uint_as_float(v[l].y) + __uint_as_float(v[l].z) + __uint_as_float(v[l].w);
  for (int o = 32; o; o >>= 1) f += __shfl_xor(f, o);
  if ((threadIdx.x & 63) == 0) dst[i >> 6] = f;
}

// the same streaming kernel with the decode kernel's footprint: dynamic LDS and/or ~90 live VGPRs
template <int NL, bool FAT> __global__ __launch_bounds__(256) void k_foot(const u32x4* src, float* dst, int n, int touch_lds) {
  extern __shared__ float lds[];
  const int i = blockIdx.x * 256 + threadIdx.x;
  u32x4 v[NL];
#pragma unroll
  for (int l = 0; l < NL; ++l) v[l] = __builtin_nontemporal_load(src + (size_t)l * n + i);
  float pad[64];
  if (FAT) {
#pragma unroll
    for (int k = 0; k < 64; ++k) { pad[k] = (float)(threadIdx.x + k); asm volatile("" : "+v"(pad[k])); }
  }
  if (touch_lds) { lds[threadIdx.x] = (float)i; __syncthreads(); }
  float f = touch_lds ? lds[threadIdx.x ^ 1] : 0.f;
#pragma unroll
  for (int l = 0; l < NL; ++l) f += __uint_as_float(v[l].x) + __uint_as_float(v[l].y) + __uint_as_float(v[l].z) + __uint_as_float(v[l].w);
  if (FAT) {
#pragma unroll
    for (int k = 0; k < 64; ++k) { asm volatile("" : "+v"(pad[k])); f += pad[k]; }
  }
  for (int o = 32; o; o >>= 1) f += __shfl_xor(f, o);
  if ((threadIdx.x & 63) == 0) dst[i >> 6] = f;
}

template <class F> static double time_graph(F launch, int chain, int reps) {
  hipStream_t st; hipStreamCreate(&st);
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < chain; ++i) launch(st, i);
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraphLaunch(ge, st); hipStreamSynchronize(st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, st);
  for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, st);
  hipEventRecord(e1, st); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3 / (reps * chain);
}

int main() {
  const int chain = 256, reps = 20;
  for (int wgs : {256, 1024, 4096}) {
    const int n = wgs * 256;
    const size_t pool = 64;   // distinct source buffers so that loads miss L2 / MALL (64 x 4..64 MiB)
    u32x4* src; float* dst;
    hipMalloc(&src, pool * n * 16); hipMemset(src, 1, pool * n * 16); hipMalloc(&dst, n / 64 * 4);
    double te = time_graph([&](hipStream_t st, int i) { S s{src, dst, n}; hipLaunchKernelGGL(k_empty, dim3(wgs), dim3(256), 0, st, s); }, chain, reps);
    double ts = time_graph([&](hipStream_t st, int i) { S s{src + (size_t)(i % pool) * n, dst, n}; hipLaunchKernelGGL(k_struct, dim3(wgs), dim3(256), 0, st, s); }, chain, reps);
    double tp = time_graph([&](hipStream_t st, int i) { hipLaunchKernelGGL(k_scalar, dim3(wgs), dim3(256), 0, st, (const u32x4*)(src + (size_t)(i % pool) * n), dst, n); }, chain, reps);
    printf("%5d WGs (%6.1f MiB/launch): empty %.2f us   struct-kernarg %.2f us   preloaded-kernarg %.2f us\n", wgs, n * 16.0 / (1 << 20), te, ts, tp);
    hipFree(src); hipFree(dst);
  }
  {
    const int wgs = 1024, n = wgs * 256;
    u32x4* src; float* dst;
    const size_t pool = 16;
    hipMalloc(&src, pool * 8 * n * 16); hipMemset(src, 1, pool * 8 * n * 16); hipMalloc(&dst, n / 64 * 4);
    double t2 = time_graph([&](hipStream_t st, int i) { hipLaunchKernelGGL(k_multi<2>, dim3(wgs), dim3(256), 0, st, (const u32x4*)(src + (size_t)(i % pool) * 8 * n), dst, n); }, chain, reps);
    double t4 = time_graph([&](hipStream_t st, int i) { hipLaunchKernelGGL(k_multi<4>, dim3(wgs), dim3(256), 0, st, (const u32x4*)(src + (size_t)(i % pool) * 8 * n), dst, n); }, chain, reps);
    double t8 = time_graph([&](hipStream_t st, int i) { hipLaunchKernelGGL(k_multi<8>, dim3(wgs), dim3(256), 0, st, (const u32x4*)(src + (size_t)(i % pool) * 8 * n), dst, n); }, chain, reps);
    printf("1024 WGs, all loads up front: 8 MiB %.2f us   16 MiB %.2f us   32 MiB %.2f us\n", t2, t4, t8);
    double f0 = time_graph([&](hipStream_t st, int i) { hipLaunchKernelGGL((k_foot<2, false>), dim3(wgs), dim3(256), 0, st, (const u32x4*)(src + (size_t)(i % pool) * 8 * n), dst, n, 0); }, chain, reps);
    double f1 = time_graph([&](hipStream_t st, int i) { hipLaunchKernelGGL((k_foot<2, false>), dim3(wgs), dim3(256), 8704, st, (const u32x4*)(src + (size_t)(i % pool) * 8 * n), dst, n, 1); }, chain, reps);
    double f2 = time_graph([&](hipStream_t st, int i) { hipLaunchKernelGGL((k_foot<2, true>), dim3(wgs), dim3(256), 0, st, (const u32x4*)(src + (size_t)(i % pool) * 8 * n), dst, n, 0); }, chain, reps);
    double f3 = time_graph([&](hipStream_t st, int i) { hipLaunchKernelGGL((k_foot<2, true>), dim3(wgs), dim3(256), 8704, st, (const u32x4*)(src + (size_t)(i % pool) * 8 * n), dst, n, 1); }, chain, reps);
    printf("8 MiB with footprint: plain %.2f us   +8.5 KiB LDS & barrier %.2f us   +64 live VGPRs %.2f us   both %.2f us\n", f0, f1, f2, f3);
  }
  return 0;
}
