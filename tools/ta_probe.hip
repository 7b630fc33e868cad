// ta_probe.hip — what one wave load instruction costs the CU's address/data path (development aid).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ta_probe.hip -o tools/lab_ta.bin
// 16 waves per CU (1024 workgroups x 256 threads) each issue NL loads of one kind from an L2-resident buffer; reported:
// cycles until the last load has been *issued* and until the data is back, per wave, averaged; divided by 16 NL = per
// instruction per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int NL = 8;

template <int KIND> __global__ __launch_bounds__(256) void k(const uint8_t* src, float* dst, unsigned long long* ts, int pred) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t wbase = ((size_t)(blockIdx.x * 4 + wave) * NL) * 1024 % (8u << 20);
  const unsigned long long t0 = __builtin_readcyclecounter();
  float f = 0.f;
  if constexpr (KIND == 0) {          // dwordx4, distinct 16 B per lane (1 KiB per instruction)
    u32x4 v[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) v[l] = *reinterpret_cast<const u32x4*>(src + wbase + l * 1024 + lane * 16);
    const unsigned long long t1 = __builtin_readcyclecounter();
#pragma unroll
    for (int l = 0; l < NL; ++l) f += __uint_as_float(v[l].x ^ v[l].y ^ v[l].z ^ v[l].w);
    const unsigned long long t2 = __builtin_readcyclecounter();
    if (lane == 0) { ts[(blockIdx.x * 4 + wave) * 2] = t1 - t0; ts[(blockIdx.x * 4 + wave) * 2 + 1] = t2 - t0; }
  } else if constexpr (KIND == 1) {   // dwordx4, every lane the same 16 B
    u32x4 v[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) v[l] = *reinterpret_cast<const u32x4*>(src + wbase + l * 1024 + (lane & pred) * 16);
    const unsigned long long t1 = __builtin_readcyclecounter();
#pragma unroll
    for (int l = 0; l < NL; ++l) f += __uint_as_float(v[l].x ^ v[l].y ^ v[l].z ^ v[l].w);
    const unsigned long long t2 = __builtin_readcyclecounter();
    if (lane == 0) { ts[(blockIdx.x * 4 + wave) * 2] = t1 - t0; ts[(blockIdx.x * 4 + wave) * 2 + 1] = t2 - t0; }
  } else if constexpr (KIND == 2) {   // ushort, contiguous (128 B per instruction)
    uint16_t v[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) v[l] = *reinterpret_cast<const uint16_t*>(src + wbase + l * 1024 + lane * 2);
    const unsigned long long t1 = __builtin_readcyclecounter();
#pragma unroll
    for (int l = 0; l < NL; ++l) f += (float)v[l];
    const unsigned long long t2 = __builtin_readcyclecounter();
    if (lane == 0) { ts[(blockIdx.x * 4 + wave) * 2] = t1 - t0; ts[(blockIdx.x * 4 + wave) * 2 + 1] = t2 - t0; }
  } else if constexpr (KIND == 3) {   // dword, contiguous (256 B per instruction)
    uint32_t v[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) v[l] = *reinterpret_cast<const uint32_t*>(src + wbase + l * 1024 + lane * 4);
    const unsigned long long t1 = __builtin_readcyclecounter();
#pragma unroll
    for (int l = 0; l < NL; ++l) f += __uint_as_float(v[l]);
    const unsigned long long t2 = __builtin_readcyclecounter();
    if (lane == 0) { ts[(blockIdx.x * 4 + wave) * 2] = t1 - t0; ts[(blockIdx.x * 4 + wave) * 2 + 1] = t2 - t0; }
  } else if constexpr (KIND == 4) {   // dwordx4 with EXEC = 0 lanes (predicate false everywhere, unknown to the compiler)
    u32x4 v[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) v[l] = u32x4{0u, 0u, 0u, 0u};
    const uint8_t* p = src + wbase + lane * 16;
    asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b64 exec, 0\n"
                 "global_load_dwordx4 %0, %8, off\n global_load_dwordx4 %1, %8, off offset:1024\n"
                 "global_load_dwordx4 %2, %8, off offset:2048\n global_load_dwordx4 %3, %8, off offset:3072\n"
                 "global_load_dwordx4 %4, %8, off\n global_load_dwordx4 %5, %8, off offset:1024\n"
                 "global_load_dwordx4 %6, %8, off offset:2048\n global_load_dwordx4 %7, %8, off offset:3072\n"
                 "s_mov_b64 exec, s[20:21]\n"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "v"(p) : "s20", "s21", "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int l = 0; l < NL; ++l) f += __uint_as_float(v[l].x ^ v[l].y ^ v[l].z ^ v[l].w);
    const unsigned long long t2 = __builtin_readcyclecounter();
    if (lane == 0) { ts[(blockIdx.x * 4 + wave) * 2] = t1 - t0; ts[(blockIdx.x * 4 + wave) * 2 + 1] = t2 - t0; }
  }
  if (f == 123.456f) dst[threadIdx.x] = f;
}

template <int KIND> static void run(const char* what, const uint8_t* src, float* dst, unsigned long long* ts) {
  const int wgs = 1024;
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<KIND>, dim3(wgs), dim3(256), 0, 0, src, dst, ts, 0);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(wgs * 4 * 2);
  hipMemcpy(h.data(), ts, h.size() * 8, hipMemcpyDeviceToHost);
  double a = 0, b = 0, amax = 0, bmax = 0;
  for (int w = 0; w < wgs * 4; ++w) { a += h[w * 2]; b += h[w * 2 + 1]; if (h[w*2] > amax) amax = h[w*2]; if (h[w*2+1] > bmax) bmax = h[w*2+1]; }
  a /= wgs * 4; b /= wgs * 4;
  printf("%-34s issued after %7.0f (max %7.0f) cycles, data after %7.0f (max %7.0f); max/(16 waves x %d) = %.1f cycles per instruction per CU\n", what, a, amax, b, bmax, NL, bmax / (16.0 * NL));
}

int main() {
  uint8_t* src; float* dst; unsigned long long* ts;
  hipMalloc(&src, 16u << 20); hipMemset(src, 1, 16u << 20); hipMalloc(&dst, 4096); hipMalloc(&ts, 1024 * 4 * 2 * 8);
  run<0>("dwordx4 distinct (1 KiB)", src, dst, ts);
  run<1>("dwordx4 same 16 B in every lane", src, dst, ts);
  run<2>("ushort contiguous (128 B)", src, dst, ts);
  run<3>("dword contiguous (256 B)", src, dst, ts);
  run<4>("dwordx4 with EXEC = 0", src, dst, ts);
  return 0;
}
