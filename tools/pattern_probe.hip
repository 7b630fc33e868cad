// pattern_probe.hip — HBM bandwidth of the two ways a wave can read a [rows, K]-byte matrix (development aid):
//   A: one row per wave-load (64 lanes x 16 B = 1 KiB contiguous)             — gemv.hip
//   B: 16 rows x 64 B per wave-load (lane (r, c): row r, bytes 16 c)          — the MFMA A-operand layout of skinny.hip
//   C: as B but a lane fetches 64 contiguous bytes of its row with 4 loads (16 rows x 256 B per group of 4 loads)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pattern_probe.hip -o tools/lab_pattern.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int K = 4096;            // bytes per row (int4, K = 4096)
constexpr int UN = 8;              // loads in flight per wave

template <int PAT> __global__ __launch_bounds__(256) void k(const uint8_t* base, float* dst, int rows) {
  const int lane = threadIdx.x & 63;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
  uint32_t acc = 0;
  if (PAT == 0) {
    for (int row = gw; row < rows; row += nw) {            // 4 KiB per row: 4 loads
      u32x4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base + (size_t)row * K + j * 1024 + lane * 16));
#pragma unroll
      for (int j = 0; j < 4; ++j) acc += v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    }
  } else {
    const int r = lane & 15, c = lane >> 4;
    for (int tile = gw; tile * 16 < rows; tile += nw) {    // 16 rows x 4 KiB = 64 KiB per tile: 64 loads, UN at a time
      const uint8_t* p = base + (size_t)(tile * 16 + r) * K;
      for (int b = 0; b < 64; b += UN) {
        u32x4 v[UN];
#pragma unroll
        for (int j = 0; j < UN; ++j) {
          const int blk = b + j;
          const int off = PAT == 1 ? blk * 64 + c * 16 : (blk >> 2) * 256 + c * 64 + (blk & 3) * 16;
          v[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + off));
        }
#pragma unroll
        for (int j = 0; j < UN; ++j) acc += v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
      }
    }
  }
  if (acc == 0x12345678u) dst[threadIdx.x] = 1.f;
}

template <int PAT> static void run(const char* what, const uint8_t* base, float* dst, int rows, int wgs) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<PAT>, dim3(wgs), dim3(256), 0, 0, base, dst, rows);
  hipEventRecord(e0, 0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<PAT>, dim3(wgs), dim3(256), 0, 0, base, dst, rows);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s %4d WGs: %7.1f GB/s\n", what, wgs, (double)rows * K * 5 / (ms * 1e-3) / 1e9);
}

int main() {
  const int rows = 1 << 17;   // 512 MiB
  uint8_t* base; float* dst; hipMalloc(&base, (size_t)rows * K); hipMemset(base, 1, (size_t)rows * K); hipMalloc(&dst, 4096);
  for (int wgs : {512, 1024, 2048}) {
    run<0>("A: row per wave-load (1 KiB contiguous)", base, dst, rows, wgs);
    run<1>("B: 16 rows x 64 B per wave-load", base, dst, rows, wgs);
    run<2>("C: 16 rows x 64 B, 256 B per row back-to-back", base, dst, rows, wgs);
  }
  return 0;
}
