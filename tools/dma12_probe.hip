// dma12_probe.hip — where does global_load_lds with 12 bytes per lane put its data?  (lab; answers a layout question of gemm_pipe.hip's 3-bit path)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((address_space(3))) void* lds_t;
typedef const __attribute__((address_space(1))) void* glb_t;
__global__ void k(const uint32_t* src, uint32_t* out) {
  __shared__ __attribute__((aligned(1024))) uint32_t lds[512];
  for (int i = threadIdx.x; i < 512; i += 64) lds[i] = 0xDEAD0000u + i;
  __syncthreads();
  __builtin_amdgcn_global_load_lds((glb_t)(src + threadIdx.x * 3), (lds_t)lds, 12, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) out[i] = lds[i];
}
int main() {
  uint32_t h[192], *d, *o, r[512];
  for (int i = 0; i < 192; ++i) h[i] = (i / 3) * 16 + (i % 3);   // lane * 16 + dword
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
  hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  for (int i = 0; i < 40; ++i) printf("lds[%d] = 0x%x\n", i, r[i]);
  int stride12 = 1, stride16 = 1;
  for (int l = 0; l < 64; ++l) for (int q = 0; q < 3; ++q) { if (r[l * 3 + q] != (uint32_t)(l * 16 + q)) stride12 = 0; if (r[l * 4 + q] != (uint32_t)(l * 16 + q)) stride16 = 0; }
  printf("stride12 %d stride16 %d\n", stride12, stride16);
  return 0;
}
