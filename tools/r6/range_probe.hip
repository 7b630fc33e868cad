// range_probe.hip — what the gfx950 buffer range check covers (raw buffer, stride 0):  hipcc --offload-arch=gfx950 -O2 -o range_probe.bin range_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const uint32_t* base, int num_records, int soff, int voff_mul, uint32_t* out) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(base), 0, num_records, 0x00020000);
  const int lane = threadIdx.x;
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, lane * voff_mul, soff, 0);
  const uint32_t h = __builtin_amdgcn_raw_buffer_load_b16(r, lane * 2, soff, 0);
  out[lane * 5 + 0] = v.x; out[lane * 5 + 1] = v.y; out[lane * 5 + 2] = v.z; out[lane * 5 + 3] = v.w; out[lane * 5 + 4] = h;
}
int main() {
  const int n = 1 << 16;
  std::vector<uint32_t> h(n);
  for (int i = 0; i < n; ++i) h[i] = 0xA0000000u | i;   // dword index tagged
  uint32_t *d, *o;
  hipMalloc(&d, n * 4); hipMalloc(&o, 64 * 5 * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  std::vector<uint32_t> r(64 * 5);
  struct { int nr, soff, vm; const char* what; } cases[] = {
    {1024, 0, 16, "nr=1024 soff=0: lanes 0..63 in range"},
    {1000, 0, 16, "nr=1000 soff=0: lane 62 partial (992..1007), lane 63 out"},
    {1024, 512, 16, "nr=1024 soff=512: is soff part of the check? lanes >= 32 are past nr when it is"},
    {0, 0, 16, "nr=0: everything out"},
    {0, 4096, 16, "nr=0 soff=4096"},
  };
  for (auto& c : cases) {
    hipMemset(o, 0xFF, 64 * 5 * 4);
    probe<<<1, 64>>>(d, c.nr, c.soff, c.vm, o);
    hipDeviceSynchronize();
    hipMemcpy(r.data(), o, 64 * 5 * 4, hipMemcpyDeviceToHost);
    printf("%s\n", c.what);
    for (int l : {0, 31, 32, 61, 62, 63}) printf("  lane %2d: %08x %08x %08x %08x | b16 %08x\n", l, r[l * 5], r[l * 5 + 1], r[l * 5 + 2], r[l * 5 + 3], r[l * 5 + 4]);
  }
  // null base with nr = 0
  hipMemset(o, 0xFF, 64 * 5 * 4);
  probe<<<1, 64>>>(nullptr, 0, 0, 16, o);
  hipError_t e = hipDeviceSynchronize();
  hipMemcpy(r.data(), o, 64 * 5 * 4, hipMemcpyDeviceToHost);
  printf("null base, nr=0: %s  lane0 %08x\n", hipGetErrorString(e), r[0]);
  return 0;
}
