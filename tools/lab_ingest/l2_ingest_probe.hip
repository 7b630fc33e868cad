// l2_ingest_probe.hip — how fast can ONE CU pull bytes that sit in its XCD's L2 (or stream from HBM), as a function of the waves that pull and of the
// loads each keeps in flight?  (development aid; DESIGN.md section 3.12's "a CU fills its L1 at ~30 GB/s" put to a direct test.)
//   hipcc --offload-arch=gfx950 -O3 -o l2_ingest_probe l2_ingest_probe.hip && ./l2_ingest_probe
// Every workgroup reads the SAME `span` bytes (span <= 1 MiB: resident in each XCD's 4 MiB L2 after the first pass; span = 0: a private HBM stream
// per workgroup) `passes` times with 16-byte loads, U loads in flight per thread; the xor of everything read is stored so nothing is elided.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ __launch_bounds__(1024) void pull(const u32x4* __restrict__ buf, size_t span_vec, size_t private_vec, int passes, u32x4* out) {
  const int T = blockDim.x;
  const u32x4* base = buf + (private_vec ? static_cast<size_t>(blockIdx.x) * private_vec : 0);
  const size_t n = private_vec ? private_vec : span_vec;
  u32x4 acc = {0, 0, 0, 0};
  for (int p = 0; p < passes; ++p) {
    for (size_t i = threadIdx.x; i + static_cast<size_t>(U - 1) * T < n; i += static_cast<size_t>(U) * T) {
      u32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(base + i + static_cast<size_t>(u) * T) : base[i + static_cast<size_t>(u) * T];
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= v[u];
    }
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[blockIdx.x * T + threadIdx.x] = acc;
}

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); exit(1); } } while (0)

template <int U, bool NT>
static double run(const u32x4* buf, size_t span, size_t priv, int passes, int wgs, int threads, u32x4* out) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((pull<U, NT>), dim3(wgs), dim3(threads), 0, 0, buf, span / 16, priv / 16, passes, out);
  CK(hipDeviceSynchronize());
  double best = 1e30;
  for (int r = 0; r < 3; ++r) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((pull<U, NT>), dim3(wgs), dim3(threads), 0, 0, buf, span / 16, priv / 16, passes, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  return best * 1e3;   // us
}

int main() {
  const size_t total = size_t(1) << 30;
  u32x4* buf; u32x4* out;
  CK(hipMalloc(&buf, total)); CK(hipMalloc(&out, size_t(2048) * 1024 * 16));
  CK(hipMemset(buf, 1, total));
  printf("source, waves per CU, loads in flight per thread -> us, GB/s per CU (256 CUs busy), aggregate TB/s\n");
  const int cus = 256;
  struct Geo { int wgs_per_cu, threads; } geos[] = {{1, 64}, {1, 256}, {1, 512}, {1, 1024}, {2, 1024}};
  for (int src = 0; src < 3; ++src) {          // 0: 256 KiB shared (L2), 1: 1 MiB shared (L2), 2: private 1 MiB HBM stream per workgroup
    for (const Geo& g : geos) {
      const int wgs = cus * g.wgs_per_cu;
      const size_t span = src == 0 ? (256u << 10) : (1u << 20);
      const size_t priv = src == 2 ? ((size_t(1) << 20) / g.wgs_per_cu) : 0;
      const int passes = src == 2 ? 1 : (src == 0 ? 16 : 4) / g.wgs_per_cu;
      const double bytes_per_cu = src == 2 ? double(1 << 20) : double(span) * passes * g.wgs_per_cu;
      double t1 = run<1, false>(buf, span, priv, passes, wgs, g.threads, out);
      double t4 = run<4, false>(buf, span, priv, passes, wgs, g.threads, out);
      double t8 = run<8, false>(buf, span, priv, passes, wgs, g.threads, out);
      double t8n = run<8, true>(buf, span, priv, passes, wgs, g.threads, out);
      const char* nm = src == 0 ? "L2 256K shared" : src == 1 ? "L2 1M shared  " : "HBM private 1M";
      printf("%s  waves/CU %2d :  U1 %7.1f us %6.1f GB/s/CU | U4 %7.1f us %6.1f | U8 %7.1f us %6.1f (%.2f TB/s) | U8 nt %7.1f us %6.1f\n", nm,
             g.wgs_per_cu * g.threads / 64, t1, bytes_per_cu / t1 / 1e3, t4, bytes_per_cu / t4 / 1e3, t8, bytes_per_cu / t8 / 1e3, bytes_per_cu * cus / t8 / 1e6, t8n, bytes_per_cu / t8n / 1e3);
      fflush(stdout);
    }
  }
  // a few CUs only: is the limit per CU or shared?
  for (int wgs : {8, 32, 64, 128}) {
    double t8 = run<8, false>(buf, 1u << 20, 0, 4, wgs, 1024, out);
    printf("L2 1M shared, %3d workgroups x 16 waves: U8 %7.1f us %6.1f GB/s per workgroup\n", wgs, t8, double(4u << 20) / t8 / 1e3);
  }
  return 0;
}
