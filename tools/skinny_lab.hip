// skinny_lab.hip — stand-alone timing / timestamp harness for the skinny-GEMM kernel (development aid; no Python, no torch).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include [-DSK_LAB_TS] tools/skinny_lab.hip hqq_amd/csrc/common.hip -o tools/lab_skinny.bin
unsigned long long* g_sk_lab_ts = nullptr;
#include "../hqq_amd/csrc/skinny.hip"
#include <vector>
#include <algorithm>
#include <stdlib.h>
using namespace hqq;
int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 11008, K = argc > 2 ? atoi(argv[2]) : 4096, M = argc > 3 ? atoi(argv[3]) : 8;
  const size_t wq_b = (size_t)N * K / 2, meta_b = (size_t)N * (K / 64) * 2;
  const int pool = 24;
  std::vector<void*> wq(pool), sc(pool), ze(pool);
  std::vector<uint16_t> hm(meta_b / 2);
  for (int i = 0; i < pool; ++i) {
    hipMalloc(&wq[i], wq_b); hipMalloc(&sc[i], meta_b); hipMalloc(&ze[i], meta_b);
    hipMemset(wq[i], 0x5A, wq_b);
    for (auto& v : hm) v = 0x1C00 + (rand() & 0x3FF);
    hipMemcpy(sc[i], hm.data(), meta_b, hipMemcpyHostToDevice);
    for (auto& v : hm) v = 0x4400 + (rand() & 0x7FF);
    hipMemcpy(ze[i], hm.data(), meta_b, hipMemcpyHostToDevice);
  }
  void *x, *y; hipMalloc(&x, (size_t)M * K * 2); hipMalloc(&y, (size_t)M * N * 2);
  std::vector<uint16_t> hx((size_t)M * K); for (auto& v : hx) v = 0x3800 + (rand() & 0x3FF);
  hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
  hipStream_t st; hipStreamCreate(&st);
  const int64_t Ns[1] = {N};
  auto call = [&](int i) { const void* w1[1] = {wq[i]}; const void* s1[1] = {sc[i]}; const void* z1[1] = {ze[i]}; void* y1[1] = {y};
    int rc = skinny_run(4, 1, x, w1, s1, z1, nullptr, y1, Ns, M, K, HQQ_F16, st); if (rc) { printf("rc=%d %s\n", rc, hqq_hip_last_error()); exit(1); } };
#ifdef SK_LAB_TS
  const int nw = 4096 * 8;
  hipMalloc(&g_sk_lab_ts, nw * 64); hipMemset(g_sk_lab_ts, 0, nw * 64);
  for (int i = 0; i < pool; ++i) call(i);
  hipStreamSynchronize(st);
  std::vector<unsigned long long> h(nw * 8);
  hipMemcpy(h.data(), g_sk_lab_ts, nw * 64, hipMemcpyDeviceToHost);
  // the LAST launch of the loop is what the buffer holds: stamps against the earliest wave start of that launch (one clock for all CUs)
  unsigned long long t0 = ~0ull;
  int live = 0;
  for (int w = 0; w < nw; ++w) if (h[w * 8]) { t0 = h[w * 8] < t0 ? h[w * 8] : t0; ++live; }
  printf("%d waves; time since the first wave start in units of 10 ns (s_memrealtime): min / median / max over the waves\n", live);
  const char* names[8] = {"wave start", "arguments read", "group constants requested", "x + both units requested", "group constants in LDS", "half-iteration 1", "half-iteration 2", "wave end"};
  for (int i = 0; i < 8; ++i) {
    std::vector<unsigned long long> v;
    for (int w = 0; w < nw; ++w) if (h[w * 8] && h[w * 8 + i]) v.push_back(h[w * 8 + i] - t0);
    if (v.empty()) continue;
    std::sort(v.begin(), v.end());
    printf("  %-28s n=%5zu  %7llu %7llu %7llu\n", names[i], v.size(), v.front(), v[v.size() / 2], v.back());
  }
  return 0;
#endif
  for (int i = 0; i < pool; ++i) call(i);
  hipStreamSynchronize(st);
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < pool; ++i) call(i);
  hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraphLaunch(ge, st); hipStreamSynchronize(st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, st);
  for (int r = 0; r < 5; ++r) hipGraphLaunch(ge, st);
  hipEventRecord(e1, st); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double t = ms * 1e-3 / (5 * pool);
  printf("int4 %dx%d M=%d: %.2f us  %.0f GB/s\n", N, K, M, t * 1e6, (wq_b + 2 * meta_b) / t / 1e9);
  return 0;
}
