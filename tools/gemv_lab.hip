// gemv_lab.hip — stand-alone timing harness for the fused dequant-GEMV (development aid; no Python, no torch).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DGV_...] tools/gemv_lab.hip hqq_amd/csrc/common.hip -o /tmp/gemv_lab
// Streams a pool of distinct random layers (> 256 MiB) back-to-back on one stream and prints the mean time per launch.
unsigned long long* g_lab_ts = nullptr;
#include "../hqq_amd/csrc/gemv.hip"
#include <vector>
#include <algorithm>
#include <stdlib.h>

struct LayerBuf { void *wq, *sc, *ze, *y; };

static double run_case(int nbits, int n_group, int N, int K, int M, int reps) {
  const int gs = 64;
  const size_t wq_b = (size_t)N * K * nbits / 8, meta_b = (size_t)N * (K / gs) * 2;
  const size_t per_layer = wq_b + 2 * meta_b;
  int pool = (int)(700e6 / (per_layer * n_group)) + 1;
  if (pool < 3) pool = 3;
  if (getenv("LAB_POOL")) pool = atoi(getenv("LAB_POOL"));
  std::vector<LayerBuf> L(pool * n_group);
  std::vector<uint16_t> hmeta(meta_b / 2);
  for (auto& b : L) {
    hipMalloc(&b.wq, wq_b); hipMalloc(&b.sc, meta_b); hipMalloc(&b.ze, meta_b); hipMalloc(&b.y, (size_t)M * N * 2);
    hipMemset(b.wq, 0x5A, wq_b);
    for (auto& v : hmeta) v = 0x1C00 + (rand() & 0x3FF);          // fp16 ~ 0.004..0.008
    hipMemcpy(b.sc, hmeta.data(), meta_b, hipMemcpyHostToDevice);
    for (auto& v : hmeta) v = 0x4400 + (rand() & 0x7FF);          // fp16 ~ 4..16
    hipMemcpy(b.ze, hmeta.data(), meta_b, hipMemcpyHostToDevice);
  }
  void* x; hipMalloc(&x, (size_t)M * K * 2);
  std::vector<uint16_t> hx((size_t)M * K);
  for (auto& v : hx) v = 0x3800 + (rand() & 0x3FF) + ((rand() & 1) << 15);
  hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
  hipStream_t st; hipStreamCreate(&st);
  auto sweep = [&]() {
    for (int p = 0; p < pool; ++p) {
      const void *wq[8], *sc[8], *ze[8]; void* y[8]; int64_t Ns[8];
      for (int i = 0; i < n_group; ++i) { auto& b = L[p * n_group + i]; wq[i] = b.wq; sc[i] = b.sc; ze[i] = b.ze; y[i] = b.y; Ns[i] = N; }
      int rc = hqq_hip_gemv_grouped(nbits, n_group, x, wq, sc, ze, nullptr, y, Ns, M, K, gs, HQQ_F16, getenv("LAB_EXACT4") ? 0u : HQQ_OPT_META_SCALABLE, nullptr, 0, st);
      if (rc) { printf("rc=%d %s\n", rc, hqq_hip_last_error()); exit(1); }
    }
  };
  sweep(); hipStreamSynchronize(st);
  // eager launches are host-bound below ~3.4 us each: replay the sweep from a graph (LAB_EAGER=1 for the eager number)
  hipGraphExec_t ge = nullptr;
  if (!getenv("LAB_EAGER")) {
    hipGraph_t g;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    sweep();
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, st);
  for (int r = 0; r < reps; ++r) { if (ge) hipGraphLaunch(ge, st); else sweep(); }
  hipEventRecord(e1, st);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double t = ms * 1e-3 / (reps * pool);
  const double bytes = (double)n_group * (per_layer + 2.0 * K * M / n_group + 2.0 * N * M);
  printf("int%d %d x %5dx%-5d M=%d  %8.2f us  %7.1f GB/s  %5.1f%% of 8 TB/s (pool %d)\n", nbits, n_group, N, K, M, t * 1e6, bytes / t / 1e9, bytes / t / 8e12 * 100, pool);
  for (auto& b : L) { hipFree(b.wq); hipFree(b.sc); hipFree(b.ze); hipFree(b.y); }
  hipFree(x); hipStreamDestroy(st);
  return t;
}

int main(int argc, char** argv) {
  const int reps = getenv("LAB_REPS") ? atoi(getenv("LAB_REPS")) : 3;
#ifdef GV_LAB_TS
  {
    // gemv_lab N K [layers in the group] [launches]: per-wave time stamps of the LAST of `launches` back-to-back launches
    const int nw = 1024 * 8;
    hipMalloc(&g_lab_ts, nw * 8 * 8); hipMemset(g_lab_ts, 0, nw * 8 * 8);
    setenv("LAB_EAGER", "1", 1);
    setenv("LAB_POOL", argc > 4 ? argv[4] : "6", 1);
    run_case(4, argc > 3 ? atoi(argv[3]) : 1, atoi(argv[1]), atoi(argv[2]), 1, 1);
    std::vector<unsigned long long> h(nw * 8);
    hipMemcpy(h.data(), g_lab_ts, nw * 64, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull; int live = 0;
    for (int w = 0; w < nw; ++w) if (h[w * 8]) { t0 = h[w * 8] < t0 ? h[w * 8] : t0; ++live; }
    const char* names[8] = {"wave start", "unit A requested", "x staged (stores issued)", "barrier passed", "unit B requested", "unit A consumed", "wave end", "arguments there"};
    printf("%d waves; us since the first wave's start: min / median / p90 / max\n", live);
    for (int i : {0, 7, 1, 2, 3, 4, 5, 6}) {
      std::vector<double> v;
      for (int w = 0; w < nw; ++w) if (h[w * 8] && h[w * 8 + i]) v.push_back((h[w * 8 + i] - t0) * 0.01);
      if (v.empty()) continue;
      std::sort(v.begin(), v.end());
      printf("  %-26s %6.2f %6.2f %6.2f %6.2f   (%zu waves)\n", names[i], v.front(), v[v.size() / 2], v[v.size() * 9 / 10], v.back(), v.size());
    }
    return 0;
  }
#endif
  if (argc > 1 && argv[1][0] == 's') { run_case(4, 1, 1024, 28672, 1, reps); run_case(4, 1, 1024, 8192, 1, reps); run_case(4, 1, 8192, 28672, 1, reps); run_case(4, 1, 2048, 28672, 1, reps); run_case(4, 1, 128, 8192, 1, reps); run_case(4, 1, 4096, 11008, 1, reps); return 0; }
  if (argc > 1 && argv[1][0] == 'm') { for (int M : {1, 4, 8, 16}) { run_case(4, 1, 11008, 4096, M, reps); run_case(4, 1, 28672, 8192, M, reps); } return 0; }
  if (argc > 1) { run_case(4, 1, 4096, 4096, 1, reps); run_case(4, 1, 11008, 4096, 1, reps); run_case(4, 1, 28672, 8192, 1, reps); return 0; }
  run_case(4, 1, 4096, 4096, 1, reps);
  run_case(4, 1, 11008, 4096, 1, reps);
  run_case(4, 1, 4096, 11008, 1, reps);
  run_case(4, 3, 4096, 4096, 1, reps);
  run_case(4, 2, 11008, 4096, 1, reps);
  run_case(4, 1, 28672, 8192, 1, reps);
  run_case(4, 1, 11008, 4096, 2, reps);
  run_case(2, 2, 11008, 4096, 1, reps);
  return 0;
}
