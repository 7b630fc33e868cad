// batch_lab.hip — stand-alone timing / timestamp harness for the no-K-split batched-decode kernel (development aid; no Python, no torch).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I hqq_amd/csrc [-DBT_LAB_TS] tools/lab_batch/batch_lab.hip hqq_amd/csrc/common.hip -o tools/lab_batch/batch_lab[_ts].bin
//   batch_lab.bin <shape: o|qkv|gateup|down> <M>
#ifdef BT_LAB_TS
unsigned long long* g_bt_lab_ts = nullptr;
#endif
#include "batch.hip"
#include <vector>
#include <algorithm>
#include <stdlib.h>
#include <string.h>
using namespace hqq;
int main(int argc, char** argv) {
  const char* shape = argc > 1 ? argv[1] : "o";
  const int M = argc > 2 ? atoi(argv[2]) : 32;
  int nl = 1; int64_t Ns[4] = {4096, 0, 0, 0}; int K = 4096;
  if (!strcmp(shape, "qkv")) { nl = 3; Ns[0] = Ns[1] = Ns[2] = 4096; }
  if (!strcmp(shape, "gateup")) { nl = 2; Ns[0] = Ns[1] = 11008; }
  if (!strcmp(shape, "down")) { K = 11008; }
  const int pool = 12;
  std::vector<std::vector<void*>> wq(pool), sc(pool), ze(pool);
  std::vector<uint16_t> hm;
  for (int i = 0; i < pool; ++i)
    for (int l = 0; l < nl; ++l) {
      const size_t wq_b = (size_t)Ns[l] * K / 2, meta_b = (size_t)Ns[l] * (K / 64) * 2;
      void *w, *s, *z; hipMalloc(&w, wq_b); hipMalloc(&s, meta_b); hipMalloc(&z, meta_b);
      hipMemset(w, 0x5A, wq_b);
      hm.resize(meta_b / 2);
      for (auto& v : hm) v = 0x1C00 + (rand() & 0x3FF);
      hipMemcpy(s, hm.data(), meta_b, hipMemcpyHostToDevice);
      for (auto& v : hm) v = 0x4400 + (rand() & 0x7FF);
      hipMemcpy(z, hm.data(), meta_b, hipMemcpyHostToDevice);
      wq[i].push_back(w); sc[i].push_back(s); ze[i].push_back(z);
    }
  void* x; hipMalloc(&x, (size_t)M * K * 2);
  std::vector<void*> ys(nl);
  for (int l = 0; l < nl; ++l) hipMalloc(&ys[l], (size_t)M * Ns[l] * 2);
  std::vector<uint16_t> hx((size_t)M * K); for (auto& v : hx) v = 0x3800 + (rand() & 0x3FF);
  hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
  hipStream_t st; hipStreamCreate(&st);
  auto call = [&](int i) {
    int rc = bt::batch_run(4, nl, x, (const void* const*)wq[i].data(), (const void* const*)sc[i].data(), (const void* const*)ze[i].data(), nullptr, ys.data(), Ns, M, K, HQQ_F16, st);
    if (rc) { printf("rc=%d %s\n", rc, hqq_hip_last_error()); exit(1); } };
#ifdef BT_LAB_TS
  const int nw = 600 * 9;
  hipMalloc(&g_bt_lab_ts, nw * 64); hipMemset(g_bt_lab_ts, 0, nw * 64);
  for (int i = 0; i < pool; ++i) call(i);
  hipStreamSynchronize(st);
  std::vector<unsigned long long> h(nw * 8);
  hipMemcpy(h.data(), g_bt_lab_ts, nw * 64, hipMemcpyDeviceToHost);
  unsigned long long t0 = ~0ull;
  for (int w = 0; w < nw; ++w) if (h[w * 8]) t0 = h[w * 8] < t0 ? h[w * 8] : t0;
#ifdef BT_LAB_FINE
  const char* names[8] = {"wave start", "iteration 2 begins", "  its wait done", "  its barrier passed", "  reads+rebuild+MFMA issued, lgkm drained", "  DMA issued (end)", "iteration 3 ends", "wave end"};
#else
  const char* names[8] = {"wave start", "units looked up", "prologue issued", "barrier 0 passed", "barrier 1 passed", "last barrier passed", "loop end", "wave end"};
#endif
  for (int kind = 0; kind < 2; ++kind) {
    printf("%s %s M=%d: time since the first wave start, units of 10 ns: min / median / max over the waves\n", shape, kind ? "LOADER waves" : "compute waves", M);
    for (int i = 0; i < 8; ++i) {
      std::vector<unsigned long long> v;
      for (int w = 0; w < nw; ++w) if (((w % 9) == 0) == (kind == 1) && h[w * 8] && h[w * 8 + i]) v.push_back(h[w * 8 + i] - t0);
      if (v.empty()) continue;
      std::sort(v.begin(), v.end());
      printf("  %-42s n=%5zu  %7llu %7llu %7llu\n", names[i], v.size(), v.front(), v[v.size() / 2], v.back());
    }
  }
  return 0;
#else
  for (int i = 0; i < pool; ++i) call(i);
  hipStreamSynchronize(st);
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < pool; ++i) call(i);
  hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraphLaunch(ge, st); hipStreamSynchronize(st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, st);
  const int reps = 20;
  for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, st);
  hipEventRecord(e1, st); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%s M=%d: %.2f us per launch\n", shape, M, ms * 1000.f / (reps * pool));
  return 0;
#endif
}
