// gemm_lab.hip — per-role cycle accounting of the wave-specialised dequant-GEMM (development aid).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DGEMM_LAB_TS tools/gemm_lab.hip hqq_amd/csrc/common.hip hqq_amd/csrc/gemv.hip hqq_amd/csrc/gemv_mfma.hip hqq_amd/csrc/gemv3.hip -o tools/gemm_lab.bin
#include "../hqq_amd/csrc/gemm.hip"
#include <vector>
#include <stdlib.h>
int main() {
  const int M = 8192, N = 4096, K = 4096, gs = 64;
  void *x, *wq, *sc, *ze, *y; unsigned long long* ts;
  hipMalloc(&x, (size_t)M * K * 2); hipMalloc(&wq, (size_t)N * K / 2); hipMalloc(&sc, (size_t)N * K / gs * 2); hipMalloc(&ze, (size_t)N * K / gs * 2); hipMalloc(&y, (size_t)M * N * 2);
  hipMalloc(&ts, 64 * 8 * 5 * 8); hipMemset(ts, 0, 64 * 8 * 5 * 8);
  std::vector<uint16_t> h((size_t)M * K);
  for (auto& v : h) v = 0x3800 + (rand() & 0x3FF) + ((rand() & 1) << 15);
  hipMemcpy(x, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  std::vector<uint8_t> hw((size_t)N * K / 2); for (auto& v : hw) v = rand();
  hipMemcpy(wq, hw.data(), hw.size(), hipMemcpyHostToDevice);
  std::vector<uint16_t> hm((size_t)N * K / gs);
  for (auto& v : hm) v = 0x1C00 + (rand() & 0x3FF);
  hipMemcpy(sc, hm.data(), hm.size() * 2, hipMemcpyHostToDevice);
  for (auto& v : hm) v = 0x4400 + (rand() & 0x7FF);
  hipMemcpy(ze, hm.data(), hm.size() * 2, hipMemcpyHostToDevice);
#ifdef GEMM_LAB_TS
  hipMemcpyToSymbol(HIP_SYMBOL(hqq::g_ws_ts_dev), &ts, sizeof(ts));
#endif
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) {
    hipEventRecord(e0);
    int rc = hqq_hip_gemm(4, x, wq, sc, ze, nullptr, y, M, N, K, gs, HQQ_F16, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("rc=%d  %.3f ms  %.0f TFLOP/s\n", rc, ms, 2.0 * M * N * K / ms / 1e9);
  }
  std::vector<unsigned long long> t(64 * 8 * 5);
  hipMemcpy(t.data(), ts, t.size() * 8, hipMemcpyDeviceToHost);
  for (int b : {0, 17}) {
    printf("block %d (cycles per K-step):\n", b);
    for (int w = 0; w < 8; ++w) {
      const unsigned long long* o = &t[(b * 8 + w) * 5];
      const double nk = o[4] ? (double)o[4] : 1.0;
      printf("  wave %d: write_x %.0f  dequant+issue %.0f  ds_read+mfma %.0f  barrier %.0f\n", w, o[0] / nk, o[1] / nk, o[2] / nk, o[3] / nk);
    }
  }
  return 0;
}
