// Loads-only probe of a BALANCED tile plan for decode with a batch of 32 rows: what would a kernel reach that
//   * runs G persistent workgroups of 8 waves (1 or 2 per CU), each with an equal share of a launch's (K slice, panel, chunk) items — no partial waves,
//   * stages the K slice of x (32 rows) ONCE per workgroup (again only if its range crosses into the next slice),
//   * streams the packed weights in skinny.hip's access pattern (16 packed rows x 64 B per wave instruction = the MFMA A-operand layout) with DI items
//     (2 KiB of weights per wave) in flight per wave, the group constants of every (panel, slice) segment of the range fetched up front in whole pieces of their lines — every iteration issues the same loads, so the waits are counted (the ISA is
//     checked for it: s_waitcnt vmcnt(n > 0) in the loop),
//   * writes one fp32 partial tile (64 packed rows x 2 slabs x 32 tokens = 16 KiB) per (panel, slice) segment it finishes,
// and does NO arithmetic and NO finish?  The 7B stack's 128 dependent launches (q|k|v, o, gate|up, down x 32 blocks, distinct weights: 3.5 GB), hipGraph replay.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o balanced_probe.bin balanced_probe.hip && ./balanced_probe.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct Args {
  const uint8_t* W; const uint16_t* zero; const uint16_t* scale; const uint16_t* x; float* part; uint32_t* sink;
  int Np, Kb, KS, P, C, CS;   // packed rows, bytes per packed row (= K at 4 bits), K slices, panels of 64 rows, 256-byte chunks per row, chunks per slice
  int T;                      // items = KS * P * CS in (slice, panel, chunk-in-slice) order
  int x_mode, part_mode, lds_bytes;
};

constexpr int WAVES = 8;

template <int DI, bool META>
__global__ __launch_bounds__(WAVES * 64) void probe_kernel(Args a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int G = gridDim.x, b = blockIdx.x;
  const int i0 = static_cast<int>(static_cast<int64_t>(a.T) * b / G), i1 = static_cast<int>(static_cast<int64_t>(a.T) * (b + 1) / G);
  if (i0 >= i1) return;
  uint32_t acc = 0;
  // a panel-chunk = 64 rows x 256 B = 16 wave loads (4 row groups x 4 quarters of 64 B); wave w takes row group w & 3, quarters 2 (w >> 2) and + 1;
  // its group constants: 4 B per lane (2 of the chunk's 4 groups) for (row, slab, zero | scale): 2 KiB per item over the eight waves
  const int rg = wave & 3, q0 = (wave >> 2) * 2;
  const int r = lane & 15, o = lane >> 4;
  const int mrow = (lane & 15), mslab = (lane >> 4) & 1, mt = lane >> 5;
  const int Gr = a.C * 4;
  struct Slot { u32x4 w0, w1; };
  Slot ring[DI];
  auto issue = [&](Slot& sl, int it_) {
    const int it = it_ < i1 ? it_ : i1 - 1;   // (past the range: the last item again — cache hits; every iteration issues the same loads)
    const int c = it % a.CS, p = (it / a.CS) % a.P, s = it / (a.CS * a.P);
    int ch = s * a.CS + c; ch = ch < a.C ? ch : a.C - 1;   // (an uneven last slice: its phantom chunks re-read the row's last one)
    int row = p * 64 + rg * 16 + r; row = row < a.Np ? row : a.Np - 1;
    const uint8_t* w = a.W + static_cast<size_t>(row) * a.Kb + static_cast<size_t>(ch) * 256 + q0 * 64 + o * 16;
    sl.w0 = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(w));
    sl.w1 = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(w + 64));
  };
#pragma unroll
  for (int d = 0; d < DI; ++d) issue(ring[d], i0 + d);
  // ---- group constants of every (panel, slice) segment of the range, up front (as skinny.hip does for its K split): 256 (row, slab, tensor) lines per segment,
  //      CS x 8 bytes of each; one lane per line and 16-byte piece, into LDS ----
  if (META) {
    const int seg0 = i0 / a.CS, seg1 = (i1 - 1) / a.CS;          // segment = (slice, panel) index in item order
    const int pieces = (a.CS * 8 + 15) / 16;                      // 16-byte pieces per line
    for (int sg = seg0; sg <= seg1; ++sg) {
      const int p = sg % a.P, s = sg / a.P;
      for (int q = tid; q < 256 * pieces; q += WAVES * 64) {
        const int line = q / pieces, pc = q % pieces;
        int row = p * 64 + (line & 63); row = row < a.Np ? row : a.Np - 1;
        const int slab = (line >> 6) & 1, t = line >> 7;
        const uint16_t* base = t ? a.scale : a.zero;
        int g0 = s * a.CS * 4 + pc * 8; g0 = g0 + 8 <= Gr ? g0 : Gr - 8;
        const u32x4 v = *reinterpret_cast<const u32x4*>(base + (static_cast<size_t>(row) + static_cast<size_t>(slab) * a.Np) * Gr + g0);
        *reinterpret_cast<u32x4*>(smem + (static_cast<size_t>(q + (sg - seg0) * 256 * pieces) * 16) % a.lds_bytes) = v;
      }
    }
    if (!a.x_mode) __syncthreads();
  }
  // ---- x: the slice(s) this workgroup's range touches, once, behind the first requests ----
  if (a.x_mode) {
    const int per_slice = a.P * a.CS;
    const int s0 = i0 / per_slice, s1 = (i1 - 1) / per_slice;
    for (int s = s0; s <= s1; ++s) {
      const int kbytes = a.CS * 256 * 2;             // k values of the slice x 2 bytes, per row of x
      const int n16 = 32 * kbytes / 16;              // 16-byte pieces: 32 rows
      for (int q = tid; q < n16; q += WAVES * 64) {
        const int row = q / (kbytes / 16), c = q % (kbytes / 16);
        const u32x4 v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint8_t*>(a.x) + (static_cast<size_t>(row) * 16384 * 2) + static_cast<size_t>(s) * kbytes + c * 16);
        *reinterpret_cast<u32x4*>(smem + (static_cast<size_t>(q) * 16) % a.lds_bytes) = v;
      }
    }
    __syncthreads();
  }
  const int n = i1 - i0;
  for (int base = 0; base < n; base += DI) {
#pragma unroll
    for (int d = 0; d < DI; ++d) {
      const int it = i0 + base + d;
      const Slot v = ring[d];
      issue(ring[d], it + DI);
      acc ^= v.w0.x ^ v.w0.y ^ v.w0.z ^ v.w0.w ^ v.w1.x ^ v.w1.y ^ v.w1.z ^ v.w1.w;
      // a (panel, slice) segment ends when the next item is another panel / slice or the range ends: its partial tile goes out
      if (a.part_mode && it < i1 && ((it + 1 == i1) || ((it + 1) % a.CS == 0))) {
        const int p = (it / a.CS) % a.P, s = it / (a.CS * a.P);
        float* dst = a.part + (static_cast<size_t>(s) * a.P + p) * (64 * 2 * 32) + ((it + 1 == i1 && (it + 1) % a.CS != 0) ? static_cast<size_t>(a.KS) * a.P * 64 * 2 * 32 : 0);
#pragma unroll
        for (int f = 0; f < 2; ++f) {   // 16 KiB per segment = 1024 float4 over 512 threads
          float4 w4 = {__uint_as_float(acc), 0.f, 0.f, 0.f};
          reinterpret_cast<float4*>(dst)[f * WAVES * 64 + tid] = w4;
        }
      }
    }
  }
  if (acc == 0x12345678u) a.sink[0] = acc;
}

struct Shape { const char* name; int Np, Kb; };

int main(int argc, char** argv) {
  const Shape shapes[4] = {{"q|k|v", 6144, 4096}, {"o", 2048, 4096}, {"gate|up", 11008, 4096}, {"down", 2048, 11008}};
  const int NB = 32, M = 32;
  size_t wbytes = 0, mbytes = 0;
  for (auto& s : shapes) { wbytes += static_cast<size_t>(s.Np) * s.Kb; mbytes += static_cast<size_t>(s.Np) * 2 * (s.Kb / 64) * 2; }
  uint8_t* W; uint16_t *Z, *S, *X; float* part; uint32_t* sink;
  CK(hipMalloc(&W, wbytes * NB)); CK(hipMalloc(&Z, mbytes * NB)); CK(hipMalloc(&S, mbytes * NB));
  CK(hipMalloc(&X, 32 * 16384 * 2)); CK(hipMalloc(&part, 256u << 20)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(W, 0x5a, wbytes * NB)); CK(hipMemset(Z, 1, mbytes * NB)); CK(hipMemset(S, 2, mbytes * NB)); CK(hipMemset(X, 3, 32 * 16384 * 2));
  // algorithmic bytes of the stack at 32 rows (bench.py: W_q + 2 R 2 + 2 K M + 2 N M)
  double alg = 0;
  for (auto& s : shapes) alg += NB * (static_cast<double>(s.Np) * s.Kb + s.Np * 2.0 * (s.Kb / 64) * 2 * 2 + 2.0 * s.Kb * M + 2.0 * (2.0 * s.Np) * M);
  printf("7B stack at %d rows: %.3f GB algorithmic per step, 128 dependent launches\n", M, alg / 1e9);
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

  auto run_cfg = [&](int DI, int wg_per_cu, int KS_small, int KS_big, int x_mode, int part_mode, int meta_mode, bool only_shape, int which) -> double {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    size_t woff = 0, moff = 0;
    for (int blk = 0; blk < NB; ++blk) {
      for (int si = 0; si < 4; ++si) {
        const Shape& s = shapes[si];
        if (!only_shape || si == which) {
          Args a;
          a.W = W + woff; a.zero = Z + moff / 2; a.scale = S + moff / 2; a.x = X; a.part = part; a.sink = sink;
          a.Np = s.Np; a.Kb = s.Kb; a.P = (s.Np + 63) / 64; a.C = s.Kb / 256;
          int KS = (s.Np > 4096) ? KS_big : KS_small;
          if (KS > a.C) KS = a.C;
          a.KS = KS; a.CS = (a.C + KS - 1) / KS; a.T = KS * a.P * a.CS;
          a.x_mode = x_mode; a.part_mode = part_mode;
          const int G = 256 * wg_per_cu;
          const size_t lds = (x_mode || meta_mode) ? 96 * 1024 / wg_per_cu : 0;
          a.lds_bytes = static_cast<int>(lds ? lds : 16);
#define LAUNCH(DD) do { if (meta_mode) hipLaunchKernelGGL((probe_kernel<DD, true>), dim3(G), dim3(WAVES * 64), lds, st, a); else hipLaunchKernelGGL((probe_kernel<DD, false>), dim3(G), dim3(WAVES * 64), lds, st, a); } while (0)
          switch (DI) { case 1: LAUNCH(1); break; case 2: LAUNCH(2); break; case 3: LAUNCH(3); break; case 4: LAUNCH(4); break; case 6: LAUNCH(6); break; case 8: LAUNCH(8); break; default: printf("no such instantiation\n"); exit(1); }
        }
        woff += static_cast<size_t>(s.Np) * s.Kb; moff += static_cast<size_t>(s.Np) * 2 * (s.Kb / 64) * 2;
      }
    }
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    const int reps = 20;
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return ms / reps;
  };
#define ATTR(DD) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe_kernel<DD, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)); CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe_kernel<DD, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024))
  ATTR(1); ATTR(2); ATTR(3); ATTR(4); ATTR(6); ATTR(8);

  struct Cfg { int DI, wgcu, KSs, KSb, x, part, meta; };
  const Cfg cfgs[] = {
    // weights only: the streaming pattern's own ceiling in this launch structure, by items (2 KiB per wave) in flight
    {1, 1, 1, 1, 0, 0, 0}, {2, 1, 1, 1, 0, 0, 0}, {1, 2, 1, 1, 0, 0, 0},
    // + group constants (per segment, up front)
    {1, 1, 1, 1, 0, 0, 1}, {2, 1, 1, 1, 0, 0, 1}, {1, 1, 4, 4, 0, 0, 1}, {1, 1, 8, 4, 0, 0, 1},
    // + partial tiles only (K slices: small launches / big launches)
    {1, 1, 2, 2, 0, 1, 1}, {1, 1, 4, 2, 0, 1, 1}, {1, 1, 4, 4, 0, 1, 1}, {1, 1, 8, 4, 0, 1, 1},
    // + x only
    {1, 1, 2, 2, 1, 0, 1}, {1, 1, 4, 2, 1, 0, 1}, {1, 1, 4, 4, 1, 0, 1}, {1, 1, 8, 4, 1, 0, 1}, {1, 1, 8, 8, 1, 0, 1}, {1, 1, 16, 8, 1, 0, 1},
    // everything
    {1, 1, 2, 2, 1, 1, 1}, {1, 1, 4, 2, 1, 1, 1}, {1, 1, 4, 4, 1, 1, 1}, {1, 1, 8, 4, 1, 1, 1}, {1, 1, 8, 8, 1, 1, 1}, {1, 1, 16, 8, 1, 1, 1}, {2, 1, 4, 4, 1, 1, 1}, {2, 1, 8, 4, 1, 1, 1}, {2, 1, 8, 8, 1, 1, 1}, {1, 2, 8, 4, 1, 1, 1}, {1, 2, 8, 8, 1, 1, 1},
  };

  printf("%-72s %8s %7s   per launch us: q|k|v o gate|up down\n", "configuration (8 waves per workgroup)", "ms/step", "frac");
  for (const Cfg& c : cfgs) {
    const double ms = run_cfg(c.DI, c.wgcu, c.KSs, c.KSb, c.x, c.part, c.meta, false, 0);
    double per[4];
    for (int si = 0; si < 4; ++si) per[si] = run_cfg(c.DI, c.wgcu, c.KSs, c.KSb, c.x, c.part, c.meta, true, si) / NB * 1e3;
    char name[160];
    snprintf(name, sizeof name, "%d WG/CU, %d items (%2d KiB per wave) in flight, KS %d/%d%s%s%s", c.wgcu, c.DI, 2 * c.DI, c.KSs, c.KSb, c.meta ? ", meta" : "", c.x ? ", x once" : "", c.part ? ", partials" : "");
    printf("%-72s %8.3f %7.4f   %6.2f %6.2f %6.2f %6.2f\n", name, ms, alg / (ms * 1e-3) / 8e12, per[0], per[1], per[2], per[3]);
    fflush(stdout);
  }
  return 0;
}
