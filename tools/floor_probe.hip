// floor_probe.hip — what 128 DEPENDENT launches cost when they only stream the bytes of the Llama-2-7B int4 stack (development aid).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/floor_probe.hip -o tools/floor_probe.bin
// The kernel has the decode kernel's launch shape (persistent grid of 1024 workgroups x 4 waves, a wave walks whole 4 KiB / 11 KiB rows in
// 2 KiB units, two units in flight, buffer loads with nt) and NONE of its arithmetic: every 16-byte vector is folded into one float with
// four adds, a row ends in a wave reduction and a 2-byte store.  Variants: + an 8 KiB x row staged through LDS behind a barrier before the
// first unit is consumed (what the GEMV's prologue does).  Launch sizes: q|k|v 28.3 MB, o 9.45, gate|up 50.8, down 25.4, 32 blocks, every
// launch on its own bytes of one 3.7 GB buffer (HBM traffic, no cache reuse), one hipGraph, events around 20 replays.
// Prints ms per "token" and the fraction of 8 TB/s on the stack's algorithmic bytes — the ceiling ANY kernel behind the same launch
// structure has on this chip.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// NF units in flight per wave (a ring of NF register sets), UL loads of 16 B per lane and unit (UL KiB per wave and unit)
template <bool STAGE_X, int NF, int UL, int VW = 0, int ML = 0>   // VW: packed-fp16 VALU ops per 16-byte vector (0: four adds); ML: 2-byte meta loads per unit
__global__ __launch_bounds__(1024) void stream_kernel(const uint8_t* __restrict__ base, int rows, int row_bytes, const _Float16* __restrict__ x, _Float16* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = blockDim.x >> 6;
  const int stride = gridDim.x * nw;
  const int nunits = (row_bytes + UL * 1024 - 1) / (UL * 1024);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(base), 0, -1, 0x00020000);
  auto issue = [&](u32x4 (&v)[UL], int r, int u) {
    const bool live = r < rows;
    const uint32_t roff = live ? static_cast<uint32_t>(r) * static_cast<uint32_t>(row_bytes) : 0u;
#pragma unroll
    for (int h = 0; h < UL; ++h) {
      int k0 = (u * UL + h) * 1024 + lane * 16;
      k0 = (live && k0 < row_bytes) ? k0 : 0;
      v[h] = __builtin_amdgcn_raw_buffer_load_b128(rw, k0, roff, 2);
    }
  };
  u32x4 ring[NF][UL];
  uint32_t meta[NF][ML > 0 ? ML : 1];
  const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(x), 0, -1, 0x00020000);
  int rr[NF], ru[NF];
  u32x4 xv;
  if (STAGE_X) xv = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint8_t*>(x) + (tid & 511) * 16);
  int r = blockIdx.x * nw + wave, u = 0;
  auto next = [&](int& r_, int& u_) { if (++u_ >= nunits) { u_ = 0; r_ += stride; } };
  auto issue_meta = [&](uint32_t (&m)[ML > 0 ? ML : 1], int r_, int u_) {
#pragma unroll
    for (int q = 0; q < ML; ++q) m[q] = __builtin_amdgcn_raw_buffer_load_b16(rm, ((r_ * 64 + u_ * 16 * UL + lane) & 16383) * 2, q * 32768, 0);
  };
#pragma unroll
  for (int f = 0; f < NF; ++f) { rr[f] = r; ru[f] = u; issue_meta(meta[f], r, u); issue(ring[f], r, u); next(r, u); if (f == 0 && STAGE_X) { reinterpret_cast<u32x4*>(smem)[tid] = xv; __syncthreads(); } }
  float acc = STAGE_X ? __uint_as_float(reinterpret_cast<const uint32_t*>(smem)[lane ^ 1]) * 1e-30f : 0.f;
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  auto fold = [&](const u32x4 (&v)[UL], const uint32_t (&m)[ML > 0 ? ML : 1]) {
    if constexpr (VW == 0) {
#pragma unroll
      for (int h = 0; h < UL; ++h) acc += __uint_as_float(v[h].x & 0x3F800000u) + __uint_as_float(v[h].y & 0x3F800000u) + __uint_as_float(v[h].z & 0x3F800000u) + __uint_as_float(v[h].w & 0x3F800000u);
    } else {
      uint32_t mm = 0x3C003C00u;
#pragma unroll
      for (int q = 0; q < ML; ++q) mm ^= (m[q] & 0x00010001u);
      const h2 c1 = __builtin_bit_cast(h2, mm), c2 = {static_cast<_Float16>(0.25f), static_cast<_Float16>(-0.25f)};
#pragma unroll
      for (int h = 0; h < UL; ++h) {
        h2 t[4] = {__builtin_bit_cast(h2, v[h].x & 0x03FF03FFu), __builtin_bit_cast(h2, v[h].y & 0x03FF03FFu), __builtin_bit_cast(h2, v[h].z & 0x03FF03FFu), __builtin_bit_cast(h2, v[h].w & 0x03FF03FFu)};
#pragma unroll
        for (int i = 0; i < VW / 4; ++i)
#pragma unroll
          for (int d = 0; d < 4; ++d) t[d] = __builtin_elementwise_fma(t[d], c1, c2);
        acc += static_cast<float>(t[0].x + t[1].y) + static_cast<float>(t[2].x + t[3].y);
      }
    }
  };
  auto row_end = [&](int r_) {
    float f = acc;
    for (int o = 32; o; o >>= 1) f += __shfl_xor(f, o);
    if (lane == 0) y[r_] = static_cast<_Float16>(f);
    acc = 0.f;
  };
  bool more = rr[0] < rows;
  while (more) {
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      if (rr[f] < rows) { fold(ring[f], meta[f]); if (ru[f] + 1 >= nunits) row_end(rr[f]); }
      rr[f] = r; ru[f] = u;
      issue_meta(meta[f], r, u);
      issue(ring[f], r, u);
      next(r, u);
    }
    more = rr[0] < rows;
  }
}

// the same with perfect balance: wave w streams the contiguous KiB range [w * n / W, (w + 1) * n / W) of the launch, whatever the rows
template <int NF, int UL>
__global__ __launch_bounds__(1024) void flat_kernel(const uint8_t* __restrict__ base, int rows, int row_bytes, const _Float16* __restrict__ x, _Float16* __restrict__ y) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x >> 6, W = gridDim.x * nw, w = blockIdx.x * nw + wave;
  const long long total = (static_cast<long long>(rows) * row_bytes) / (UL * 1024);   // units of the launch
  const long long u0 = total * w / W, u1 = total * (w + 1) / W;
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(base), 0, -1, 0x00020000);
  u32x4 ring[NF][UL];
  long long cur = u0;
  auto issue = [&](u32x4 (&v)[UL], long long u) {
    const uint32_t off = static_cast<uint32_t>((u < u1 ? u : u0) * (UL * 1024));
#pragma unroll
    for (int h = 0; h < UL; ++h) v[h] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16 + h * 1024, off, 2);
  };
#pragma unroll
  for (int f = 0; f < NF; ++f) { issue(ring[f], cur); ++cur; }
  float acc = 0.f;
  long long done = u0;
  while (done < u1) {
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      if (done < u1) {
#pragma unroll
        for (int h = 0; h < UL; ++h) acc += __uint_as_float(ring[f][h].x & 0x3F800000u) + __uint_as_float(ring[f][h].y & 0x3F800000u) + __uint_as_float(ring[f][h].z & 0x3F800000u) + __uint_as_float(ring[f][h].w & 0x3F800000u);
      }
      ++done;
      issue(ring[f], cur); ++cur;
    }
  }
  for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) y[w] = static_cast<_Float16>(acc);
}

// access patterns of the tile kernels (skinny.hip, gemm_pipe.hip), perfect balance, 1 KiB per wave instruction:
//   PAT 0: 1 KiB contiguous   PAT 1: 16 rows x 64 B (an MFMA A fragment's natural load)   PAT 2: 8 rows x 128 B (whole lines)
template <int NF, int PAT>
__global__ __launch_bounds__(1024) void panel_kernel(const uint8_t* __restrict__ base, int rows, int row_bytes, const _Float16* __restrict__ x, _Float16* __restrict__ y) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x >> 6, W = gridDim.x * nw, w = blockIdx.x * nw + wave;
  const long long total = (static_cast<long long>(rows) * row_bytes) / 1024;
  const long long u0 = total * w / W, u1 = total * (w + 1) / W;
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(base), 0, -1, 0x00020000);
  const int upp = PAT == 1 ? row_bytes / 64 : row_bytes / 128;
  const uint32_t lane_off = PAT == 0 ? lane * 16u : PAT == 1 ? static_cast<uint32_t>(lane & 15) * row_bytes + (lane >> 4) * 16u : static_cast<uint32_t>(lane & 7) * row_bytes + (lane >> 3) * 16u;
  u32x4 ring[NF];
  long long cur = u0;
  auto issue = [&](u32x4& v, long long u) {
    const long long uu = u < u1 ? u : u0;
    uint32_t off;
    if (PAT == 0) off = static_cast<uint32_t>(uu * 1024);
    else { const uint32_t p = static_cast<uint32_t>(uu / upp), kb = static_cast<uint32_t>(uu % upp); off = p * (PAT == 1 ? 16u : 8u) * row_bytes + kb * (PAT == 1 ? 64u : 128u); }
    v = __builtin_amdgcn_raw_buffer_load_b128(rw, lane_off, off, 2);
  };
#pragma unroll
  for (int f = 0; f < NF; ++f) { issue(ring[f], cur); ++cur; }
  float acc = 0.f;
  long long done = u0;
  while (done < u1) {
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      if (done < u1) acc += __uint_as_float(ring[f].x & 0x3F800000u) + __uint_as_float(ring[f].y & 0x3F800000u) + __uint_as_float(ring[f].z & 0x3F800000u) + __uint_as_float(ring[f].w & 0x3F800000u);
      ++done;
      issue(ring[f], cur); ++cur;
    }
  }
  for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) y[w] = static_cast<_Float16>(acc);
}

int main() {
  struct L { int rows, row_bytes; };   // packed rows x bytes per packed row (int4: K bytes per packed row, 2 output rows each)
  const L blk[4] = {{6144, 4096}, {2048, 4096}, {11008, 4096}, {2048, 11008}};   // q|k|v, o, gate|up, down
  size_t per_block = 0;
  for (auto& l : blk) per_block += static_cast<size_t>(l.rows) * l.row_bytes;
  const int nblocks = 32;
  const double alg_bytes = 3647750144.0;   // the stack's algorithmic bytes per token (packed weights + meta + x + y)
  uint8_t* buf; _Float16 *x, *y;
  hipMalloc(&buf, per_block * nblocks); hipMemset(buf, 0x3C, per_block * nblocks);
  hipMalloc(&x, 65536); hipMemset(x, 0, 65536); hipMalloc(&y, 65536 * 2);
  auto run = [&](const char* name, auto kern, int threads, int wg_per_cu, size_t lds) {
    hipStream_t st; hipStreamCreate(&st);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    size_t off = 0;
    const int nw = threads / 64;
    for (int b = 0; b < nblocks; ++b)
      for (auto& l : blk) {
        const int tiles = (l.rows + nw - 1) / nw, cap = 256 * wg_per_cu, grid = tiles < cap ? tiles : cap;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, st, (const uint8_t*)(buf + off), l.rows, l.row_bytes, (const _Float16*)x, y);
        off += static_cast<size_t>(l.rows) * l.row_bytes;
      }
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int w = 0; w < 5; ++w) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, st);
    const int reps = 30;
    for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, st);
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("%-58s %.4f ms per token   %.0f GB/s packed   %.3f of 8 TB/s (algorithmic bytes)   %.2f us per launch\n",
           name, ms, per_block * nblocks / (ms * 1e-3) / 1e9, alg_bytes / (ms * 1e-3) / 8e12, ms * 1e3 / (4 * nblocks));
    hipGraphExecDestroy(ge); hipGraphDestroy(g); hipStreamDestroy(st);
  };
  run("4x4, 2 units x 2 KiB, stream only (the GEMV's shape)", stream_kernel<false, 2, 2>, 256, 4, 0);
  run("4x4, 3 units x 1 KiB, stream only (best found)", stream_kernel<false, 3, 1>, 256, 4, 0);
  run("4x4, 2 units x 2 KiB, stream only, PERFECT BALANCE (equal bytes per wave)", flat_kernel<2, 2>, 256, 4, 0);
  run("4x4, 3 units x 1 KiB, stream only, perfect balance", flat_kernel<3, 1>, 256, 4, 0);
  run("4x4, 4 units x 1 KiB, stream only, perfect balance", flat_kernel<4, 1>, 256, 4, 0);
  run("4x4, 2 x 2 KiB + x staging + 64 pk ops per 16 B + 4 meta loads / unit", stream_kernel<true, 2, 2, 64, 4>, 256, 4, 8704);
  run("4x4, 3 x 1 KiB + x staging + 64 pk ops per 16 B + 2 meta loads / unit", stream_kernel<true, 3, 1, 64, 2>, 256, 4, 8704);
  run("4x4, 4 x 1 KiB contiguous, perfect balance", panel_kernel<4, 0>, 256, 4, 0);
  run("4x4, 4 x (16 rows x 64 B), perfect balance", panel_kernel<4, 1>, 256, 4, 0);
  run("4x4, 4 x (8 rows x 128 B), perfect balance", panel_kernel<4, 2>, 256, 4, 0);
  run("8x1, 8 x 1 KiB contiguous, perfect balance", panel_kernel<8, 0>, 512, 1, 0);
  run("8x1, 8 x (16 rows x 64 B), perfect balance", panel_kernel<8, 1>, 512, 1, 0);
  run("8x1, 8 x (8 rows x 128 B), perfect balance", panel_kernel<8, 2>, 512, 1, 0);
  run("8x2, 4 x (16 rows x 64 B), perfect balance", panel_kernel<4, 1>, 512, 2, 0);
  run("8x2, 4 x (8 rows x 128 B), perfect balance", panel_kernel<4, 2>, 512, 2, 0);
  // ---- the metric's literal shape: one 4096 x 4096 int4 layer per launch (8.39 MB packed, 9.45 MB algorithmic), 32 dependent launches ----
  {
    const int rows = 2048, row_bytes = 4096, chain = 32;
    auto run1 = [&](const char* name, auto kern, int threads, int wg_per_cu) {
      hipStream_t st; hipStreamCreate(&st);
      hipGraph_t g; hipGraphExec_t ge;
      hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
      const int nw = threads / 64;
      for (int i = 0; i < chain; ++i) {
        const int tiles = (rows + nw - 1) / nw, cap = 256 * wg_per_cu, grid = tiles < cap ? tiles : cap;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, st, (const uint8_t*)(buf + static_cast<size_t>(i) * rows * row_bytes * 4), rows, row_bytes, (const _Float16*)x, y);
      }
      hipStreamEndCapture(st, &g);
      hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
      for (int w = 0; w < 5; ++w) hipGraphLaunch(ge, st);
      hipStreamSynchronize(st);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0, st);
      const int reps = 50;
      for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, st);
      hipEventRecord(e1, st); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double us = ms * 1e3 / (reps * chain);
      printf("one 4096x4096 layer per launch, %-40s %.2f us per launch   %.3f of 8 TB/s (9,453,568 algorithmic bytes)\n", name, us, 9453568.0 / (us * 1e-6) / 8e12);
      hipGraphExecDestroy(ge); hipGraphDestroy(g); hipStreamDestroy(st);
    };
    run1("8 waves x 2/CU, 2 units x 2 KiB, stream only", stream_kernel<false, 2, 2>, 512, 2);
    run1("4 x 4, 2 units x 2 KiB, stream only", stream_kernel<false, 2, 2>, 256, 4);
    run1("8 x 2, 4 units x 1 KiB, stream only", stream_kernel<false, 4, 1>, 512, 2);
    run1("4 x 4, 4 units x 1 KiB, stream only", stream_kernel<false, 4, 1>, 256, 4);
    run1("16 x 1, 2 units x 2 KiB, stream only", stream_kernel<false, 2, 2>, 1024, 1);
  }
  return 0;
}
