// gemv3s.hip — 3-bit decode, second kernel: every packed word is loaded ONCE and feeds all ten slabs (M <= 4 rows), gfx950.
//
// Reference chain replaced: BitPack.unpack_3bit_32 -> [: R] -> (W_r - zero) * scale -> torch.matmul(x, W.t()) (+ bias)
//   hqq/core/bitpack.py:95-110, hqq/core/quantize.py:183-199 (the 3-bit slice at :190-195), :880-898.
//
// Layout (the reference's, no repacking): P [step, 64] int32, step = ceil(R / 10), R = N * G group rows; group row r = n * G + g
// sits in slab t = r / step at packed row p = r - t * step, bits [27 - 3t, 29 - 3t].  gemv3.hip gives a wave one OUTPUT row, so
// each word is fetched ten times (nine of them from L2) for 3 of its 30 bits — ~10 instructions per weight.  Here the unit of
// work is a run of 16 PACKED rows (4 KiB): a wave loads it once, dequantises all 160 weights per lane and accumulates ten dot
// products at a time.  For slab t those 16 rows are groups g0 .. g0 + 15 of output row n0 (g0 = (p0 + t step) mod G, n0 = (p0 +
// t step) / G) and, if g0 + 15 >= G, the first groups of row n0 + 1: at most one row change per (task, slab) because G >= 16.
// A task therefore produces up to twenty partial sums per activation row — [task][slab][segment] in an fp32 scratch — and a
// small second kernel adds, for every output row, its partials in a fixed order (ascending slab, ascending task), rounds and
// adds the bias: no atomics, reproducible bits.
//
// Per task and wave: 4 weight loads of 16 B per lane (lane (c = lane & 15, o = lane >> 4) holds words 4c .. 4c + 3 of packed row
// 4i + o: a wave instruction reads 1 KiB of consecutive memory) and 6 two-byte loads of group constants (lane (o, c) fetches
// zero / scale of (slab 4q + o, row c); ds_bpermute hands them to the lanes that need them).  x sits in LDS in natural order
// with its first 16 groups repeated behind K, so that "group g0 + row" never has to wrap.
// Level extraction: v_perm_b32 picks the two bytes that hold a slab's field from a PAIR of words (no shifts; four selectors
// cover the ten slabs), v_and_or onto the fp16 exponent 0x6400 gives (1024 + F q, 1024 + F q') for two neighbouring k, one fma
// makes the exact levels, then - zero, * scale as Quantizer.dequantize does (two fp16 roundings); v_dot2_f32_f16 contracts.
#include <stdlib.h>

#include <type_traits>

#include "hqq_common.h"

namespace hqq {


constexpr int S3_MAXL = HQQ_GEMV_MAX_GROUP;
constexpr int S3_WAVES = 4;
constexpr int S3_ROWS = 16;      // packed rows per task
constexpr int S3_MAX_M = 4;

struct S3Args {
  const int32_t* Wq[S3_MAXL];
  const half_t* scale[S3_MAXL];
  const half_t* zero[S3_MAXL];
  const half_t* bias[S3_MAXL];
  half_t* y[S3_MAXL];
  int N[S3_MAXL];
  int step[S3_MAXL];        // ceil(N * G / 10)
  int task_end[S3_MAXL];    // end (exclusive) of layer i's tasks in the concatenated task space
  int smod[S3_MAXL];        // step mod G: slab t's packed row 0 starts at group (t * step) mod G of its output row — built up slab by
                            // slab with one add and one wrap in SGPRs (a table [layer][slab] indexed by the run-time layer cost one
                            // dependent scalar load per slab and task)
  const half_t* x;
  float* part;              // [task][10][2][M]
  int K, G, total_tasks, n_layers;
};

struct S3Layer {   // wave-uniform
  const int32_t* Wq;
  const half_t* scale;
  const half_t* zero;
  int N, step, task0, smod;
};

__device__ __forceinline__ S3Layer s3_select(const S3Args& a, int task) {
  S3Layer c{a.Wq[0], a.scale[0], a.zero[0], a.N[0], a.step[0], 0, a.smod[0]};
#pragma unroll
  for (int i = 1; i < S3_MAXL; ++i) {
    const bool in = task >= a.task_end[i - 1];
    c.Wq = pick(in, a.Wq[i], c.Wq);   // selects of VALUES (hqq_common.h): `in ? a.f[i] : c.f` selects the address and loads through it
    c.scale = pick(in, a.scale[i], c.scale);
    c.zero = pick(in, a.zero[i], c.zero);
    c.N = pick(in, a.N[i], c.N);
    c.step = pick(in, a.step[i], c.step);
    c.task0 = pick(in, a.task_end[i - 1], c.task0);
    c.smod = pick(in, a.smod[i], c.smod);
  }
  return c;
}

__device__ __forceinline__ float s3_wave_sum(float v) {
  auto dpp_add = [](float x, auto ctrl) {
    const int y = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xF, 0xF, true);
    return x + __builtin_bit_cast(float, y);
  };
  v = dpp_add(v, std::integral_constant<int, 0xB1>{});    // quad_perm [1,0,3,2]
  v = dpp_add(v, std::integral_constant<int, 0x4E>{});    // quad_perm [2,3,0,1]
  v = dpp_add(v, std::integral_constant<int, 0x141>{});   // row_half_mirror
  v = dpp_add(v, std::integral_constant<int, 0x140>{});   // row_mirror
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return (r0 + r1) + (r2 + r3);
}

// slab t: which byte pair of a word holds its field (class), and the field's bit position inside that pair
template <int T> struct S3Slab {
  static constexpr int cls = T >= 7 ? 0 : T >= 4 ? 1 : T >= 2 ? 2 : 3;                      // bytes (0,1) / (1,2) / (2,3) / (3)
  static constexpr int e = (27 - 3 * T) - (cls == 0 ? 0 : cls == 1 ? 8 : cls == 2 ? 16 : 24);   // 0 .. 7: F = 2^e, 7 F < 1024
};
// the same e for a run-time slab index (0x0363_1472_5303 read from the low nibble up: e of slabs 0..9 = 3,0,5,2,7,4,1,6,3,0)
__host__ __device__ __forceinline__ constexpr int s3_field_e(int t) { return static_cast<int>((0x0361472503ull >> (4 * t)) & 15ull); }
__device__ __forceinline__ constexpr uint32_t s3_sel(int cls) {
  return cls == 0 ? 0x05040100u : cls == 1 ? 0x06050201u : cls == 2 ? 0x07060302u : 0x0C070C03u;
}

struct S3Unit {
  u32x4 w[4];             // load i: words 4c .. 4c + 3 of packed row p0 + 4 i + o
  uint32_t z[3], sc[3];   // load q: zero / scale of (slab 4 q + o, packed row p0 + c); 32-bit holders of the zero-extended 2-byte loads
                          // (a uint16_t carried round the loop is masked — and waited for — in front of the next task's requests)
};

template <int M, bool SUB>
__global__ __launch_bounds__(S3_WAVES * 64) void gemv3s_kernel(const S3Args a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];   // x[M][K + 1024] fp16

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, o = lane >> 4;
  const int K = a.K, G = a.G;
  const int KP = K + S3_ROWS * 64;
  const int total = a.total_tasks;
  const int nwaves = gridDim.x * S3_WAVES;

  // buffer loads: the layer's base pointer in a wave-uniform descriptor + one 32-bit byte offset per lane (no 64-bit VALU address
  // arithmetic, fewer address temporaries: with global loads the register allocator reused the previous task's load destinations
  // for them and the compiler had to wait for that task's data in front of the next task's requests)
  auto issue = [&](S3Unit& un, const S3Layer& ly, int task, bool live) {
    const int p0 = (task - ly.task0) * S3_ROWS;
    const int R = ly.N * G;
    const __amdgpu_buffer_rsrc_t rw = buffer_rsrc(ly.Wq), rz = buffer_rsrc(ly.zero), rs = buffer_rsrc(ly.scale);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int p = p0 + 4 * i + o;
      p = p < ly.step ? p : ly.step - 1;   // rows past the tensor (last task): re-read the last row, weighted by zero below
      un.w[i] = __builtin_amdgcn_raw_buffer_load_b128(rw, live ? p * 256 + c * 16 : 0, 0, 2 /* nt */);   // dead task: one line for every wave
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int t = 4 * q + o, p = p0 + c;
      const int r = p + t * ly.step;
      const bool ok = t < 10 && p < ly.step && r < R;
      const int idx = ok && live ? r : 0;
      un.z[q] = __builtin_amdgcn_raw_buffer_load_b16(rz, idx * 2, 0, 0);
      un.sc[q] = __builtin_amdgcn_raw_buffer_load_b16(rs, idx * 2, 0, 0);
    }
  };

  int task = blockIdx.x * S3_WAVES + wave;
  S3Layer la = s3_select(a, task < total ? task : total - 1);
  S3Unit ua, ub;
  issue(ua, la, task < total ? task : total - 1, task < total);   // (waves without a task: uniform load counts; they leave after the barrier)

  // ---- x (first 16 groups repeated behind K) ----
  for (int v = tid; v < M * (KP >> 3); v += S3_WAVES * 64) {
    const int m = v / (KP >> 3), j = v - m * (KP >> 3);
    const int kk = j * 8 < K ? j * 8 : j * 8 - K;
    *reinterpret_cast<u32x4*>(smem + (static_cast<size_t>(m) * KP + j * 8) * 2) = *reinterpret_cast<const u32x4*>(a.x + static_cast<int64_t>(m) * K + kk);
  }
  __syncthreads();
  if (task >= total) return;

  uint32_t magic;
  asm volatile("v_mov_b32 %0, 0x64006400" : "=v"(magic));
  const uint32_t xlane = (o * 64 + c * 4) * 2;   // byte offset of this lane's k-quad inside "group g0 + 0" of activation row 0

  auto consume = [&](const S3Unit& un, const S3Layer& ly, int task_) {
    // wave-uniform bookkeeping in SGPRs (the branches below are then scalar branches, not exec-mask regions)
    const int task = __builtin_amdgcn_readfirstlane(task_);
    const int p0 = __builtin_amdgcn_readfirstlane((task - ly.task0) * S3_ROWS);
    const int p0g = __builtin_amdgcn_readfirstlane(p0 % G);
    const int R = ly.N * G;
    // group constants: (zero | scale << 16) of (slab 4 q + o, row c), zero weight for rows that do not exist
    uint32_t zs[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int t = 4 * q + o, p = p0 + c;
      const bool ok = t < 10 && p < ly.step && p + t * ly.step < R;
      zs[q] = ok ? (un.z[q] | (un.sc[q] << 16)) : 0u;
      if constexpr (SUB) {   // three-op rebuild (decode_common.h): (z, s) -> (z 2^-J, s 2^J), J = 9 - e(slab); exact by hqq_hip_meta_check
        const int J = 9 - s3_field_e(t < 10 ? t : 0);
        const half2_t f = {__builtin_bit_cast(half_t, static_cast<uint16_t>((15 - J) << 10)), __builtin_bit_cast(half_t, static_cast<uint16_t>((15 + J) << 10))};   // (2^-J, 2^J)
        zs[q] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2_t, zs[q]) * f);
      }
    }
    // the two bytes that hold a slab's field, from word pairs (k, k + 1): pk[class][load][pair]
    uint32_t pk[4][4][2];
#pragma unroll
    for (int cls = 0; cls < 4; ++cls)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) pk[cls][i][h] = __builtin_amdgcn_perm(un.w[i][2 * h + 1], un.w[i][2 * h], s3_sel(cls));
    float* dst = a.part + static_cast<int64_t>(task) * (10 * 2 * M);
    int tmod = 0;   // (T * step) mod G for the slab in hand

    auto slab = [&](auto tc) {
      constexpr int T = decltype(tc)::value;
      constexpr int e = S3Slab<T>::e, cls = S3Slab<T>::cls;
      constexpr uint32_t msk = (7u << e) | (7u << (e + 16));
      constexpr float inv = 1.0f / static_cast<float>(1 << e);
      const half2_t k1 = {static_cast<half_t>(inv), static_cast<half_t>(inv)};
      const half2_t k2 = {static_cast<half_t>(-1024.0f * inv), static_cast<half_t>(-1024.0f * inv)};
      int g0 = p0g + tmod;   // group (inside its output row) of the task's first packed row in this slab
      g0 = g0 >= G ? g0 - G : g0;
      tmod += ly.smod;       // slabs are visited in ascending order
      tmod = tmod >= G ? tmod - G : tmod;
      const int bd = G - g0;             // first local row of the NEXT output row (>= 16: none in this task)
      const uint32_t xa = xlane + static_cast<uint32_t>(g0) * 128u;
      uint32_t mine[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) mine[i] = __builtin_amdgcn_ds_bpermute((((T & 3) * 16 + 4 * i + o) << 2), zs[T >> 2]);
      half2_t w[4][2];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const half2_t pr = __builtin_bit_cast(half2_t, mine[i]);
        const half2_t zz = {pr.x, pr.x}, ss = {pr.y, pr.y};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if constexpr (SUB) {   // the masked field read as fp16 is the subnormal q 2^(e-24): one fma lifts it and subtracts z 2^-J (rounding 1), one mul by s 2^J (rounding 2)
            const half2_t lift = {static_cast<half_t>(32768.0f), static_cast<half_t>(32768.0f)};
            const half2_t q = __builtin_elementwise_fma(__builtin_bit_cast(half2_t, pk[cls][i][h] & msk), lift, -zz);
            w[i][h] = q * ss;
          } else {
            uint32_t b;
            asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(b) : "v"(pk[cls][i][h]), "s"(msk), "v"(magic));
            const half2_t q = __builtin_elementwise_fma(__builtin_bit_cast(half2_t, b), k1, k2);   // exact level
            w[i][h] = (q - zz) * ss;                                                             // two roundings, as Quantizer.dequantize
          }
        }
      }
      float sa[M], sb[M];
      if (bd >= S3_ROWS) {
#pragma unroll
        for (int m = 0; m < M; ++m) {
          float acc = 0.f;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const u32x2 xq = *reinterpret_cast<const u32x2*>(smem + xa + i * 512 + static_cast<size_t>(m) * KP * 2);
            const uint32_t x01 = xq.x, x23 = xq.y;   // (scalars first: __builtin_bit_cast of a vector ELEMENT reads element 0 with this compiler)
            acc = __builtin_amdgcn_fdot2(w[i][0], __builtin_bit_cast(half2_t, x01), acc, false);
            acc = __builtin_amdgcn_fdot2(w[i][1], __builtin_bit_cast(half2_t, x23), acc, false);
          }
          sa[m] = s3_wave_sum(acc);
        }
        if (lane == 0) {
#pragma unroll
          for (int m = 0; m < M; ++m) dst[(T * 2 + 0) * M + m] = sa[m];
        }
      } else {   // this task's rows of slab T belong to two output rows: local rows < bd to the first, the rest to the second
#pragma unroll
        for (int m = 0; m < M; ++m) {
          float accA = 0.f, accB = 0.f;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const u32x2 xq = *reinterpret_cast<const u32x2*>(smem + xa + i * 512 + static_cast<size_t>(m) * KP * 2);
            const uint32_t x01 = xq.x, x23 = xq.y;
            float d = __builtin_amdgcn_fdot2(w[i][0], __builtin_bit_cast(half2_t, x01), 0.f, false);
            d = __builtin_amdgcn_fdot2(w[i][1], __builtin_bit_cast(half2_t, x23), d, false);
            const bool first = (4 * i + o) < bd;
            accA += first ? d : 0.f;
            accB += first ? 0.f : d;
          }
          sa[m] = s3_wave_sum(accA);
          sb[m] = s3_wave_sum(accB);
        }
        if (lane == 0) {
#pragma unroll
          for (int m = 0; m < M; ++m) { dst[(T * 2 + 0) * M + m] = sa[m]; dst[(T * 2 + 1) * M + m] = sb[m]; }
        }
      }
    };
    slab(std::integral_constant<int, 0>{}); slab(std::integral_constant<int, 1>{}); slab(std::integral_constant<int, 2>{});
    slab(std::integral_constant<int, 3>{}); slab(std::integral_constant<int, 4>{}); slab(std::integral_constant<int, 5>{});
    slab(std::integral_constant<int, 6>{}); slab(std::integral_constant<int, 7>{}); slab(std::integral_constant<int, 8>{});
    slab(std::integral_constant<int, 9>{});
  };

  // every issue() emits the same ten loads (past the end: one cache line, never consumed), so the waits the compiler
  // derives are exact counts; one loop shape with ONE exit at the bottom (an exit from the middle runs through the loop latch
  // once the control flow is structurised, and the compiler then drains vmcnt in front of the next request: gemv.hip)
  bool more;
  do {
    const int t1 = task + nwaves;
    const bool live1 = t1 < total;
    const S3Layer lb = s3_select(a, live1 ? t1 : total - 1);
    issue(ub, lb, live1 ? t1 : total - 1, live1);
    consume(ua, la, task);
    const int t2 = t1 + nwaves;
    more = t2 < total;
    la = s3_select(a, more ? t2 : total - 1);
    issue(ua, la, more ? t2 : total - 1, more);
    if (live1) consume(ub, lb, t1);
    task = t2;
  } while (more);
}

// y[m][n] = sum of row n's partials: ascending slab, ascending task — a fixed order.  A row has G / 16 + 1 contributions per slab
// (5 at K = 4096, 29 at K = 28672): they are fetched eight at a time, all loads of a batch in flight before the first add (one
// load per loop iteration made this kernel five dependent memory round trips long: 4.9 us per call).
template <int M>
__global__ __launch_bounds__(256) void gemv3s_finish_kernel(const S3Args a) {
  const int l = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
  if (l >= a.n_layers || n >= a.N[l]) return;
  const int G = a.G, step = a.step[l];
  const int task0 = l ? a.task_end[l - 1] : 0;
  const int r0 = n * G, r1 = r0 + G - 1;
  float sum[M];
#pragma unroll
  for (int m = 0; m < M; ++m) sum[m] = 0.f;
  constexpr int B = 8;
  for (int t = r0 / step; t <= r1 / step; ++t) {
    const int ts = t * step;
    const int lo = (r0 > ts ? r0 : ts) - ts, hi = (r1 < ts + step - 1 ? r1 : ts + step - 1) - ts;
    const int jhi = hi / S3_ROWS;
    for (int j0 = lo / S3_ROWS; j0 <= jhi; j0 += B) {
      float v[B][M];
#pragma unroll
      for (int u = 0; u < B; ++u) {
        const int j = j0 + u <= jhi ? j0 + u : jhi;        // past the end: the last one again, dropped below
        const int seg = (j * S3_ROWS + ts >= r0) ? 0 : 1;   // 0: row n is the task's first output row in this slab, 1: its second
        const float* p = a.part + ((static_cast<int64_t>(task0 + j) * 10 + t) * 2 + seg) * M;
#pragma unroll
        for (int m = 0; m < M; ++m) v[u][m] = p[m];
      }
#pragma unroll
      for (int u = 0; u < B; ++u) {
        const bool keep = j0 + u <= jhi;
#pragma unroll
        for (int m = 0; m < M; ++m) sum[m] += keep ? v[u][m] : 0.f;
      }
    }
  }
#pragma unroll
  for (int m = 0; m < M; ++m) {
    half_t o = static_cast<half_t>(sum[m]);
    if (a.bias[l]) o = o + a.bias[l][n];   // `out += bias` on the rounded matmul result (quantize.py:896-897)
    a.y[l][static_cast<int64_t>(m) * a.N[l] + n] = o;
  }
}

template <int M, bool SUB>
static int s3_launch(S3Args& a, int max_n, void* ws, size_t ws_bytes, hipStream_t st) {
  const size_t lds = static_cast<size_t>(M) * (a.K + S3_ROWS * 64) * 2;
  const size_t need = WS_COUNTER_BYTES + static_cast<size_t>(a.total_tasks) * 10 * 2 * M * sizeof(float);   // (the head belongs to the split-K counters)
  if (!ws || ws_bytes < need) { set_error("hqq_hip_gemv(3-bit): the slab-sharing kernel parks %zu bytes of partial sums in the workspace (hqq_hip_gemv_workspace_bytes), got %zu", need, ws ? ws_bytes : size_t(0)); return HQQ_ERR_WORKSPACE; }
  if (!aligned16(ws)) { set_error("hqq_hip_gemv: workspace must be 16-byte aligned"); return HQQ_ERR_ALIGN; }
  a.part = reinterpret_cast<float*>(static_cast<char*>(ws) + WS_COUNTER_BYTES);
  int n_cus = 256, dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cus <= 0) n_cus = 256;
  int per_cu = static_cast<int>(160 * 1024 / (lds + 256));
  per_cu = per_cu > 4 ? 4 : (per_cu < 1 ? 1 : per_cu);
  const int wgs = (a.total_tasks + S3_WAVES - 1) / S3_WAVES;
  const int cap = n_cus * per_cu;
  auto kern = gemv3s_kernel<M, SUB>;
  if (lds > 64 * 1024) {
    static LdsRaised raised;
    if (const int rc = raise_lds_limit(raised, reinterpret_cast<const void*>(kern), 144 * 1024, "hqq_hip_gemv")) return rc;
  }
  hipLaunchKernelGGL(kern, dim3(wgs < cap ? wgs : cap), dim3(S3_WAVES * 64), lds, st, a);
  int rc = check_launch("hqq_hip_gemv(3-bit)");
  if (rc) return rc;
  hipLaunchKernelGGL(gemv3s_finish_kernel<M>, dim3((max_n + 255) / 256, a.n_layers), dim3(256), 0, st, a);
  return check_launch("hqq_hip_gemv(3-bit)");
}

// shapes this kernel takes over from gemv3.hip: at least 16 groups per output row (a task's rows of one slab then span at most
// two output rows) and x + its 16-group tail within the LDS budget
bool gemv3s_covers(int64_t M, int64_t K, int64_t group_size) {
  if (group_size != 64 || M < 1 || M > S3_MAX_M || K % 64 != 0 || K / 64 < S3_ROWS) return false;
  return static_cast<size_t>(M) * (K + S3_ROWS * 64) * 2 <= 144 * 1024;
}

// called by gemv3_run (gemv3.hip) after the common argument checks; same contract
// bytes of partial sums a launch of this shape parks in the caller's workspace
size_t gemv3s_workspace_bytes(int n_layers, const int64_t* N, int64_t M, int64_t K) {
  const int64_t G = K / 64;
  int64_t tasks = 0;
  for (int i = 0; i < n_layers; ++i) tasks += ((N[i] * G + 9) / 10 + S3_ROWS - 1) / S3_ROWS;
  return WS_COUNTER_BYTES + static_cast<size_t>(tasks) * 10 * 2 * static_cast<size_t>(M) * sizeof(float);
}

int gemv3s_run(int n_layers, const void* x, const void* const* Wq, const void* const* scale, const void* const* zero,
               const void* const* bias, void* const* y, const int64_t* N, int64_t M, int64_t K, uint32_t opts, void* ws, size_t ws_bytes, hipStream_t st) {
  S3Args a;
  const int64_t G = K / 64;
  int64_t tasks = 0, max_n = 0;
  for (int i = 0; i < n_layers; ++i) {
    const int64_t R = N[i] * G;
    a.Wq[i] = static_cast<const int32_t*>(Wq[i]);
    a.scale[i] = static_cast<const half_t*>(scale[i]);
    a.zero[i] = static_cast<const half_t*>(zero[i]);
    a.bias[i] = bias ? static_cast<const half_t*>(bias[i]) : nullptr;
    a.y[i] = static_cast<half_t*>(y[i]);
    a.N[i] = static_cast<int>(N[i]);
    a.step[i] = static_cast<int>((R + 9) / 10);
    tasks += (a.step[i] + S3_ROWS - 1) / S3_ROWS;
    if (tasks > INT32_MAX / 128) { set_error("hqq_hip_gemv: size overflow"); return HQQ_ERR_SHAPE; }
    a.task_end[i] = static_cast<int>(tasks);
    a.smod[i] = static_cast<int>(a.step[i] % G);
    max_n = N[i] > max_n ? N[i] : max_n;
  }
  for (int i = n_layers; i < S3_MAXL; ++i) {
    a.Wq[i] = a.Wq[n_layers - 1]; a.scale[i] = a.scale[n_layers - 1]; a.zero[i] = a.zero[n_layers - 1]; a.bias[i] = a.bias[n_layers - 1];
    a.y[i] = a.y[n_layers - 1]; a.N[i] = a.N[n_layers - 1]; a.step[i] = a.step[n_layers - 1]; a.task_end[i] = a.task_end[n_layers - 1];
    a.smod[i] = a.smod[n_layers - 1];
  }
  a.x = static_cast<const half_t*>(x);
  a.K = static_cast<int>(K);
  a.G = static_cast<int>(G);
  a.total_tasks = static_cast<int>(tasks);
  a.n_layers = n_layers;
  const bool sub = (opts & HQQ_OPT_META_SCALABLE) != 0;
  switch (M) {
    case 1: return sub ? s3_launch<1, true>(a, static_cast<int>(max_n), ws, ws_bytes, st) : s3_launch<1, false>(a, static_cast<int>(max_n), ws, ws_bytes, st);
    case 2: return sub ? s3_launch<2, true>(a, static_cast<int>(max_n), ws, ws_bytes, st) : s3_launch<2, false>(a, static_cast<int>(max_n), ws, ws_bytes, st);
    case 3: return sub ? s3_launch<3, true>(a, static_cast<int>(max_n), ws, ws_bytes, st) : s3_launch<3, false>(a, static_cast<int>(max_n), ws, ws_bytes, st);
    case 4: return sub ? s3_launch<4, true>(a, static_cast<int>(max_n), ws, ws_bytes, st) : s3_launch<4, false>(a, static_cast<int>(max_n), ws, ws_bytes, st);
  }
  return HQQ_ERR_SHAPE;
}

}  // namespace hqq
