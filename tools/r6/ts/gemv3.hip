// gemv3.hip — fused unpack -> dequantize -> GEMV for 3-bit HQQ layers (M <= 4 activation rows), gfx950.
//
// Reference chain replaced: BitPack.unpack_3bit_32 -> [: R] -> (W_r - zero) * scale -> torch.matmul(x, W.t()) (+ bias)
//   hqq/core/bitpack.py:95-110, hqq/core/quantize.py:183-199 (the 3-bit slice at :190-195), :880-898.
//
// Layout (as stored by the reference, no repacking): P [step, gs=64] int32, step = ceil(R/10), R = N*G groups; the unpacked
// group row r = n*G + g lives in slab s = r / step at packed row p = r - s*step, bits [27-3s, 29-3s] of every word.  Ten
// *unrelated* output rows share each packed row, and step is not a multiple of G, so there is no packed-row range that
// maps to whole output rows for all ten slabs at once.
//
// Decomposition chosen: one wave per OUTPUT row.  Row n's G groups are G consecutive packed rows of one slab (two slabs
// when the row straddles a slab boundary — handled per lane): 256*G contiguous bytes of which the wave uses 3 bits per
// word.  The other nine users of the same bytes are the rows of the other slabs that start in the same "p-block"; rows are
// handed out so that those ten readers are neighbouring waves of ONE XCD (g3_row_of): HBM and the fabric see each packed byte
// about once, L2->CU traffic is 10x the packed bytes.  That trades L2 bandwidth (plentiful) for a kernel with no atomics, no
// cross-wave reduction and a deterministic summation order.  Round-1 status: 1.15 TB/s of packed bytes (14 % of 8 TB/s; 0.85
// before the XCD-aware order) — ~10 instructions per weight, of which 4 are the exact dequantisation.  Launches of >= 19 MB go to
// gemv3s.hip (one load feeding all ten slabs, partial sums added in a fixed order by a finishing pass); this kernel keeps the
// small ones, where its ~4 us of fixed cost beats the other's ~10.
//
// Per wave instruction: 64 lanes x 16 B = four groups; lane (i = lane & 15, j = lane >> 4) holds words 4i..4i+3 of group
// 4u + j.  Levels of a slab are pulled out two words at a time (v_lshrrev x2, v_perm, v_and_or onto the fp16 exponent
// 0x6400), then rebuilt exactly as Quantizer.dequantize does (-1024, -zero, *scale: two fp16 roundings) and contracted on
// the matrix core with the diagonal trick of gemv.hip (all lanes hold the same output row; D[i][i] are the partial sums).
#include <stdlib.h>

#include <type_traits>

#include "hqq_common.h"

namespace hqq {

constexpr int G3_MAXL = HQQ_GEMV_MAX_GROUP;
constexpr int G3_WAVES = 4;
constexpr int G3_U = 4;                    // loads per unit -> 16 groups = 1024 k
constexpr int G3_MAX_M = 4;

typedef _Float16 g3_h8_t __attribute__((ext_vector_type(8)));

struct G3Args {
  const int32_t* Wq[G3_MAXL];
  const half_t* scale[G3_MAXL];
  const half_t* zero[G3_MAXL];
  const half_t* bias[G3_MAXL];
  half_t* y[G3_MAXL];
  int N[G3_MAXL];
  int step[G3_MAXL];       // ceil(N*G / 10)
  int e_end[G3_MAXL];      // end (exclusive) of layer i's entries in one XCD's concatenated row stream (see the kernel)
  const half_t* x;
  int K, G, total_e;       // total_e: entries per XCD stream over all layers
};

struct G3Layer {
  const int32_t* Wq;
  const half_t* scale;
  const half_t* zero;
  const half_t* bias;
  half_t* y;
  int N, step, e0, end, li;
};

__device__ __forceinline__ G3Layer g3_select(const G3Args& a, int e) {
  G3Layer c{a.Wq[0], a.scale[0], a.zero[0], a.bias[0], a.y[0], a.N[0], a.step[0], 0, a.e_end[0], 0};
#pragma unroll
  for (int i = 1; i < G3_MAXL; ++i) {
    const bool in = e >= a.e_end[i - 1];
    c.Wq = pick(in, a.Wq[i], c.Wq);   // selects of VALUES (hqq_common.h): `in ? a.f[i] : c.f` selects the address and loads through it
    c.scale = pick(in, a.scale[i], c.scale);
    c.zero = pick(in, a.zero[i], c.zero);
    c.bias = pick(in, a.bias[i], c.bias);
    c.y = pick(in, a.y[i], c.y);
    c.N = pick(in, a.N[i], c.N);
    c.step = pick(in, a.step[i], c.step);
    c.e0 = pick(in, a.e_end[i - 1], c.e0);
    c.end = pick(in, a.e_end[i], c.end);
    c.li = pick(in, i, c.li);
  }
  return c;
}

// Which output row an entry of an XCD's stream is.  Ten output rows — one per slab — read (almost) the same packed rows: row n of
// slab t starts at packed row n*G - t*step.  With rows handed out in index order those ten readers sit ~N/10 rows apart, on
// different XCDs, and every XCD pulls the words through the fabric for itself: v1 ran at 0.85 TB/s of packed bytes = 8.5 TB/s of
// fabric traffic.  Here entry e of XCD x (workgroup b runs on XCD b % 8 — observed, a speed assumption only) is slab t = e % 10
// of "p-block" i = 4 * (8 * (q / 4) + x) + q % 4 with q = e / 10, i.e. output row n = i + ceil(t*step / G): the ten readers of a p-block are consecutive
// entries of ONE XCD's stream, taken by neighbouring waves at the same time, and nine of the ten reads hit that XCD's L2.
// Returns -1 for the few (i, t) past the end of a slab.
// `start` = the layer's table in LDS: start[t] = first output row whose groups begin in slab t = ceil(t*step / G), start[10] = N
__device__ __forceinline__ int g3_row_of(int e_local, int xcd, const int* start) {
  const int q = e_local / 10, t = e_local - 10 * q;
  const int n = (((q >> 2) * 8 + xcd) << 2) + (q & 3) + start[t];   // p-blocks go to XCDs four at a time: neighbours overlap by a row's worth
  return n < start[t + 1] ? n : -1;
}

__device__ __forceinline__ float g3_wave_sum(float v) {
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

struct G3Unit {
  u32x4 w[G3_U];          // words 4i..4i+3 of group (16*unit + 4u + j)
  uint32_t sh[G3_U];      // bit position 27 - 3*slab of that group
  uint32_t z, sc;         // lanes 0..15: zero / scale of group 16*unit + lane (raw fp16 bits, zero-extended: a uint16_t carried round
                          // the loop is masked — and waited for — in front of the next unit's requests)
};

template <int M>
__global__ __launch_bounds__(G3_WAVES * 64) void gemv3_f16_kernel(const G3Args a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];   // x[M][Kpad] fp16, natural order, zero padded to 1024-k units

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lj = lane >> 4;
  const int K = a.K, G = a.G;
  const int nunits = (G + 15) >> 4;
  const int kpad = nunits * 1024;
  const int xcd = blockIdx.x & 7;                       // see g3_row_of
  const int stride = (gridDim.x >> 3) * G3_WAVES;       // entries of this XCD's stream taken per sweep of its waves
  const int total = a.total_e;

  // ---- slab start table of every layer (11 entries each), once per workgroup ----
  int* start_tab = reinterpret_cast<int*>(smem + static_cast<size_t>(M) * kpad * 2);
  if (tid < G3_MAXL * 11) {
    const int l = tid / 11, t = tid - 11 * l;
    // (the layer's fields by selects over scalar loads: indexing the argument arrays with a per-lane l made two dependent VECTOR
    //  loads from the kernel-argument segment — two memory round trips in front of everything else in the kernel)
    int step_l = a.step[0], N_l = a.N[0];
#pragma unroll
    for (int i = 1; i < G3_MAXL; ++i) {
      step_l = pick(l == i, a.step[i], step_l);
      N_l = pick(l == i, a.N[i], N_l);
    }
    const int st = (t * step_l + G - 1) / G;             // (zero-padded rows at the end of the last slabs: clamp to N)
    start_tab[l * 12 + t] = (t == 10 || st > N_l) ? N_l : st;
  }
  // ---- stage x (natural k order) ----
  for (int v = tid; v < M * (kpad >> 3); v += G3_WAVES * 64) {
    const int m = v / (kpad >> 3), j = v - m * (kpad >> 3);
    u32x4 val = {0u, 0u, 0u, 0u};
    if (j * 8 < K) val = *reinterpret_cast<const u32x4*>(a.x + static_cast<int64_t>(m) * K + j * 8);
    *reinterpret_cast<u32x4*>(smem + (static_cast<size_t>(m) * kpad + j * 8) * 2) = val;
  }

  // r0 = first group row of output row `row`; groups past G (last unit of a row) re-read group 0 and meet zero x
  // (buffer loads: the layer's base pointer in a wave-uniform descriptor + one 32-bit byte offset per lane — no 64-bit VALU address
  //  arithmetic and fewer address temporaries, which the register allocator otherwise parks in the previous unit's load destinations,
  //  forcing a wait for that unit's data in front of the next requests)
  auto issue = [&](G3Unit& un, const G3Layer& ly, int row, int unit, bool live) {
    const __amdgpu_buffer_rsrc_t rw = buffer_rsrc(ly.Wq), rz = buffer_rsrc(ly.zero), rs = buffer_rsrc(ly.scale);
    int n = g3_row_of(row - ly.e0, xcd, start_tab + ly.li * 12);
    n = n < 0 ? 0 : n;                                 // an entry without a row: same loads on row 0, nothing stored
    const int r0 = n * G;
    const int s0 = r0 / ly.step;                       // slab of the row's first group (wave-uniform)
    const int bound = (s0 + 1) * ly.step;              // first group row of the next slab
#pragma unroll
    for (int u = 0; u < G3_U; ++u) {
      int g = unit * 16 + u * 4 + lj;
      g = g < G ? g : 0;
      const int r = r0 + g;
      const int s = s0 + (r >= bound ? 1 : 0);         // a row spans at most two slabs (G <= step)
      const int p = r - s * ly.step;
      un.sh[u] = 27 - 3 * s;
      un.w[u] = __builtin_amdgcn_raw_buffer_load_b128(rw, live ? p * 256 + li * 16 : 0, 0, 0);   // dead unit: one line, the same for every wave
    }
    int gm = unit * 16 + (lane & 15);
    gm = gm < G ? gm : 0;
    un.z = __builtin_amdgcn_raw_buffer_load_b16(rz, live ? (r0 + gm) * 2 : 0, 0, 0);
    un.sc = __builtin_amdgcn_raw_buffer_load_b16(rs, live ? (r0 + gm) * 2 : 0, 0, 0);
  };

  int row = (blockIdx.x >> 3) * G3_WAVES + wave;         // entry index in this XCD's stream (not an output row)
  int unit = 0;
  const bool live0 = row < total;
  row = live0 ? row : total - 1;                         // waves without an entry request the last one (uniform load counts) and leave
  G3Layer ly = g3_select(a, row);
  G3Unit ua, ub;
  __syncthreads();                                       // x and the start tables are in LDS
  issue(ua, ly, row, 0, live0);

  uint32_t magic;
  asm volatile("v_mov_b32 %0, 0x64006400" : "=v"(magic));
  f32x4 acc[M];
#pragma unroll
  for (int m = 0; m < M; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};

  // four exactly-dequantised weights (one lane's words of one load) as two fp16 pairs in natural k order
  auto deq4 = [&](const u32x4& w, uint32_t sh, uint32_t zs, uint32_t (&o)[2]) {
    const half2_t pr = __builtin_bit_cast(half2_t, zs);
    const half2_t zz = {pr.x, pr.x}, ss = {pr.y, pr.y};
    const half2_t k1024 = {static_cast<half_t>(1024.0f), static_cast<half_t>(1024.0f)};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const uint32_t t0 = w[2 * h] >> sh, t1 = w[2 * h + 1] >> sh;
      const uint32_t pk = __builtin_amdgcn_perm(t1, t0, 0x0C040C00u);   // byte0 <- t0.b0, byte2 <- t1.b0, others 0
      uint32_t b;
      asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(b) : "v"(pk), "s"(0x00070007u), "v"(magic));
      const half2_t q = __builtin_bit_cast(half2_t, b) - k1024;           // exact level
      o[h] = __builtin_bit_cast(uint32_t, (q - zz) * ss);                 // two roundings, as Quantizer.dequantize
    }
  };

  auto consume = [&](const G3Unit& cur, int orow, int unit) {
    const uint32_t mine = cur.z | (cur.sc << 16);
#pragma unroll
    for (int u = 0; u < G3_U; u += 2) {
      // loads u and u+1 -> one MFMA: this lane's k-octet = 4 values of group (4u+j) and 4 of group (4u+4+j)
      const uint32_t zs0 = __builtin_amdgcn_ds_bpermute((u * 4 + lj) << 2, mine);
      const uint32_t zs1 = __builtin_amdgcn_ds_bpermute(((u + 1) * 4 + lj) << 2, mine);
      uint32_t o0[2], o1[2];
      deq4(cur.w[u], cur.sh[u], zs0, o0);
      deq4(cur.w[u + 1], cur.sh[u + 1], zs1, o1);
      const g3_h8_t A = __builtin_bit_cast(g3_h8_t, u32x4{o0[0], o0[1], o1[0], o1[1]});
      const int k0 = (unit * 16 + u * 4 + lj) * 64 + li * 4;   // past K: zero-padded x
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const u32x2 xa = *reinterpret_cast<const u32x2*>(smem + (static_cast<size_t>(m) * kpad + k0) * 2);
        const u32x2 xb = *reinterpret_cast<const u32x2*>(smem + (static_cast<size_t>(m) * kpad + k0 + 256) * 2);
        const g3_h8_t B = __builtin_bit_cast(g3_h8_t, u32x4{xa.x, xa.y, xb.x, xb.y});
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, acc[m], 0, 0, 0);
      }
    }
    if (unit == nunits - 1) {   // row finished (its layer is looked up again: the issuing side may have moved on)
      const G3Layer oly = g3_select(a, orow);
      const int n = g3_row_of(orow - oly.e0, xcd, start_tab + oly.li * 12);
      float mine_out = 0.f;
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const float v = diag_sum(acc[m]);   // hqq_common.h: same association as the masked wave sum it replaces
        acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        mine_out = lane == m ? v : mine_out;
      }
      if (lane < M && n >= 0) {
        half_t o = static_cast<half_t>(mine_out);
        if (oly.bias) o = o + oly.bias[n];   // `out += bias` on the rounded matmul result (quantize.py:896-897)
        oly.y[static_cast<int64_t>(lane) * oly.N + n] = o;
      }
    }
  };

  auto advance = [&](int& r, int& u, G3Layer& c) {
    if (++u == nunits) {
      u = 0;
      r += stride;
      if (r >= c.end && r < total) c = g3_select(a, r);
    }
  };

  // one loop shape, ONE exit at the bottom: a unit past the wave's last is still requested (one cache line) and not consumed,
  // so every consume has exactly one unit's loads behind it and nothing waits in front of a request (gemv.hip has the story)
  if (live0) {
    bool more;
    do {
      int r1 = row, u1 = unit;
      advance(r1, u1, ly);
      const bool live1 = r1 < total;
      issue(ub, ly, live1 ? r1 : row, live1 ? u1 : unit, live1);
      consume(ua, row, unit);
      int r2 = r1, u2 = u1;
      advance(r2, u2, ly);
      more = r2 < total;   // (r1 >= total implies r2 >= total)
      issue(ua, ly, more ? r2 : (live1 ? r1 : row), more ? u2 : unit, more);
      if (live1) consume(ub, r1, u1);
      row = r2;
      unit = u2;
    } while (more);
  }
}

template <int M>
static int g3_launch(const G3Args& a, hipStream_t st) {
  const int nunits = (a.G + 15) >> 4;
  const size_t lds = static_cast<size_t>(M) * nunits * 1024 * 2 + G3_MAXL * 12 * sizeof(int);
  if (lds > 144 * 1024) { set_error("hqq_hip_gemv: x[M=%d, K=%d] does not fit the LDS staging budget", M, a.K); return HQQ_ERR_UNSUPPORTED; }
  int n_cus = 256, dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cus <= 0) n_cus = 256;
  int per_cu = static_cast<int>(160 * 1024 / (lds + 256));
  per_cu = per_cu > 4 ? 4 : (per_cu < 1 ? 1 : per_cu);
  const int tiles = 8 * ((a.total_e + G3_WAVES - 1) / G3_WAVES);   // per XCD: one wave per entry of its stream
  const int cap = (n_cus * per_cu) & ~7;
  auto kern = gemv3_f16_kernel<M>;
  if (lds > 64 * 1024) {
    static LdsRaised raised;
    if (const int rc = raise_lds_limit(raised, reinterpret_cast<const void*>(kern), 144 * 1024, "hqq_hip_gemv")) return rc;
  }
  hipLaunchKernelGGL(kern, dim3(tiles < cap ? tiles : cap), dim3(G3_WAVES * 64), lds, st, a);
  return check_launch("hqq_hip_gemv(3-bit)");
}

bool gemv3s_covers(int64_t M, int64_t K, int64_t group_size);   // gemv3s.hip: each packed word loaded once for all ten slabs
size_t gemv3s_workspace_bytes(int n_layers, const int64_t* N, int64_t M, int64_t K);
int gemv3s_run(int n_layers, const void* x, const void* const* Wq, const void* const* scale, const void* const* zero,
               const void* const* bias, void* const* y, const int64_t* N, int64_t M, int64_t K, uint32_t opts, void* ws, size_t ws_bytes, hipStream_t st);

// which of the two 3-bit kernels a launch takes: the slab-sharing one (gemv3s.hip) from 19 MB of packed weights on — 2.2 TB/s of
// packed bytes at the margin against 1.2 here, but ~10 us of fixed cost (a task is a 4 us chain of instructions in one wave, plus
// the finishing launch) against ~4 us; measured crossover on MI355X (tools/sweep_int3.py).  HQQ_OPT_GEMV3_ROWWISE / _SLABS force one.
static bool gemv3_wants_slabs(int n_layers, const int64_t* N, int64_t M, int64_t K, int64_t group_size, uint32_t opts) {
  if (group_size != 64 || !gemv3s_covers(M, K, group_size)) return false;
  int64_t packed = 0;
  for (int i = 0; i < n_layers; ++i) packed += ((N[i] * (K / 64) + 9) / 10) * 256;
  return (opts & HQQ_OPT_GEMV3_SLABS) || (packed >= (int64_t(19) << 20) && !(opts & HQQ_OPT_GEMV3_ROWWISE));
}
size_t gemv3_workspace_bytes(int n_layers, const int64_t* N, int64_t M, int64_t K, int64_t group_size, uint32_t opts) {
  return gemv3_wants_slabs(n_layers, N, M, K, group_size, opts) ? gemv3s_workspace_bytes(n_layers, N, M, K) : 0;
}

// called by hqq_hip_gemv_grouped (gemv.hip) for nbits == 3 after the common argument checks
int gemv3_run(int n_layers, const void* x, const void* const* Wq, const void* const* scale, const void* const* zero,
              const void* const* bias, void* const* y, const int64_t* N, int64_t M, int64_t K, int64_t group_size, uint32_t opts,
              void* ws, size_t ws_bytes, hipStream_t st) {
  if (group_size != 64) { set_error("hqq_hip_gemv: the fused 3-bit kernel covers group_size 64 (got %lld)", (long long)group_size); return HQQ_ERR_UNSUPPORTED; }
  if (M > G3_MAX_M) { set_error("hqq_hip_gemv: the fused 3-bit kernel covers M <= %d (got %lld)", G3_MAX_M, (long long)M); return HQQ_ERR_UNSUPPORTED; }
  G3Args a;
  const int64_t G = K / 64;
  int64_t ents = 0;
  for (int i = 0; i < n_layers; ++i) {
    if (N[i] <= 0) { set_error("hqq_hip_gemv: bad N"); return HQQ_ERR_SHAPE; }
    if (!Wq[i] || !scale[i] || !zero[i] || !y[i]) { set_error("hqq_hip_gemv: null layer pointer"); return HQQ_ERR_SHAPE; }
    if (!aligned16(Wq[i])) { set_error("hqq_hip_gemv: pointers must be 16-byte aligned"); return HQQ_ERR_ALIGN; }
    const int64_t R = N[i] * G;
    if (R > INT32_MAX / 2) { set_error("hqq_hip_gemv: size overflow"); return HQQ_ERR_SHAPE; }
    a.Wq[i] = static_cast<const int32_t*>(Wq[i]);
    a.scale[i] = static_cast<const half_t*>(scale[i]);
    a.zero[i] = static_cast<const half_t*>(zero[i]);
    a.bias[i] = bias ? static_cast<const half_t*>(bias[i]) : nullptr;
    a.y[i] = static_cast<half_t*>(y[i]);
    a.N[i] = static_cast<int>(N[i]);
    a.step[i] = static_cast<int>((R + 9) / 10);
    {   // entries per XCD stream: 10 slabs x 4 p-blocks x ceil(max rows starting in one slab / 32)
      const int64_t st = a.step[i];
      int64_t most = 0;
      for (int t = 0; t < 10; ++t) {
        int64_t start = (t * st + G - 1) / G, stop = t == 9 ? N[i] : ((t + 1) * st + G - 1) / G;
        start = start > N[i] ? N[i] : start;
        stop = stop > N[i] ? N[i] : stop;
        most = stop - start > most ? stop - start : most;
      }
      ents += 10 * 4 * ((most + 31) / 32);
      if (ents > INT32_MAX / 2) { set_error("hqq_hip_gemv: size overflow"); return HQQ_ERR_SHAPE; }
    }
    a.e_end[i] = static_cast<int>(ents);
    if (G > a.step[i]) { set_error("hqq_hip_gemv: 3-bit layer with fewer than 10 output rows per slab is not covered"); return HQQ_ERR_UNSUPPORTED; }
  }
  if (gemv3_wants_slabs(n_layers, N, M, K, group_size, opts)) return gemv3s_run(n_layers, x, Wq, scale, zero, bias, y, N, M, K, opts, ws, ws_bytes, st);
  for (int i = n_layers; i < G3_MAXL; ++i) {
    a.Wq[i] = a.Wq[n_layers - 1]; a.scale[i] = a.scale[n_layers - 1]; a.zero[i] = a.zero[n_layers - 1]; a.bias[i] = a.bias[n_layers - 1];
    a.y[i] = a.y[n_layers - 1]; a.N[i] = a.N[n_layers - 1]; a.step[i] = a.step[n_layers - 1]; a.e_end[i] = a.e_end[n_layers - 1];
  }
  a.x = static_cast<const half_t*>(x);
  a.K = static_cast<int>(K);
  a.G = static_cast<int>(G);
  a.total_e = static_cast<int>(ents);
  switch (M) {
    case 1: return g3_launch<1>(a, st);
    case 2: return g3_launch<2>(a, st);
    case 3: return g3_launch<3>(a, st);
    case 4: return g3_launch<4>(a, st);
  }
  return HQQ_ERR_SHAPE;
}

}  // namespace hqq
