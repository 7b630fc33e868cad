// w3s.h — the 3-bit STREAM layout of this build and its exact weight rebuild (gfx950).
//
// Why a second layout.  The reference's 3-bit container (BitPack.pack_3bit_32, hqq/core/bitpack.py:69-91) ORs ten row slabs of the
// [R, 64] level matrix into one int32, with step = ceil(R / 10) NOT a multiple of a row's groups: one word mixes ten unrelated
// output rows, and a kernel that streams it needs fp32 scratch partials and a second launch (gemv3s.hip).  Every optimised backend of
// the reference re-lays the levels out when a layer is patched (hqq/backends/torchao.py:202-241, marlin.py:74-123; contract: SURVEY.md
// section 8b path 2); so does HQQLinearHIP — Quantizer.quantize / HQQLinear.state_dict() keep the reference's bytes (hqq_hip_w3s_unpack
// restores them bit for bit).
//
// Layout (axis = 1, group_size 64, N even, K % 64 == 0): the 4-bit container's shape at 3 bits per level.
//   W3s [N/2, K/16, 3] uint32.  Packed row p holds output rows p (slab 0) and p + N/2 (slab 1); chunk c their k = 16 c .. 16 c + 15:
//   3 dwords = 96 bits = 32 levels.  A dword holds 5 PAIR fields: field f = bits [3f, 3f+3) (the pair's even k) and [16+3f, 16+3f+3)
//   (its odd k), so that one AND yields both levels of a pair in the two halves of a register (an fp16 pair after one more op).
//   Pair j of a slab = (k 2j, k 2j + 1), j = 0..7:
//     dword 0, fields 0..4 : slab 0 pairs 0..4          dword 1, fields 0..4 : slab 1 pairs 0..4
//     dword 2, fields 0..2 : slab 0 pairs 5..7          dword 2, fields 3..4 : slab 1 pairs 5, 6
//     slab 1 pair 7: bit i of its even-k level is bit 15 of dword i, bit i of its odd-k level bit 31 of dword i (i = 0, 1, 2)
//   Exactly 3 bits per level: N K 3/8 bytes (the reference container: 3.2 bits per level + padding).  A lane's 12 bytes are 16 k of two
//   rows — the unit every decode / GEMM kernel of this build works on for 4 bits — so 3-bit layers run through the same kernels.
//   scale / zero are untouched ([N K / 64], row n uses [n G, (n + 1) G)).
#pragma once
#include "decode_common.h"

namespace hqq {

constexpr int W3S_CHUNK_K = 16;       // k per chunk
constexpr int W3S_CHUNK_BYTES = 12;   // bytes per chunk (two rows x 16 levels)

// where level (slab s, i = k % 16) of a chunk lives: dword d, bit offset b of a 3-bit field; d < 0: the scattered pair (slab 1, pair 7)
struct W3sPos { int d, b; };
__host__ __device__ constexpr W3sPos w3s_pos(int s, int i) {
  const int j = i >> 1, h = (i & 1) * 16;
  if (s == 0) return j < 5 ? W3sPos{0, 3 * j + h} : W3sPos{2, 3 * (j - 5) + h};
  if (j < 5) return W3sPos{1, 3 * j + h};
  if (j < 7) return W3sPos{2, 3 * (j - 2) + h};
  return W3sPos{-1, 15 + h};
}
__host__ __device__ inline uint32_t w3s_get(const uint32_t (&D)[3], int s, int i) {
  const W3sPos q = w3s_pos(s, i);
  if (q.d >= 0) return (D[q.d] >> q.b) & 7u;
  return ((D[0] >> q.b) & 1u) | (((D[1] >> q.b) & 1u) << 1) | (((D[2] >> q.b) & 1u) << 2);
}
__host__ __device__ inline void w3s_put(uint32_t (&D)[3], int s, int i, uint32_t lv) {
  const W3sPos q = w3s_pos(s, i);
  if (q.d >= 0) { D[q.d] |= (lv & 7u) << q.b; return; }
  D[0] |= (lv & 1u) << q.b;
  D[1] |= ((lv >> 1) & 1u) << q.b;
  D[2] |= ((lv >> 2) & 1u) << q.b;
}

// ---- the pair fields of a chunk, as the kernels extract them: f[s][j] holds pair j of slab s as two 3-bit fields at bit offset
//      3 W3S_CLS[s][j] of the two halves (everything else zero): 3 shifts + 15 ANDs + 5 for the scattered pair = 23 VALU per 32 levels ----
constexpr int W3S_CLS[2][8] = {{0, 1, 2, 0, 1, 0, 1, 2}, {0, 1, 2, 0, 1, 0, 1, 0}};
__device__ __forceinline__ void w3s_fields(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t (&f)[2][8]) {
  constexpr uint32_t M0 = 0x00070007u, M1 = 0x00380038u, M2 = 0x01C001C0u;
  const uint32_t t0 = w0 >> 9, t1 = w1 >> 9, t2 = w2 >> 9;   // fields 3, 4 -> offsets 0, 3; bit 15 / 31 -> bit 6 / 22
  uint32_t sp = (t0 >> 6) & 0x00010001u;
  sp = ((t1 >> 5) & 0x00020002u) | sp;
  sp = ((t2 >> 4) & 0x00040004u) | sp;
  f[0][0] = w0 & M0; f[0][1] = w0 & M1; f[0][2] = w0 & M2; f[0][3] = t0 & M0; f[0][4] = t0 & M1;
  f[0][5] = w2 & M0; f[0][6] = w2 & M1; f[0][7] = w2 & M2;
  f[1][0] = w1 & M0; f[1][1] = w1 & M1; f[1][2] = w1 & M2; f[1][3] = t1 & M0; f[1][4] = t1 & M1;
  f[1][5] = t2 & M0; f[1][6] = t2 & M1; f[1][7] = sp;
}

// fp16: the lane's 16 weights of both slabs, rebuilt exactly as Quantizer.dequantize does (round16(round16(q - z) * s), quantize.py:198),
// as MFMA A operands in NATURAL k order: a0[s] = k 0..7 of the chunk, a1[s] = k 8..15 (x needs no permutation).
//   SUB (HQQ_OPT_META_SCALABLE, three ops per pair as decode_common.h's): a field at bit offset p read as fp16 IS the subnormal
//   q 2^(p-24); one v_pk_fma lifts it by 2^15 and subtracts z 2^-J in a single rounding (J = 9 - p: 9, 6 or 3), one v_pk_mul by s 2^J
//   rounds again.  Needs z 2^-9 exact and s 2^9 finite for every group (hqq_hip_meta_check with the stream-layout flag).
//   else: (field | 0x6400) = 1024 + q 2^p, one fma gives the exact level, then - z, * s.
// zs[s] = (zero | scale << 16) of the lane's group of slab s, raw.
template <bool SUB>
__device__ __forceinline__ void w3s_rebuild_f16(uint32_t w0, uint32_t w1, uint32_t w2, const uint32_t (&zs)[2], uint32_t magic, h8_t (&a0)[2], h8_t (&a1)[2]) {
  uint32_t f[2][8];
  w3s_fields(w0, w1, w2, f);
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    uint32_t o[8];
    if constexpr (SUB) {
      const half2_t lift = {static_cast<half_t>(32768.0f), static_cast<half_t>(32768.0f)};
      half2_t nz[3], ss[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int J = 9 - 3 * c;
        const half2_t fj = {static_cast<half_t>(1.0f / static_cast<float>(1 << J)), static_cast<half_t>(static_cast<float>(1 << J))};
        const half2_t pr = as_h2(zs[s]) * fj;   // (z 2^-J, s 2^J): exact for every group of a layer that passed the meta check
        nz[c] = half2_t{-pr.x, -pr.x};
        ss[c] = half2_t{pr.y, pr.y};
      }
      half2_t q[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) q[j] = __builtin_elementwise_fma(as_h2(f[s][j]), lift, nz[W3S_CLS[s][j]]);   // rounding 1
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = __builtin_bit_cast(uint32_t, q[j] * ss[W3S_CLS[s][j]]);               // rounding 2
    } else {
      const half2_t pr = as_h2(zs[s]);
      const half2_t zz = {pr.x, pr.x}, ss = {pr.y, pr.y};
      half2_t q[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int p = 3 * W3S_CLS[s][j];
        const float inv = 1.0f / static_cast<float>(1 << p);
        const half2_t k1 = {static_cast<half_t>(inv), static_cast<half_t>(inv)};
        const half2_t k2 = {static_cast<half_t>(-1024.0f * inv), static_cast<half_t>(-1024.0f * inv)};
        q[j] = __builtin_elementwise_fma(as_h2(f[s][j] | magic), k1, k2);   // exact integer level
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) q[j] = q[j] - zz;                                  // rounding 1
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = __builtin_bit_cast(uint32_t, q[j] * ss);   // rounding 2
    }
    a0[s] = __builtin_bit_cast(h8_t, u32x4{o[0], o[1], o[2], o[3]});
    a1[s] = __builtin_bit_cast(h8_t, u32x4{o[4], o[5], o[6], o[7]});
  }
}

// bf16: the two roundings are to bf16 (quantize.py:198 on bf16 tensors); gfx950 has no packed bf16 arithmetic, so a weight goes through
// fp32 as in gemv.hip's SlabExactBF16: the masked field read as fp16 is the subnormal q 2^(p-24) (exact in fp32), one fma forms q - z
// with ONE fp32 rounding, v_cvt_pk_bf16_f32 rounds it (RNE), v_dot2_f32_bf16 against (s, 0) / (0, s) forms the exact product with s, a
// second v_cvt_pk rounds again.  zs[s] = (zero | scale << 16) as raw bf16 patterns.
typedef __bf16 w3s_bf2_t __attribute__((ext_vector_type(2)));
typedef __bf16 w3s_bf8_t __attribute__((ext_vector_type(8)));
typedef float w3s_f2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void w3s_rebuild_bf16(uint32_t w0, uint32_t w1, uint32_t w2, const uint32_t (&zs)[2], w3s_bf8_t (&a0)[2], w3s_bf8_t (&a1)[2]) {
  uint32_t f[2][8];
  w3s_fields(w0, w1, w2, f);
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const float zf = __uint_as_float(zs[s] << 16);
    const w3s_bf2_t s_lo = __builtin_bit_cast(w3s_bf2_t, zs[s] >> 16);          // (s, 0)
    const w3s_bf2_t s_hi = __builtin_bit_cast(w3s_bf2_t, zs[s] & 0xFFFF0000u);  // (0, s)
    uint32_t o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float up = static_cast<float>(1 << (24 - 3 * W3S_CLS[s][j]));
      const half2_t h = as_h2(f[s][j]);
      const w3s_f2_t dq = {__builtin_fmaf(static_cast<float>(h.x), up, -zf), __builtin_fmaf(static_cast<float>(h.y), up, -zf)};
      const w3s_bf2_t dr = __builtin_convertvector(dq, w3s_bf2_t);                 // rounding 1
      const w3s_f2_t pw = {__builtin_amdgcn_fdot2_f32_bf16(dr, s_lo, 0.f, false), __builtin_amdgcn_fdot2_f32_bf16(dr, s_hi, 0.f, false)};
      o[j] = __builtin_bit_cast(uint32_t, __builtin_convertvector(pw, w3s_bf2_t));   // rounding 2
    }
    a0[s] = __builtin_bit_cast(w3s_bf8_t, u32x4{o[0], o[1], o[2], o[3]});
    a1[s] = __builtin_bit_cast(w3s_bf8_t, u32x4{o[4], o[5], o[6], o[7]});
  }
}

// one 12-byte chunk (16 k of both slabs) against the lane's 16 x-values of M activation rows: the row-per-wave decode kernel's step
// (gemv_kernel.inc with NBITS = 3; acc[m][slab] as there)
template <int M, bool BF16, bool SUB>
struct SlabW3s {
  template <class FRAG>
  static __device__ __forceinline__ void run(const u32x4& w, const uint32_t (&zs)[2], const FRAG (&b0)[M], const FRAG (&b1)[M], f32x4 (&acc)[M][2], uint32_t magic) {
    if constexpr (BF16) {
      w3s_bf8_t a0[2], a1[2];
      w3s_rebuild_bf16(w.x, w.y, w.z, zs, a0, a1);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int m = 0; m < M; ++m) {
          acc[m][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[s], b0[m], acc[m][s], 0, 0, 0);
          acc[m][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[s], b1[m], acc[m][s], 0, 0, 0);
        }
    } else {
      h8_t a0[2], a1[2];
      w3s_rebuild_f16<SUB>(w.x, w.y, w.z, zs, magic, a0, a1);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int m = 0; m < M; ++m) {
          acc[m][s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[s], b0[m], acc[m][s], 0, 0, 0);
          acc[m][s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[s], b1[m], acc[m][s], 0, 0, 0);
        }
    }
  }
};

}  // namespace hqq
