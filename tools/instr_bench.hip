// instr_bench.hip — VALU issue-rate probe on gfx950 (development aid).  hipcc --offload-arch=gfx950 -O3 tools/instr_bench.hip -o /tmp/instr_bench
// Each kernel runs ITER x 32 independent instances of ONE instruction per wave, 8 waves per CU on all CUs, and reports
// cycles per wave-instruction per SIMD (2 waves share a SIMD, so a 2-cycle instruction reads ~2.0 here).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define ITER 2000
#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define KERNEL(name, DECL, BODY)                                                        \
  __global__ __launch_bounds__(512) void name(uint32_t* out, uint32_t seed) {           \
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
    uint32_t b = seed * 0x9E3779B9u + threadIdx.x, c = seed ^ 0x3C003C00u;               \
    DECL                                                                                 \
    for (int i = 0; i < ITER; ++i) {                                                     \
      BODY BODY BODY BODY                                                                \
    }                                                                                    \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;  \
  }

#define ASM3(op, i) asm volatile(op " %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
#define ASM3D(op, i) asm volatile(op " %0, %0, %1" : "+v"(a##i) : "v"(b));
#define ASM4(op, i) asm volatile(op " %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));

#define B_PKFMA(i) ASM4("v_pk_fma_f16", i)
#define B_PKADD(i) ASM3D("v_pk_add_f16", i)
#define B_PKMUL(i) ASM3D("v_pk_mul_f16", i)
#define B_DOT2C(i) ASM3("v_dot2c_f32_f16", i)
#define B_DOT2(i) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(a##i) : "v"(b), "v"(c));
#define B_DOT2BF(i) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(a##i) : "v"(b), "v"(c));
#define B_ANDOR(i) ASM4("v_and_or_b32", i)
#define B_BFI(i) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a##i) : "v"(b), "v"(c));
#define B_AND(i) ASM3D("v_and_b32", i)
#define B_FMA32(i) ASM4("v_fma_f32", i)
#define B_PKFMA32(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p##i) : "v"(pb), "v"(pc));
#define B_PERM(i) ASM4("v_perm_b32", i)
#define B_LSHR(i) asm volatile("v_lshrrev_b32 %0, 8, %0" : "+v"(a##i));
#define B_CVTPK(i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define B_FMAMIX(i) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,1,0]" : "+v"(a##i) : "v"(b), "v"(c));
#define B_MADMIXLO(i) asm volatile("v_fma_mixlo_f16 %0, %1, %2, %0" : "+v"(a##i) : "v"(b), "v"(c));
#define B_CVTF16(i) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(a##i));
#define B_CVTU8(i) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(a##i));
#define B_BFE(i) asm volatile("v_bfe_u32 %0, %0, 4, 4" : "+v"(a##i));
#define B_MOVDPP(i) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1" : "+v"(a##i));

// the decode loop's own forms (round 6): constants in SGPRs / literals, op_sel and neg modifiers
#define B_PKFMA_S(i) asm volatile("v_pk_fma_f16 %0, %0, %1, %2 op_sel_hi:[1,0,0] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "+v"(a##i) : "s"(sc), "v"(c));
#define B_PKFMA_V(i) asm volatile("v_pk_fma_f16 %0, %0, %1, %2 op_sel_hi:[1,0,0] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "+v"(a##i) : "v"(b), "v"(c));
#define B_PKMUL_OS(i) asm volatile("v_pk_mul_f16 %0, %1, %0 op_sel:[1,0]" : "+v"(a##i) : "v"(c));
#define B_AND_LIT(i) asm volatile("v_and_b32 %0, 0xf000f0, %0" : "+v"(a##i));
#define B_AND_S(i) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a##i) : "s"(sc));
#define B_MUL16(i) asm volatile("v_mul_f16 %0, %1, %0" : "+v"(a##i) : "v"(c));
#define B_FMA16(i) asm volatile("v_fma_f16 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
KERNEL(k_pkfma_s, uint32_t sc = __builtin_amdgcn_readfirstlane(seed | 0x78007800u);, R8(B_PKFMA_S))
KERNEL(k_pkfma_v, , R8(B_PKFMA_V))
KERNEL(k_pkmul_os, , R8(B_PKMUL_OS))
#define B_PKMUL_OSH(i) asm volatile("v_pk_mul_f16 %0, %1, %0 op_sel_hi:[0,1]" : "+v"(a##i) : "v"(c));
#define B_PKFMA_OS2(i) asm volatile("v_pk_fma_f16 %0, %0, %1, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "+v"(a##i) : "s"(sc), "v"(c));
#define B_PKMUL_S(i) asm volatile("v_pk_mul_f16 %0, %1, %0 op_sel_hi:[0,1]" : "+v"(a##i) : "s"(sc));
KERNEL(k_pkmul_osh, , R8(B_PKMUL_OSH))
KERNEL(k_pkfma_os2, uint32_t sc = __builtin_amdgcn_readfirstlane(seed | 0x78007800u);, R8(B_PKFMA_OS2))
KERNEL(k_pkmul_s, uint32_t sc = __builtin_amdgcn_readfirstlane(seed | 0x3c003c00u);, R8(B_PKMUL_S))
KERNEL(k_and_lit, , R8(B_AND_LIT))
KERNEL(k_and_s, uint32_t sc = __builtin_amdgcn_readfirstlane(seed | 0x00f000f0u);, R8(B_AND_S))
KERNEL(k_mul16, , R8(B_MUL16))
KERNEL(k_fma16, , R8(B_FMA16))
// the rebuild's own mix: per dword 1 shift + 4 and + 4 pk_fma + 4 pk_mul (13 instructions), dependent as in the kernel
#define B_MIX(i) asm volatile("v_lshrrev_b32 %1, 8, %0\n v_and_b32 %2, 0xf000f0, %0\n v_and_b32 %3, 0xf000f0, %1\n v_and_b32 %4, 0xf000f, %0\n v_and_b32 %1, 0xf000f, %1\n" \
  "v_pk_fma_f16 %2, %2, %5, %6 op_sel_hi:[1,0,0] neg_lo:[0,0,1] neg_hi:[0,0,1]\n v_pk_fma_f16 %3, %3, %5, %6 op_sel_hi:[1,0,0] neg_lo:[0,0,1] neg_hi:[0,0,1]\n" \
  "v_pk_fma_f16 %4, %4, %5, %6 op_sel_hi:[1,0,0] neg_lo:[0,0,1] neg_hi:[0,0,1]\n v_pk_fma_f16 %1, %1, %5, %6 op_sel_hi:[1,0,0] neg_lo:[0,0,1] neg_hi:[0,0,1]\n" \
  "v_pk_mul_f16 %2, %6, %2 op_sel:[1,0]\n v_pk_mul_f16 %3, %6, %3 op_sel:[1,0]\n v_pk_mul_f16 %4, %6, %4 op_sel:[1,0]\n v_pk_mul_f16 %1, %6, %1 op_sel:[1,0]\n v_xor_b32 %0, %2, %3\n v_xor_b32 %0, %0, %4\n v_xor_b32 %0, %0, %1" \
  : "+v"(a##i), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "s"(sc), "v"(c));
KERNEL(k_mix, uint32_t sc = __builtin_amdgcn_readfirstlane(seed | 0x78007800u); uint32_t t0 = 0 ; uint32_t t1 = 0 ; uint32_t t2 = 0 ; uint32_t t3 = 0;, R8(B_MIX))
KERNEL(k_pkfma, , R8(B_PKFMA))
KERNEL(k_pkadd, , R8(B_PKADD))
KERNEL(k_pkmul, , R8(B_PKMUL))
KERNEL(k_dot2c, , R8(B_DOT2C))
KERNEL(k_dot2, , R8(B_DOT2))
KERNEL(k_dot2bf, , R8(B_DOT2BF))
KERNEL(k_andor, , R8(B_ANDOR))
KERNEL(k_bfi, , R8(B_BFI))
KERNEL(k_and, , R8(B_AND))
KERNEL(k_fma32, , R8(B_FMA32))
KERNEL(k_perm, , R8(B_PERM))
KERNEL(k_lshr, , R8(B_LSHR))
KERNEL(k_cvtpk, , R8(B_CVTPK))
KERNEL(k_fmamix, , R8(B_FMAMIX))
KERNEL(k_mixlo, , R8(B_MADMIXLO))
KERNEL(k_cvtf16, , R8(B_CVTF16))
KERNEL(k_cvtu8, , R8(B_CVTU8))
KERNEL(k_bfe, , R8(B_BFE))
KERNEL(k_dpp, , R8(B_MOVDPP))

__global__ __launch_bounds__(512) void k_pkfma32(uint32_t* out, uint32_t seed) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {1.f * threadIdx.x, 2.f}, p1 = p0 * 3.f, p2 = p0 * 5.f, p3 = p0 * 7.f, p4 = p0 * 9.f, p5 = p0 * 11.f, p6 = p0 * 13.f, p7 = p0 * 15.f;
  f2 pb = {1.0001f, 0.9999f}, pc = {1e-3f * seed, 2e-3f};
  for (int i = 0; i < ITER; ++i) { R8(B_PKFMA32) R8(B_PKFMA32) R8(B_PKFMA32) R8(B_PKFMA32) }
  f2 s = p0 + p1 + p2 + p3 + p4 + p5 + p6 + p7;
  out[blockIdx.x * blockDim.x + threadIdx.x] = __float_as_uint(s.x + s.y);
}

template <typename F>
static void run(const char* name, F kern, uint32_t* d, int cus, double ghz) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(cus), dim3(512), 0, 0, d, 1u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(cus), dim3(512), 0, 0, d, 2u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_simd = 2.0 * ITER * 32;   // 2 waves per SIMD
  printf("%-12s %8.3f ms   %.2f cycles/wave-instr/SIMD at %.2f GHz\n", name, ms, ms * 1e-3 * ghz * 1e9 / instr_per_simd, ghz);
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  const double ghz = p.clockRate * 1e-6;
  printf("%s  CUs=%d clock=%.2f GHz\n", p.name, cus, ghz);
  uint32_t* d; hipMalloc(&d, cus * 512 * 4);
  run("v_fma_f32", k_fma32, d, cus, ghz);
  run("v_pk_fma_f32", k_pkfma32, d, cus, ghz);
  run("v_pk_fma_f16", k_pkfma, d, cus, ghz);
  run("pk_fma sgpr+mods", k_pkfma_s, d, cus, ghz);
  run("pk_fma vgpr+mods", k_pkfma_v, d, cus, ghz);
  run("pk_mul op_sel", k_pkmul_os, d, cus, ghz);
  run("pk_mul op_sel_hi0", k_pkmul_osh, d, cus, ghz);
  run("pk_fma src2 hi->both", k_pkfma_os2, d, cus, ghz);
  run("pk_mul sgpr", k_pkmul_s, d, cus, ghz);
  run("v_and literal", k_and_lit, d, cus, ghz);
  run("v_and sgpr", k_and_s, d, cus, ghz);
  run("v_mul_f16", k_mul16, d, cus, ghz);
  run("v_fma_f16", k_fma16, d, cus, ghz);
  run("rebuild mix/16", k_mix, d, cus, ghz);
  run("v_pk_add_f16", k_pkadd, d, cus, ghz);
  run("v_pk_mul_f16", k_pkmul, d, cus, ghz);
  run("v_dot2c_f16", k_dot2c, d, cus, ghz);
  run("v_dot2_f16", k_dot2, d, cus, ghz);
  run("v_dot2_bf16", k_dot2bf, d, cus, ghz);
  run("v_and_or", k_andor, d, cus, ghz);
  run("v_bfi", k_bfi, d, cus, ghz);
  run("v_and", k_and, d, cus, ghz);
  run("v_perm", k_perm, d, cus, ghz);
  run("v_lshrrev", k_lshr, d, cus, ghz);
  run("v_cvt_pk_bf16", k_cvtpk, d, cus, ghz);
  run("v_fma_mix", k_fmamix, d, cus, ghz);
  run("v_fma_mixlo", k_mixlo, d, cus, ghz);
  run("v_cvt_f32_f16", k_cvtf16, d, cus, ghz);
  run("v_cvt_ubyte0", k_cvtu8, d, cus, ghz);
  run("v_bfe_u32", k_bfe, d, cus, ghz);
  run("v_add_dpp", k_dpp, d, cus, ghz);
  return 0;
}
