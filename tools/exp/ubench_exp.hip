// EXPERIMENT: per-SIMD instruction throughput on gfx950 with 1..4 waves per SIMD (what bounds the attention softmax).
// Each wave runs REPS iterations of an unrolled asm body and reports s_memtime ticks (wave 0 of block 0).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short bf16x8;

#define R4(x) x x x x
#define R8(x) R4(x) R4(x)
#define R16(x) R8(x) R8(x)
#define R32(x) R16(x) R16(x)

template <int MODE>
__global__ __launch_bounds__(1024) void ub_kernel(unsigned long long* out, int reps, float seed) {
  float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
  float b0 = seed * 0.5f, b1 = b0 + 1, b2 = b0 + 2, b3 = b0 + 3;
  f32x16 c0, c1, c2, c3;
  for (int i = 0; i < 16; ++i) { c0[i] = seed; c1[i] = seed; c2[i] = seed; c3[i] = seed; }
  bf16x8 fa, fb;
  for (int i = 0; i < 8; ++i) { fa[i] = (short)(0x3c00 + threadIdx.x); fb[i] = (short)(0x3c00 + i); }
  extern __shared__ char lds_pad[];                 // >= 96 KB requested at launch: exactly one block per CU
  if (seed == 123.f) lds_pad[threadIdx.x] = 1;
  __syncthreads();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
    if (MODE == 0) {          // 32 independent v_exp_f32
      asm volatile(R4("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if (MODE == 1) {   // 32 independent v_add_f32
      asm volatile(R4("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));
    } else if (MODE == 2) {   // 16 exp + 16 add interleaved
      asm volatile(R4("v_exp_f32 %0, %0\n v_add_f32 %4, %4, %8\n v_exp_f32 %1, %1\n v_add_f32 %5, %5, %8\n v_exp_f32 %2, %2\n v_add_f32 %6, %6, %8\n v_exp_f32 %3, %3\n v_add_f32 %7, %7, %8\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));
    } else if (MODE == 3) {   // 16 MFMA on 4 accumulators (2 x R4 of 2; an earlier label said 8 and halved the derived rate)
      asm volatile(R4("v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n v_mfma_f32_32x32x16_bf16 %1, %4, %5, %1\n")
                   R4("v_mfma_f32_32x32x16_bf16 %2, %4, %5, %2\n v_mfma_f32_32x32x16_bf16 %3, %4, %5, %3\n")
                   : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(fa), "v"(fb));
    } else if (MODE == 4) {   // 8 MFMA + 32 v_add (4 per MFMA)
      asm volatile(R4("v_mfma_f32_32x32x16_bf16 %0, %12, %13, %0\n v_add_f32 %4, %4, %14\n v_add_f32 %5, %5, %14\n v_add_f32 %6, %6, %14\n v_add_f32 %7, %7, %14\n"
                      "v_mfma_f32_32x32x16_bf16 %1, %12, %13, %1\n v_add_f32 %8, %8, %14\n v_add_f32 %9, %9, %14\n v_add_f32 %10, %10, %14\n v_add_f32 %11, %11, %14\n")
                   : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                   : "v"(fa), "v"(fb), "v"(b0));
    } else if (MODE == 5) {   // 8 MFMA + 16 v_exp (2 per MFMA)
      asm volatile(R4("v_mfma_f32_32x32x16_bf16 %0, %12, %13, %0\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n"
                      "v_mfma_f32_32x32x16_bf16 %1, %12, %13, %1\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n")
                   : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                   : "v"(fa), "v"(fb), "v"(b0));
    } else if (MODE == 6) {   // 8 MFMA + 16 exp + 32 add + 8 cvt_pk (the softmax mix per 8 MFMAs, diet version)
      asm volatile(R4("v_mfma_f32_32x32x16_bf16 %0, %12, %13, %0\n v_exp_f32 %4, %4\n v_add_f32 %8, %8, %14\n v_exp_f32 %5, %5\n v_add_f32 %9, %9, %14\n v_add_f32 %10, %10, %14\n v_cvt_pk_bf16_f32 %11, %4, %5\n v_add_f32 %8, %8, %14\n"
                      "v_mfma_f32_32x32x16_bf16 %1, %12, %13, %1\n v_exp_f32 %6, %6\n v_add_f32 %8, %8, %14\n v_exp_f32 %7, %7\n v_add_f32 %9, %9, %14\n v_add_f32 %10, %10, %14\n v_cvt_pk_bf16_f32 %11, %6, %7\n v_add_f32 %9, %9, %14\n")
                   : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                   : "v"(fa), "v"(fb), "v"(b0));
    } else if (MODE == 7) {   // 32 v_max3_f32
      asm volatile(R4("v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));
    } else if (MODE == 8) {   // 32 v_cvt_pk_bf16_f32
      asm volatile(R4("v_cvt_pk_bf16_f32 %0, %8, %9\n v_cvt_pk_bf16_f32 %1, %8, %9\n v_cvt_pk_bf16_f32 %2, %8, %9\n v_cvt_pk_bf16_f32 %3, %8, %9\n v_cvt_pk_bf16_f32 %4, %8, %9\n v_cvt_pk_bf16_f32 %5, %8, %9\n v_cvt_pk_bf16_f32 %6, %8, %9\n v_cvt_pk_bf16_f32 %7, %8, %9\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));
    } else if (MODE == 9) {   // 32 v_permlane32_swap
      asm volatile(R4("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n v_permlane32_swap_b32 %1, %2\n v_permlane32_swap_b32 %3, %4\n v_permlane32_swap_b32 %5, %6\n v_permlane32_swap_b32 %7, %0\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if (MODE == 10) {  // 16 v_pk_add_f32 (32 adds)
      typedef __attribute__((ext_vector_type(2))) float f2;
      f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, q = {b0, b1};
      asm volatile(R4("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n")
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));
      a0 = p0[0]; a1 = p0[1]; a2 = p1[0]; a3 = p1[1]; a4 = p2[0]; a5 = p2[1]; a6 = p3[0]; a7 = p3[1];
    } else if (MODE == 11) {  // 8 MFMA + 16 exp issued by the SAME wave but exp grouped after the MFMAs (serial phases)
      asm volatile(R4("v_mfma_f32_32x32x16_bf16 %0, %12, %13, %0\n v_mfma_f32_32x32x16_bf16 %1, %12, %13, %1\n")
                   R4("v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n")
                   : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                   : "v"(fa), "v"(fb), "v"(b0));
    } else if (MODE == 12) {  // 32 v_fma_f32
      asm volatile(R4("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  if (s == 12345.678f) out[1] = 1;                 // keep everything live
  if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = t1 - t0; out[2] = r1 - r0; }
}

extern "C" int ubench(int mode, int waves_per_simd, int reps, unsigned long long* out, void* stream) {
  dim3 grid(256), block(256 * waves_per_simd);
  hipStream_t s = (hipStream_t)stream;
#define L(M) case M: (void)hipFuncSetAttribute((const void*)ub_kernel<M>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304); \
  hipLaunchKernelGGL(ub_kernel<M>, grid, block, 98304, s, out, reps, 1.0f); break;
  switch (mode) { L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11) L(12) default: return -1; }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
