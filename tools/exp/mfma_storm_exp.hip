// EXPERIMENT (never loaded by dove_amd): what the MFMA pipes of an MI355X sustain when every SIMD runs nothing but
// v_mfma_f32_32x32x16_bf16 - as a function of the operand DATA (zeros / constants / N(0,1) / random bits; toggling costs power)
// and of the DUTY cycle (s_nop padding after every MFMA, one wave per SIMD).  A calibrated s_nop loop gives the shader clock.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short bf16x8;

#define NOP16 "s_nop 15\n"

template <int NOPS>
__global__ __launch_bounds__(NOPS == 25 ? 256 : 1024) void storm_kernel(const bf16x8* __restrict__ adata, const bf16x8* __restrict__ bdata, int reps,
                                                     unsigned long long* out, float* sink) {
  const int lane = threadIdx.x & 63;
  bf16x8 fa[4], fb[4], fn[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    fa[i] = adata[i * 64 + lane];
    fb[i] = bdata[i * 64 + lane];
#pragma unroll
    for (int e = 0; e < 8; ++e) fn[i][e] = (short)(fb[i][e] ^ (short)0x8000);      // -b: the accumulators stay bounded
  }
  f32x16 c0, c1, c2, c3;
  typedef __attribute__((ext_vector_type(4))) float f32x4;
  f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0, d2 = d0, d3 = d0;
  f32x4 e[16];
  f32x16 g[16];
  if (NOPS == 24) {
#pragma unroll
    for (int i = 0; i < 16; ++i) e[i] = d0;
  }
  if (NOPS == 25) {
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) g[i][j] = 0.f;
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; c2[i] = 0.f; c3[i] = 0.f; }
  extern __shared__ char lds_pad[];                 // 96 KB requested at launch: exactly one block per CU
  if (reps < 0) lds_pad[threadIdx.x] = 1;
  __syncthreads();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
#define M(c, a, b) "v_mfma_f32_32x32x16_bf16 %" #c ", %" #a ", %" #b ", %" #c "\n"
  for (int r = 0; r < reps; ++r) {
    if (NOPS == 0)
      asm volatile(M(0, 4, 8) M(1, 5, 9) M(2, 6, 10) M(3, 7, 11) M(0, 5, 12) M(1, 6, 13) M(2, 7, 14) M(3, 4, 15)
                   M(0, 6, 10) M(1, 7, 11) M(2, 4, 8) M(3, 5, 9) M(0, 7, 14) M(1, 4, 15) M(2, 5, 12) M(3, 6, 13)
                   : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
                   : "v"(fa[0]), "v"(fa[1]), "v"(fa[2]), "v"(fa[3]), "v"(fb[0]), "v"(fb[1]), "v"(fb[2]), "v"(fb[3]), "v"(fn[0]), "v"(fn[1]), "v"(fn[2]), "v"(fn[3]));
    else if (NOPS == 3)
#define P3 NOP16 NOP16 NOP16
      asm volatile(M(0, 4, 8) P3 M(1, 5, 9) P3 M(2, 6, 10) P3 M(3, 7, 11) P3 M(0, 5, 12) P3 M(1, 6, 13) P3 M(2, 7, 14) P3 M(3, 4, 15) P3
                   M(0, 6, 10) P3 M(1, 7, 11) P3 M(2, 4, 8) P3 M(3, 5, 9) P3 M(0, 7, 14) P3 M(1, 4, 15) P3 M(2, 5, 12) P3 M(3, 6, 13) P3
                   : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
                   : "v"(fa[0]), "v"(fa[1]), "v"(fa[2]), "v"(fa[3]), "v"(fb[0]), "v"(fb[1]), "v"(fb[2]), "v"(fb[3]), "v"(fn[0]), "v"(fn[1]), "v"(fn[2]), "v"(fn[3]));
    else if (NOPS == 6)
#define P6 P3 P3
      asm volatile(M(0, 4, 8) P6 M(1, 5, 9) P6 M(2, 6, 10) P6 M(3, 7, 11) P6 M(0, 5, 12) P6 M(1, 6, 13) P6 M(2, 7, 14) P6 M(3, 4, 15) P6
                   M(0, 6, 10) P6 M(1, 7, 11) P6 M(2, 4, 8) P6 M(3, 5, 9) P6 M(0, 7, 14) P6 M(1, 4, 15) P6 M(2, 5, 12) P6 M(3, 6, 13) P6
                   : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
                   : "v"(fa[0]), "v"(fa[1]), "v"(fa[2]), "v"(fa[3]), "v"(fb[0]), "v"(fb[1]), "v"(fb[2]), "v"(fb[3]), "v"(fn[0]), "v"(fn[1]), "v"(fn[2]), "v"(fn[3]));
    else if (NOPS == 11)   // dependency distance 1: sixteen MFMAs back to back on ONE accumulator
      asm volatile(M(0, 4, 8) M(0, 5, 9) M(0, 6, 10) M(0, 7, 11) M(0, 5, 12) M(0, 6, 13) M(0, 7, 14) M(0, 4, 15)
                   M(0, 6, 10) M(0, 7, 11) M(0, 4, 8) M(0, 5, 9) M(0, 7, 14) M(0, 4, 15) M(0, 5, 12) M(0, 6, 13)
                   : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
                   : "v"(fa[0]), "v"(fa[1]), "v"(fa[2]), "v"(fa[3]), "v"(fb[0]), "v"(fb[1]), "v"(fb[2]), "v"(fb[3]), "v"(fn[0]), "v"(fn[1]), "v"(fn[2]), "v"(fn[3]));
    else if (NOPS == 12)   // distance 2: two accumulators alternating
      asm volatile(M(0, 4, 8) M(1, 5, 9) M(0, 6, 10) M(1, 7, 11) M(0, 5, 12) M(1, 6, 13) M(0, 7, 14) M(1, 4, 15)
                   M(0, 6, 10) M(1, 7, 11) M(0, 4, 8) M(1, 5, 9) M(0, 7, 14) M(1, 4, 15) M(0, 5, 12) M(1, 6, 13)
                   : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
                   : "v"(fa[0]), "v"(fa[1]), "v"(fa[2]), "v"(fa[3]), "v"(fb[0]), "v"(fb[1]), "v"(fb[2]), "v"(fb[3]), "v"(fn[0]), "v"(fn[1]), "v"(fn[2]), "v"(fn[3]));
    else if (NOPS == 14)   // the attention kernel's order: chains of four on one accumulator, then four on the next
      asm volatile(M(0, 4, 8) M(0, 5, 9) M(0, 6, 10) M(0, 7, 11) M(1, 5, 12) M(1, 6, 13) M(1, 7, 14) M(1, 4, 15)
                   M(2, 6, 10) M(2, 7, 11) M(2, 4, 8) M(2, 5, 9) M(3, 7, 14) M(3, 4, 15) M(3, 5, 12) M(3, 6, 13)
                   : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
                   : "v"(fa[0]), "v"(fa[1]), "v"(fa[2]), "v"(fa[3]), "v"(fb[0]), "v"(fb[1]), "v"(fb[2]), "v"(fb[3]), "v"(fn[0]), "v"(fn[1]), "v"(fn[2]), "v"(fn[3]));
    else if (NOPS == 20)   // operand pattern of a 4 x 4 register tile walked ROW-MAJOR: A held for four MFMAs, B changes every MFMA, both change at a row end
      asm volatile(M(0, 4, 8) M(1, 4, 9) M(2, 4, 10) M(3, 4, 11) M(0, 5, 12) M(1, 5, 13) M(2, 5, 14) M(3, 5, 15)
                   M(0, 6, 8) M(1, 6, 9) M(2, 6, 10) M(3, 6, 11) M(0, 7, 12) M(1, 7, 13) M(2, 7, 14) M(3, 7, 15)
                   : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
                   : "v"(fa[0]), "v"(fa[1]), "v"(fa[2]), "v"(fa[3]), "v"(fb[0]), "v"(fb[1]), "v"(fb[2]), "v"(fb[3]), "v"(fn[0]), "v"(fn[1]), "v"(fn[2]), "v"(fn[3]));
    else if (NOPS == 21)   // the same sixteen products in SNAKE order: exactly one operand changes from one MFMA to the next
      asm volatile(M(0, 4, 8) M(1, 4, 9) M(2, 4, 10) M(3, 4, 11) M(3, 5, 15) M(2, 5, 14) M(1, 5, 13) M(0, 5, 12)
                   M(0, 6, 8) M(1, 6, 9) M(2, 6, 10) M(3, 6, 11) M(3, 7, 15) M(2, 7, 14) M(1, 7, 13) M(0, 7, 12)
                   : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
                   : "v"(fa[0]), "v"(fa[1]), "v"(fa[2]), "v"(fa[3]), "v"(fb[0]), "v"(fb[1]), "v"(fb[2]), "v"(fb[3]), "v"(fn[0]), "v"(fn[1]), "v"(fn[2]), "v"(fn[3]));
    else if (NOPS == 22)   // NO operand changes at all: the same (a, +-b) pair sixteen times (the accumulators still toggle)
      asm volatile(M(0, 4, 8) M(1, 4, 8) M(2, 4, 8) M(3, 4, 8) M(0, 4, 12) M(1, 4, 12) M(2, 4, 12) M(3, 4, 12)
                   M(0, 4, 8) M(1, 4, 8) M(2, 4, 8) M(3, 4, 8) M(0, 4, 12) M(1, 4, 12) M(2, 4, 12) M(3, 4, 12)
                   : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
                   : "v"(fa[0]), "v"(fa[1]), "v"(fa[2]), "v"(fa[3]), "v"(fb[0]), "v"(fb[1]), "v"(fb[2]), "v"(fb[3]), "v"(fn[0]), "v"(fn[1]), "v"(fn[2]), "v"(fn[3]));
    else if (NOPS == 23) { // the 16 x 16 x 32 shape (half the FLOPs per instruction, half the accumulator traffic and twice the operand traffic per MAC), mode 0's operand order
#define M16(c, a, b) "v_mfma_f32_16x16x32_bf16 %" #c ", %" #a ", %" #b ", %" #c "\n"
      asm volatile(M16(0, 4, 8) M16(1, 5, 9) M16(2, 6, 10) M16(3, 7, 11) M16(0, 5, 12) M16(1, 6, 13) M16(2, 7, 14) M16(3, 4, 15)
                   M16(0, 6, 10) M16(1, 7, 11) M16(2, 4, 8) M16(3, 5, 9) M16(0, 7, 14) M16(1, 4, 15) M16(2, 5, 12) M16(3, 6, 13)
                   : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3)
                   : "v"(fa[0]), "v"(fa[1]), "v"(fa[2]), "v"(fa[3]), "v"(fb[0]), "v"(fb[1]), "v"(fb[2]), "v"(fb[3]), "v"(fn[0]), "v"(fn[1]), "v"(fn[2]), "v"(fn[3]));
    }
    else if (NOPS == 24) { // 16 x 16 x 32, a 4 x 4 register tile ROW-MAJOR on sixteen accumulators (what a register-tiled kernel in this shape issues)
      asm volatile(M16(0, 16, 20) M16(1, 16, 21) M16(2, 16, 22) M16(3, 16, 23) M16(4, 17, 20) M16(5, 17, 21) M16(6, 17, 22) M16(7, 17, 23)
                   M16(8, 18, 20) M16(9, 18, 21) M16(10, 18, 22) M16(11, 18, 23) M16(12, 19, 20) M16(13, 19, 21) M16(14, 19, 22) M16(15, 19, 23)
                   M16(0, 16, 24) M16(1, 16, 25) M16(2, 16, 26) M16(3, 16, 27) M16(4, 17, 24) M16(5, 17, 25) M16(6, 17, 26) M16(7, 17, 27)
                   M16(8, 18, 24) M16(9, 18, 25) M16(10, 18, 26) M16(11, 18, 27) M16(12, 19, 24) M16(13, 19, 25) M16(14, 19, 26) M16(15, 19, 27)
                   : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]), "+v"(e[6]), "+v"(e[7]), "+v"(e[8]), "+v"(e[9]), "+v"(e[10]),
                     "+v"(e[11]), "+v"(e[12]), "+v"(e[13]), "+v"(e[14]), "+v"(e[15])
                   : "v"(fa[0]), "v"(fa[1]), "v"(fa[2]), "v"(fa[3]), "v"(fb[0]), "v"(fb[1]), "v"(fb[2]), "v"(fb[3]), "v"(fn[0]), "v"(fn[1]), "v"(fn[2]), "v"(fn[3]));
    }
    else if (NOPS == 25) { // 32 x 32 x 16, the same 4 x 4 tile ROW-MAJOR on sixteen accumulators (the conv kernel's form), 2 x 16 MFMAs per trip like mode 24
      asm volatile(M(0, 16, 20) M(1, 16, 21) M(2, 16, 22) M(3, 16, 23) M(4, 17, 20) M(5, 17, 21) M(6, 17, 22) M(7, 17, 23)
                   M(8, 18, 20) M(9, 18, 21) M(10, 18, 22) M(11, 18, 23) M(12, 19, 20) M(13, 19, 21) M(14, 19, 22) M(15, 19, 23)
                   M(0, 16, 24) M(1, 16, 25) M(2, 16, 26) M(3, 16, 27) M(4, 17, 24) M(5, 17, 25) M(6, 17, 26) M(7, 17, 27)
                   M(8, 18, 24) M(9, 18, 25) M(10, 18, 26) M(11, 18, 27) M(12, 19, 24) M(13, 19, 25) M(14, 19, 26) M(15, 19, 27)
                   : "+a"(g[0]), "+a"(g[1]), "+a"(g[2]), "+a"(g[3]), "+a"(g[4]), "+a"(g[5]), "+a"(g[6]), "+a"(g[7]), "+a"(g[8]), "+a"(g[9]), "+a"(g[10]),
                     "+a"(g[11]), "+a"(g[12]), "+a"(g[13]), "+a"(g[14]), "+a"(g[15])
                   : "v"(fa[0]), "v"(fa[1]), "v"(fa[2]), "v"(fa[3]), "v"(fb[0]), "v"(fb[1]), "v"(fb[2]), "v"(fb[3]), "v"(fn[0]), "v"(fn[1]), "v"(fn[2]), "v"(fn[3]));
    }
    else  // NOPS == 99: no MFMA at all, 16 x 16 s_nop 15 = 4096 idle cycles per trip: the clock with the matrix pipes off
      asm volatile(P6 P6 P6 P6 P6 P6 P6 P6 P6 P6 P6 P6 P6 P6 P6 P6 P6 P6 P6 P6 P6 P6 P6 P6 P6 P6 P6 P6 P6 P6 P6 P6 P6 P6 P6 P6 P6 P6 P6 P6 P6 P6 NOP16 NOP16 NOP16 NOP16 ::: "memory");
  }
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  for (int i = 0; i < 4; ++i) s += d0[i] + d1[i] + d2[i] + d3[i];
  if (NOPS == 24) {
#pragma unroll
    for (int i = 0; i < 16; ++i) s += e[i][0] + e[i][3];
  }
  if (NOPS == 25) {
#pragma unroll
    for (int i = 0; i < 16; ++i) s += g[i][0] + g[i][15];
  }
  if (s == 12345.678f) sink[0] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = r1 - r0;      // 100 MHz ticks
}

extern "C" int mfma_storm(int nops, int waves_per_simd, int reps, const void* adata, const void* bdata, unsigned long long* out, float* sink,
                          void* stream) {
  dim3 grid(256), block(256 * waves_per_simd);
  hipStream_t s = (hipStream_t)stream;
#define L(N) case N: (void)hipFuncSetAttribute((const void*)storm_kernel<N>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304); \
  hipLaunchKernelGGL(storm_kernel<N>, grid, block, 98304, s, (const bf16x8*)adata, (const bf16x8*)bdata, reps, out, sink); break;
  switch (nops) { L(0) L(3) L(6) L(11) L(12) L(14) L(20) L(21) L(22) L(23) L(24) L(25) L(99) default: return -1; }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
