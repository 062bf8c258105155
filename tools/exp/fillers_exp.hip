// EXPERIMENT (tools/fillers.py; never loaded by dove_amd): how many single-issue VALU instructions hide behind one MFMA of a one-wave-per-SIMD
// stream, for the two MFMA shapes of this code base - the budget a consumer-side GroupNorm + SiLU fusion into conv3x3_halo4x (16x16x32 walk)
// would have to fit.  One workgroup per CU, 4 waves (one per SIMD), accumulators in AGPRs, operands constant registers; a body of 64
// MFMA slots (64 of 16x16x32 or 32 of 32x32x16: 1024 pipe cycles) with K fillers per 32 pipe cycles hand-placed behind the MFMAs (asm volatile: the order is the program's), fillers of three kinds: v_fma_f32 on rotating
// registers, a {unpack, fma, exp, add, rcp, mul}-like mix (2 trans per 7), and v_exp_f32 alone.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) unsigned short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int I> struct IC { static constexpr int value = I; };
template <int N, int I = 0, class F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(IC<I>{}); static_for<N, I + 1>(f); }
}
template <int SHAPE, int K, int KIND>
__global__ __launch_bounds__(256, 1) void fill_kernel(float* out, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (unsigned short)(0x3f80 + threadIdx.x + i); b[i] = (unsigned short)(0x3f00 + 3 * threadIdx.x + i); }
  float r[16];
  for (int i = 0; i < 16; ++i) r[i] = 1.0f + 0.001f * (float)(threadIdx.x + i);
  f32x4 acc4[16];
  f32x16 acc16[4];
  for (int i = 0; i < 16; ++i) acc4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc16[i][e] = 0.f;
  for (int it = 0; it < iters; ++it) {
    static_for<64>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      if constexpr (SHAPE == 16) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc4[m & 15]) : "v"(a), "v"(b));
      else if constexpr ((m & 1) == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc16[(m >> 1) & 3]) : "v"(a), "v"(b));
      constexpr int NF = SHAPE == 16 ? (K + (m & 1)) / 2 : ((m & 1) == 0 ? K : 0);      // K fillers per 32 MFMA cycles: per pair of 16x16x32, per 32x32x16
      static_for<NF>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr int i = (m * 3 + k) & 15, j = (i + 5) & 15, n = m * K + k;
        constexpr int kind = KIND == 1 ? (n % 7 == 2 ? 2 : (n % 7 == 4 ? 3 : 0)) : KIND;
        if constexpr (kind == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(r[j]), "v"(r[(j + 3) & 15]));
        if constexpr (kind == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
        if constexpr (kind == 3) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
      });
      __builtin_amdgcn_sched_barrier(0);
    });
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += r[i] + acc4[i][0] + acc16[i & 3][i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int SHAPE, int K, int KIND>
static float run(float* out, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((fill_kernel<SHAPE, K, KIND>), dim3(256), dim3(256), 0, 0, out, iters);
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((fill_kernel<SHAPE, K, KIND>), dim3(256), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  return best;
}
// ms per launch (best of 5) for K = 0, 1, 2, 3, 4, 5, 6, 8 fillers per 32 pipe cycles
extern "C" int fillers(int shape, int kind, float* res9, int iters) {
  float* out;
  hipMalloc((void**)&out, 256 * 256 * 4);
#define ROW(S, KD) { res9[0] = run<S, 0, KD>(out, iters); res9[1] = run<S, 1, KD>(out, iters); res9[2] = run<S, 2, KD>(out, iters); res9[3] = run<S, 3, KD>(out, iters); \
                     res9[4] = run<S, 4, KD>(out, iters); res9[5] = run<S, 5, KD>(out, iters); res9[6] = run<S, 6, KD>(out, iters); res9[7] = run<S, 8, KD>(out, iters); }
  if (shape == 16 && kind == 0) ROW(16, 0)
  else if (shape == 16 && kind == 1) ROW(16, 1)
  else if (shape == 16 && kind == 2) ROW(16, 2)
  else if (shape == 32 && kind == 0) ROW(32, 0)
  else if (shape == 32 && kind == 1) ROW(32, 1)
  else if (shape == 32 && kind == 2) ROW(32, 2)
  else return -1;
  hipFree(out);
  return 0;
}
