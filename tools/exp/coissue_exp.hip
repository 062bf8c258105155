// EXPERIMENT (tools/archive/coissue.py): do the matrix pipe and the VALU of ONE SIMD run concurrently when the two instruction streams
// come from two DIFFERENT waves?  Workgroup = 8 waves (two per SIMD): waves 0-3 issue only MFMAs (16 per body, two accumulators,
// operands N(0,1) from memory), waves 4-7 only the softmax VALU mix of one attention wave-tile (33 v_exp, 34 v_add, 16 v_max3, 8 v_max,
// 16 v_cvt_pk), no LDS, no barriers.  mode 1: only the MFMA waves work, 2: only the VALU waves, 3: both; 4 / 5: both streams in EVERY wave
// (4: the MFMA block then the VALU block; 5: interleaved 1 MFMA : ~7 VALU).  Wall time per mode answers it: max(t1, t2) or t1 + t2.
#include "../../dove_amd/csrc/common.h"

__device__ __forceinline__ void mfma_body(f32x16& a0, f32x16& a1, const bf16x8& x, const bf16x8& y) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, a1, 0, 0, 0);
  }
}
// the same 16 MFMAs, each followed by NOPS x `s_nop 7` (8 idle cycles each): the wave does not sit at the issue stage with an MFMA the
// busy matrix pipe cannot take yet
template <int NOPS>
__device__ __forceinline__ void mfma_body_paced(f32x16& a0, f32x16& a1, const bf16x8& x, const bf16x8& y) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int n = 0; n < NOPS; ++n) asm volatile("s_nop 7");
    __builtin_amdgcn_sched_barrier(0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, a1, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int n = 0; n < NOPS; ++n) asm volatile("s_nop 7");
    __builtin_amdgcn_sched_barrier(0);
  }
}
// 8 independent accumulators: no MFMA ever waits for a result
__device__ __forceinline__ void mfma_body_indep(f32x16 (&a)[8], const bf16x8& x, const bf16x8& y) {
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16((i & 1) ? y : x, (i & 1) ? x : y, a[i & 7], 0, 0, 0);
}
__device__ __forceinline__ void valu_body(float (&s)[32], float& acc, float& mx, uint32_t (&pk)[16]) {
  float m0 = s[0];
#pragma unroll
  for (int r = 0; r < 32; r += 2) m0 = fmaxf(fmaxf(m0, s[r]), s[r + 1]);
  mx = fmaxf(mx, m0);
  float ps = 0.f;
#pragma unroll
  for (int r = 0; r < 32; ++r) { const float p = __builtin_amdgcn_exp2f(s[r] - 4.0f); s[r] = p; ps += p; }
  acc += ps + __builtin_amdgcn_exp2f(mx * -0.001f);
#pragma unroll
  for (int r = 0; r < 16; ++r) pk[r] ^= pack_bf2(s[2 * r], s[2 * r + 1]);
#pragma unroll
  for (int r = 0; r < 32; ++r) s[r] = s[r] * 0.5f + 1.0f;      // keep the values bounded and data dependent (32 v_fma extra)
}

// ROLE: which waves issue the MFMAs: 0: waves 0-3, 1: even waves, 2: waves with bit 1 clear (0, 1, 4, 5)
template <int MODE, int ROLE = 0, int PRIO = 0>   // PRIO 1: s_setprio 3 for the VALU waves, 2: for the MFMA waves
__global__ __launch_bounds__(512, 1) void coissue_kernel(const bf16_t* __restrict__ ops, float* __restrict__ out, int iters, unsigned* __restrict__ hwid) {
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool mf = (ROLE == 0 || ROLE == 3) ? wave < 4 : (ROLE == 1 ? (wave & 1) == 0 : (wave & 2) == 0);   // ROLE 3: as 0, VALU waves idle (modes 20+)
  if (hwid && blockIdx.x < 4 && (tid & 63) == 0) hwid[blockIdx.x * 8 + wave] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
  bf16x8 x = *(const bf16x8*)(ops + (size_t)(blockIdx.x * 512 + tid) * 16), y = *(const bf16x8*)(ops + (size_t)(blockIdx.x * 512 + tid) * 16 + 8);
  f32x16 a0, a1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
  float s[32], acc = 0.f, mx = -1e30f;
  uint32_t pk[16];
#pragma unroll
  for (int r = 0; r < 32; ++r) s[r] = bf2f(ops[(size_t)tid * 32 + r]);
#pragma unroll
  for (int r = 0; r < 16; ++r) pk[r] = 0;
  if (PRIO == 1 && !mf) __builtin_amdgcn_s_setprio(3);
  if (PRIO == 2 && mf) __builtin_amdgcn_s_setprio(3);
  if (MODE >= 20) {                  // side by side like mode 3 with another MFMA stream: 20/21/22: paced with 1/2/3 x s_nop 7; 23: 8 accumulators
    if (mf) {
      if (MODE == 23) {
        f32x16 a[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = a0;
        for (int i = 0; i < iters; ++i) mfma_body_indep(a, x, y);
#pragma unroll
        for (int k = 0; k < 8; ++k) a1 += a[k];
      } else {
        for (int i = 0; i < iters; ++i) mfma_body_paced<MODE - 19>(a0, a1, x, y);
      }
    } else if (ROLE == 0) {
      for (int i = 0; i < iters; ++i) valu_body(s, acc, mx, pk);
    }
  } else if (MODE <= 3) {
    if (mf) {
      if (MODE & 1) for (int i = 0; i < iters; ++i) mfma_body(a0, a1, x, y);
    } else {
      if (MODE & 2) for (int i = 0; i < iters; ++i) valu_body(s, acc, mx, pk);
    }
  } else if (MODE == 4) {
    for (int i = 0; i < iters / 2; ++i) { mfma_body(a0, a1, x, y); valu_body(s, acc, mx, pk); }
  } else {
    for (int i = 0; i < iters / 2; ++i) {
      // same work, the compiler free to interleave (no barrier between the blocks: independent registers)
      mfma_body(a0, a1, x, y);
      valu_body(s, acc, mx, pk);
      asm volatile("" ::: "memory");
    }
  }
  float r = acc + mx;
#pragma unroll
  for (int i = 0; i < 16; ++i) r += a0[i] + a1[i] + __uint_as_float(pk[i] & 0x3f800000u);
#pragma unroll
  for (int i = 0; i < 32; ++i) r += s[i];
  out[(size_t)blockIdx.x * 512 + tid] = r;
}

extern "C" void dove_set_error(const char*, ...) {}
extern "C" int coissue(int mode, const void* ops, void* out, int iters, int blocks, void* hwid, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  switch (mode) {
    case 1: hipLaunchKernelGGL((coissue_kernel<1, 0>), dim3(blocks), dim3(512), 0, s, (const bf16_t*)ops, (float*)out, iters, (unsigned*)hwid); break;
    case 2: hipLaunchKernelGGL((coissue_kernel<2, 0>), dim3(blocks), dim3(512), 0, s, (const bf16_t*)ops, (float*)out, iters, (unsigned*)hwid); break;
    case 3: hipLaunchKernelGGL((coissue_kernel<3, 0>), dim3(blocks), dim3(512), 0, s, (const bf16_t*)ops, (float*)out, iters, (unsigned*)hwid); break;
    case 4: hipLaunchKernelGGL((coissue_kernel<4, 0>), dim3(blocks), dim3(512), 0, s, (const bf16_t*)ops, (float*)out, iters, (unsigned*)hwid); break;
    case 5: hipLaunchKernelGGL((coissue_kernel<5, 0>), dim3(blocks), dim3(512), 0, s, (const bf16_t*)ops, (float*)out, iters, (unsigned*)hwid); break;
    case 6: hipLaunchKernelGGL((coissue_kernel<3, 1>), dim3(blocks), dim3(512), 0, s, (const bf16_t*)ops, (float*)out, iters, (unsigned*)hwid); break;
    case 7: hipLaunchKernelGGL((coissue_kernel<3, 2>), dim3(blocks), dim3(512), 0, s, (const bf16_t*)ops, (float*)out, iters, (unsigned*)hwid); break;
    case 8: hipLaunchKernelGGL((coissue_kernel<1, 1>), dim3(blocks), dim3(512), 0, s, (const bf16_t*)ops, (float*)out, iters, (unsigned*)hwid); break;
    case 10: hipLaunchKernelGGL((coissue_kernel<3, 0, 1>), dim3(blocks), dim3(512), 0, s, (const bf16_t*)ops, (float*)out, iters, (unsigned*)hwid); break;
    case 11: hipLaunchKernelGGL((coissue_kernel<3, 0, 2>), dim3(blocks), dim3(512), 0, s, (const bf16_t*)ops, (float*)out, iters, (unsigned*)hwid); break;
    case 20: hipLaunchKernelGGL((coissue_kernel<20, 0>), dim3(blocks), dim3(512), 0, s, (const bf16_t*)ops, (float*)out, iters, (unsigned*)hwid); break;
    case 21: hipLaunchKernelGGL((coissue_kernel<21, 0>), dim3(blocks), dim3(512), 0, s, (const bf16_t*)ops, (float*)out, iters, (unsigned*)hwid); break;
    case 22: hipLaunchKernelGGL((coissue_kernel<22, 0>), dim3(blocks), dim3(512), 0, s, (const bf16_t*)ops, (float*)out, iters, (unsigned*)hwid); break;
    case 23: hipLaunchKernelGGL((coissue_kernel<23, 0>), dim3(blocks), dim3(512), 0, s, (const bf16_t*)ops, (float*)out, iters, (unsigned*)hwid); break;
    case 30: hipLaunchKernelGGL((coissue_kernel<20, 3>), dim3(blocks), dim3(512), 0, s, (const bf16_t*)ops, (float*)out, iters, (unsigned*)hwid); break;
    case 31: hipLaunchKernelGGL((coissue_kernel<21, 3>), dim3(blocks), dim3(512), 0, s, (const bf16_t*)ops, (float*)out, iters, (unsigned*)hwid); break;
    case 32: hipLaunchKernelGGL((coissue_kernel<22, 3>), dim3(blocks), dim3(512), 0, s, (const bf16_t*)ops, (float*)out, iters, (unsigned*)hwid); break;
    case 33: hipLaunchKernelGGL((coissue_kernel<23, 3>), dim3(blocks), dim3(512), 0, s, (const bf16_t*)ops, (float*)out, iters, (unsigned*)hwid); break;
    case 9: hipLaunchKernelGGL((coissue_kernel<2, 1>), dim3(blocks), dim3(512), 0, s, (const bf16_t*)ops, (float*)out, iters, (unsigned*)hwid); break;
    default: return -1;
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
