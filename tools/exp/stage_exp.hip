// EXPERIMENT (tools/archive/stage_ab.py): how fast can 256 persistent workgroups stage a GEMM's operand tiles L2 -> LDS by LDS-DMA with NO compute,
// walking the tiles exactly like gemm4x (256 x 256 tiles, same supertile order, 4 waves, 128 KB ring)?
//   MODE 0: K-32 steps, 16 rows x 64 B per instruction (gemm4x's staging shape), WAIT instructions may stay in flight per wave
//   MODE 1: K-64 steps,  8 rows x 128 B per instruction (full cache lines)
// If MODE 0 takes about as long as gemm4x itself, the GEMM is staging-bound and MODE 1 says what full-line requests would buy.
#include "../../dove_amd/csrc/common.h"

typedef __attribute__((address_space(3))) void* lds_ptr_t;
struct Tile { int m0, n0; };
__device__ __forceinline__ Tile decode(int id, int ntiles, int tiles_n) {
  const unsigned rest = xcd_remap((unsigned)id, (unsigned)ntiles);
  const unsigned GM = tiles_n > 16 ? 8u : 1u;
  const unsigned tiles_m = (unsigned)(ntiles / tiles_n);
  const unsigned per_group = GM * (unsigned)tiles_n;
  const unsigned group = rest / per_group, within = rest - group * per_group;
  const unsigned left = tiles_m - group * GM;
  const unsigned gm = left < GM ? left : GM;
  Tile q;
  q.m0 = __builtin_amdgcn_readfirstlane((int)(group * GM + within % gm) * 256);
  q.n0 = __builtin_amdgcn_readfirstlane((int)(within / gm) * 256);
  return q;
}

extern __shared__ __attribute__((aligned(16))) char smem[];
template <int MODE, int WAIT>
__device__ __forceinline__ void stage_body(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, long long M, int K, int N) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NJ = MODE == 0 ? 4 : 8;                 // instructions per operand, wave and step
  constexpr int STEPB = MODE == 0 ? 64 : 128;           // bytes of K per row and step
  constexpr int SLOTB = 256 * STEPB * 2;                // x + w
  constexpr int NSLOT = 131072 / SLOTB;
  unsigned voff[NJ];
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj) {
    const int row = MODE == 0 ? (wave * 4 + jj) * 16 + (lane >> 2) : (wave * 8 + jj) * 8 + (lane >> 3);
    const int c = MODE == 0 ? (lane & 3) : (lane & 7);
    voff[jj] = (unsigned)((row * K + c * 8) * 2);
  }
  const int tiles_n = N / 256;
  const int ntiles = (int)((M + 255) / 256) * tiles_n;
  const int nsteps = K * 2 / STEPB;
  int slot = 0;
  for (int tile = (int)blockIdx.x; tile < ntiles; tile += (int)gridDim.x) {
    const Tile q = decode(tile, ntiles, tiles_n);
    const long long left = M - q.m0;
    const int rows = left < 256 ? (int)left : 256;
    const auto srd_a = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (long long)q.m0 * K), (short)0, rows * K * 2, 0x00020000);
    const auto srd_w = __builtin_amdgcn_make_buffer_rsrc((void*)(w + (long long)q.n0 * K), (short)0, 256 * K * 2, 0x00020000);
    for (int s = 0; s < nsteps; ++s) {
      char* base = smem + slot * SLOTB + wave * (NJ * 1024);
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_a, (lds_ptr_t)(base + jj * 1024), 16, voff[jj], s * STEPB, 0, 0);
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w, (lds_ptr_t)(base + SLOTB / 2 + jj * 1024), 16, voff[jj], s * STEPB, 0, 0);
      if (WAIT == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (WAIT == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (WAIT == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (WAIT == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
      else if (WAIT == 48) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(56)" ::: "memory");
      slot = slot + 1 == NSLOT ? 0 : slot + 1;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// plain (non-template) kernels: hipcc drops the host stub of the templated form of this kernel without a diagnostic
#define KERNEL(NAME, MODE, WAIT) \
  __global__ __launch_bounds__(256, 1) void NAME(const bf16_t* x, const bf16_t* w, long long M, int K, int N) { stage_body<MODE, WAIT>(x, w, M, K, N); }
KERNEL(stage_k0, 0, 24)
KERNEL(stage_k1, 0, 56)
KERNEL(stage_k2, 1, 16)
KERNEL(stage_k3, 1, 0)
KERNEL(stage_k4, 1, 48)
KERNEL(stage_k5, 0, 8)

extern "C" int stage_exp(int variant, const void* x, const void* w, long long M, int K, int N, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const int ntiles = (int)((M + 255) / 256) * (N / 256);
  const int grid = ntiles < 256 ? ntiles : 256;
  void (*k[6])(const bf16_t*, const bf16_t*, long long, int, int) = {stage_k0, stage_k1, stage_k2, stage_k3, stage_k4, stage_k5};
  if (variant < 0 || variant > 5) return -1;
  (void)hipFuncSetAttribute((const void*)k[variant], hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipLaunchKernelGGL(k[variant], dim3(grid), dim3(256), 131072, s, (const bf16_t*)x, (const bf16_t*)w, M, K, N);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
