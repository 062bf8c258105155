// EXPERIMENT (tools/archive/attn16_ab.py; never loaded by dove_amd; its constant-shift path ran once on a GPU with round 4's last seconds -
// profiles/r04_attn16.log: correct - the running-maximum path has not run yet): the flash-attention forward of dove_attention_fwd_bf16 on
// v_mfma_f32_16x16x32_bf16, the MFMA shape the power-limited matrix pipe sustains best on real operands (the conv and the
// GEMMs moved to it, bit-identical and 3-7 % faster).  Both softmax paths of the product kernel: the constant shift from a per-head score
// bound (what every head of the DiT takes) and the lazily updated running maximum (bound == nullptr or above the cutoff of 40); either way
// the shift rides in the C operand of the first MFMA of each S chain.
//
// Same workgroup / staging as the product kernel: 4 waves x 32 queries, 64-key K and V^T tiles by LDS-DMA into XOR-swizzled 128-B rows,
// four 16 KB stages, one barrier per two tiles.  What changes is the register tile:
//   S^T[key][query] = K Q^T as 4 key blocks x 2 query blocks of 16 x 16 (8 quads = the same 32 registers): lane = (query l15 of the block,
//     keys 4 q4 .. 4 q4 + 3 of the block); A operand = K rows (key l15, d chunk 4 kk + q4), B operand = Q (query l15, d chunk 4 kk + q4).
//   P^T as the B operand of O^T = V^T P^T: a K-32 group is TWO key blocks (2 g, 2 g + 1); lane q4 owns keys 4 q4 .. 4 q4 + 3 of each, and its
//     eight probabilities ARE its K slice in the order [block 2 g: 4 q4 ..+3 | block 2 g + 1: 4 q4 ..+3].  The contraction order is free as long
//     as V^T uses the same one, so the caller stores every 32 keys of V^T as [a0-3 b0-3 | a4-7 b4-7 | a8-11 b8-11 | a12-15 b12-15] (a / b = the
//     two 16-key blocks; tools/archive/attn16_ab.py permutes; in the product that would be a third v_order of dove_qkv_post_bf16) and lane q4's
//     fragment is ONE 16-byte chunk: no cross-lane exchange anywhere, like the product kernel.
//   O^T[d][query] as 4 d blocks x 2 query blocks (8 quads, the same 32 registers); lane = (query l15, d 4 q4 .. 4 q4 + 3).
//   Row sums: per lane and query block, reduced over the four q4 lane groups once at the end.
#include <stdlib.h>

#include <type_traits>

#include "../../dove_amd/csrc/common.h"

typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ bf16x8 make_frag16(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  u32x4 v = {a, b, c, d};
  return __builtin_bit_cast(bf16x8, v);
}
// D = A B + C with C in other registers than D (C = -bound broadcast): see mfma_c_in in attention.hip
__device__ __forceinline__ f32x4 mfma16_c_in(bf16x8 a, bf16x8 b, const f32x4& c) {
  f32x4 d;
  asm("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

__global__ __launch_bounds__(256, 2) void attn16_kernel(const bf16_t* __restrict__ Qh, const bf16_t* __restrict__ Kh, const bf16_t* __restrict__ Vt,
                                                        bf16_t* __restrict__ O, long long N, long long Npad, long long ldo, int qblocks,
                                                        const float* __restrict__ bound) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STAGE = 16384, VOFF = 8192;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q4 = lane >> 4, l15 = lane & 15;
  const unsigned t = xcd_remap(blockIdx.x, gridDim.x);
  const int h = (int)(t / (unsigned)qblocks), qb = (int)(t - (unsigned)h * (unsigned)qblocks);
  const long long q0 = (long long)qb * 128 + wave * 32;

  bf16x8 qf[2][2];                                            // [query block][kk]
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    long long qrow = q0 + b * 16 + l15;
    if (qrow >= Npad) qrow = Npad - 1;
    const bf16_t* qp = Qh + ((long long)h * Npad + qrow) * 64 + q4 * 8;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) qf[b][kk] = *(const bf16x8*)(qp + kk * 32);
  }
  constexpr float THR = 6.0f;                                 // running maximum: rescale when a score exceeds it by 2^6 (as the product kernel)
  f32x4 o[4][2], negm[2];
  float lsum[2] = {0.f, 0.f}, m[2] = {0.f, 0.f};
  bool fixed = false;
  negm[0] = negm[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (bound) {
    const float bnd = 1.01f * sqrtf(bound[2 * h] * bound[2 * h + 1]);
    fixed = bnd <= 40.0f;                                     // NaN compares false: the running maximum
    if (fixed) negm[0] = negm[1] = f32x4{-bnd, -bnd, -bnd, -bnd};
  }
  fixed = __builtin_amdgcn_readfirstlane(fixed);
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int b = 0; b < 2; ++b) o[db][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int ntiles = (int)((N + 63) / 64);
  const int srow = tid >> 3;
  const int sc_ld = (tid & 7) ^ ((srow >> 1) & 7);
  const auto srd_k = __builtin_amdgcn_make_buffer_rsrc((void*)(Kh + (long long)h * Npad * 64), (short)0, (int)(Npad * 128), 0x00020000);
  const auto srd_v = __builtin_amdgcn_make_buffer_rsrc((void*)(Vt + (long long)h * 64 * Npad), (short)0, (int)(Npad * 128), 0x00020000);
  unsigned vk[2], vv[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    vk[j] = (unsigned)(((j * 32 + srow) * 64 + sc_ld * 8) * 2);
    vv[j] = (unsigned)((((long long)(j * 32 + srow)) * Npad + sc_ld * 8) * 2);
  }
  auto stage = [&](auto bufc, int tile) {
    constexpr int BUF = decltype(bufc)::value;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_k, (lds_ptr_t)(smem + BUF * STAGE + (j * 256 + wave * 64) * 16), 16, vk[j], tile * (64 * 128), 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_v, (lds_ptr_t)(smem + BUF * STAGE + VOFF + (j * 256 + wave * 64) * 16), 16, vv[j], tile * (64 * 2), 0, 0);
    }
  };
  // fragment offsets: row 16 blk + l15 of a 64-row tile, 16-byte chunk 4 kk + q4 of its 128-B row; rows 16 apart share the swizzle term
  // (row >> 1) & 7, so a block is an immediate.  16 lanes x 4 chunks cover the 64 banks once (as in gemm8p's 16 x 16 x 32 reads).
  int foff[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) foff[kk] = l15 * 128 + (((kk * 4 + q4) ^ ((l15 >> 1) & 7)) << 4);

  auto compute = [&](auto bufc, int tile, auto fixc) {
    constexpr int BUF = decltype(bufc)::value;
    constexpr bool kFixed = decltype(fixc)::value;
    f32x4 st[4][2];                                           // [key block][query block]
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const bf16x8 kf = *(const bf16x8*)(smem + BUF * STAGE + foff[kk] + kb * 2048);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          if (kk == 0) st[kb][b] = mfma16_c_in(kf, qf[b][kk], negm[b]);
          else st[kb][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[b][kk], st[kb][b], 0, 0, 0);
        }
      }
    const long long kv0 = (long long)tile * 64;
    if (kv0 + 64 > N) {
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (kv0 + kb * 16 + 4 * q4 + e >= N) { st[kb][0][e] = -1e30f; st[kb][1][e] = -1e30f; }
    }
    if (!kFixed) {                                            // lazy online softmax, per query block (a lane owns one query of each)
      float mt[2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        mt[b] = st[0][b][0];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int e = 0; e < 4; ++e) mt[b] = fmaxf(mt[b], st[kb][b][e]);
        mt[b] = fmaxf(mt[b], __shfl_xor(mt[b], 16));          // the other three key quarters of the same query column
        mt[b] = fmaxf(mt[b], __shfl_xor(mt[b], 32));
      }
      const bool first = tile == 0;
      if (first || __any(fmaxf(mt[0], mt[1]) > THR)) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const float delta = first ? mt[b] : fmaxf(mt[b], 0.f);
          const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);
          m[b] += delta;
          lsum[b] *= alpha;
#pragma unroll
          for (int db = 0; db < 4; ++db) o[db][b] *= alpha;
#pragma unroll
          for (int kb = 0; kb < 4; ++kb) st[kb][b] -= delta;
          negm[b] = f32x4{-m[b], -m[b], -m[b], -m[b]};
        }
      }
    }
    float ps[2] = {0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float p = __builtin_amdgcn_exp2f(st[kb][b][e]);
          st[kb][b][e] = p;
          ps[b] += p;
        }
    lsum[0] += ps[0];
    lsum[1] += ps[1];
    bf16x8 pf[2][2];                                          // [K-32 group g = key blocks 2 g, 2 g + 1][query block]
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int b = 0; b < 2; ++b)
        pf[g][b] = make_frag16(pack_bf2(st[2 * g][b][0], st[2 * g][b][1]), pack_bf2(st[2 * g][b][2], st[2 * g][b][3]),
                               pack_bf2(st[2 * g + 1][b][0], st[2 * g + 1][b][1]), pack_bf2(st[2 * g + 1][b][2], st[2 * g + 1][b][3]));
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const bf16x8 vf = *(const bf16x8*)(smem + BUF * STAGE + VOFF + foff[g] + db * 2048);
#pragma unroll
        for (int b = 0; b < 2; ++b) o[db][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[g][b], o[db][b], 0, 0, 0);
      }
  };

  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  using B2 = std::integral_constant<int, 2>;
  using B3 = std::integral_constant<int, 3>;
  stage(B0{}, 0);
  if (1 < ntiles) stage(B1{}, 1);
#define ATTN16_LOOP(FX)                                                     \
  for (int it = 0; it < ntiles; it += 4) {                                  \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        \
    __syncthreads();                                                        \
    if (it + 2 < ntiles) stage(B2{}, it + 2);                               \
    if (it + 3 < ntiles) stage(B3{}, it + 3);                               \
    compute(B0{}, it, FX{});                                                \
    if (it + 1 < ntiles) compute(B1{}, it + 1, FX{});                       \
    if (it + 2 >= ntiles) break;                                            \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        \
    __syncthreads();                                                        \
    if (it + 4 < ntiles) stage(B0{}, it + 4);                               \
    if (it + 5 < ntiles) stage(B1{}, it + 5);                               \
    compute(B2{}, it + 2, FX{});                                            \
    if (it + 3 < ntiles) compute(B3{}, it + 3, FX{});                       \
  }
  if (fixed) { ATTN16_LOOP(std::true_type) } else { ATTN16_LOOP(std::false_type) }
#undef ATTN16_LOOP
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    float l = lsum[b];
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const float inv = 1.0f / l;
    const long long q = q0 + b * 16 + l15;
    if (q < N) {
      bf16_t* op = O + q * ldo + h * 64;
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        uint2 w;
        w.x = pack_bf2(o[db][b][0] * inv, o[db][b][1] * inv);
        w.y = pack_bf2(o[db][b][2] * inv, o[db][b][3] * inv);
        *(uint2*)(op + db * 16 + 4 * q4) = w;
      }
    }
  }
}

extern "C" void dove_set_error(const char*, ...) {}
extern "C" int attn16(const void* Qh, const void* Kh, const void* Vt16, void* O, long long N, long long Npad, int heads, long long ldo,
                      const float* norm2, void* stream) {
  constexpr int LDS = 4 * 16384;
  static bool once = false;
  if (!once) { (void)hipFuncSetAttribute((const void*)attn16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); once = true; }
  const int qblocks = (int)((Npad + 127) / 128);
  hipLaunchKernelGGL(attn16_kernel, dim3((unsigned)(qblocks * heads)), dim3(256), LDS, (hipStream_t)stream, (const bf16_t*)Qh, (const bf16_t*)Kh,
                     (const bf16_t*)Vt16, (bf16_t*)O, N, Npad, ldo, qblocks, norm2);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
