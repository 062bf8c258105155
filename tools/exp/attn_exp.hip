// EXPERIMENT harness (not part of the product library): templated variants of the flash-attention forward for within-run
// A/B on the GPU box (tools/archive/attn_ab.py).  Operand layout and numerics contract are those of dove_attention_fwd_bf16.
//   NEGM: the running max enters the S accumulator through the MFMA C operand (S - m costs no VALU), rescale deferred
//         until some score exceeds the running max by THR (base-2)
//   SUM : 0 serial fp32 row sum, 1 four partial sums, 2 row sum on the matrix pipe (ones x P^T)
//   BAR2: one workgroup barrier per TWO 64-key tiles (4 x 16 KB LDS stages)
#include <stdlib.h>

#include <type_traits>

#include "../../dove_amd/csrc/common.h"

typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ bf16x8 make_frag(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  u32x4 v = {a, b, c, d};
  return __builtin_bit_cast(bf16x8, v);
}

// first MFMA of an S chain with the accumulator INPUT in other registers than the output (C = -m broadcast, D = S - m):
// the builtin ties C to D and the compiler would copy the 16 registers first
__device__ __forceinline__ f32x16 mfma_c_in(bf16x8 a, bf16x8 b, const f32x16& c) {
  f32x16 d;
  asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ float max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

template <int NEGM, int SUM, int BAR2, int OCC>
__global__ __launch_bounds__(256, OCC) void attn_exp_kernel(const bf16_t* __restrict__ Qh, const bf16_t* __restrict__ Kh,
                                                            const bf16_t* __restrict__ Vt, bf16_t* __restrict__ O,
                                                            long long N, long long Npad, long long ldo) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STAGE = 16384, VOFF = 8192;
  constexpr float THR = 6.0f;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int h = blockIdx.y;
  const long long q0 = (long long)blockIdx.x * 128 + wave * 32;

  bf16x8 qf[4];
  {
    long long qrow = q0 + l31;
    if (qrow >= Npad) qrow = Npad - 1;
    const bf16_t* qp = Qh + ((long long)h * Npad + qrow) * 64 + hi * 8;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[kk] = *(const bf16x8*)(qp + kk * 16);
  }

  f32x16 o[2], lacc, negm;
  float m = NEGM ? 0.f : -1e30f, lsum = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; lacc[r] = 0.f; negm[r] = 0.f; }
  const bf16x8 ones = make_frag(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
  const bf16x8 kone = make_frag(hi ? 0u : 0x3f80u, 0u, 0u, 0u);        // A operand: K'[kv][k=0] = 1, other columns 0
  bf16x8 qm = make_frag(0u, 0u, 0u, 0u);                                // B operand: Q'^T[k=0][q] = -m

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
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_k, (lds_ptr_t)(smem + BUF * STAGE + (j * 256 + wave * 64) * 16), 16, vk[j],
                                               tile * (64 * 128), 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_v, (lds_ptr_t)(smem + BUF * STAGE + VOFF + (j * 256 + wave * 64) * 16), 16, vv[j],
                                               tile * (64 * 2), 0, 0);
    }
  };

  int koff[2][4];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int row = b * 32 + l31;
    const int sw = (row >> 1) & 7;
#pragma unroll
    for (int c = 0; c < 4; ++c) koff[b][c] = row * 128 + (((c * 2 + hi) ^ sw) << 4);
  }

  auto compute = [&](auto bufc, int tile) {
    constexpr int BUF = decltype(bufc)::value;
    f32x16 st[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const bf16x8 kf = *(const bf16x8*)(smem + BUF * STAGE + koff[kb][kk]);
        if (NEGM == 1 && kk == 0) st[kb] = mfma_c_in(kf, qf[kk], negm);
        else st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kk], st[kb], 0, 0, 0);
      }
    if (NEGM == 2) {   // the -m shift as a fifth K slice: K gets a constant-one column, Q^T the row (-m) (bf16-exact by construction)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kone, qm, st[kb], 0, 0, 0);
    }
    const long long kv0 = (long long)tile * 64;
    if (kv0 + 64 > N) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const long long kv = kv0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (kv >= N) st[kb][r] = -1e30f;
        }
    }
    float mt = st[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mt = fmaxf(mt, st[kb][r]);
    {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mt), __float_as_uint(mt), false, false);
      mt = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    if (NEGM) {
      // st holds S - m.  Slow path (wave-uniform): the first tile (m is not a max yet) or a score more than THR above m
      const bool first = tile == 0;
      if (first || __any(mt > THR)) {
        float delta = first ? mt : fmaxf(mt, 0.f);
        if (NEGM == 2) {                       // keep m exactly representable in bf16 (it rides in a bf16 MFMA operand)
          const float mup = __uint_as_float((pack_bf2(m + delta, 0.f) & 0xffffu) << 16);
          delta = mup - m;
          qm = make_frag(hi ? 0u : (pack_bf2(-mup, 0.f) & 0xffffu), 0u, 0u, 0u);
        }
        const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);
        m += delta;
        lsum *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          o[0][r] *= alpha; o[1][r] *= alpha; st[0][r] -= delta; st[1][r] -= delta;
          if (NEGM == 1) negm[r] = -m;
        }
        if (SUM == 2) {
#pragma unroll
          for (int r = 0; r < 16; ++r) lacc[r] *= alpha;
        }
      }
    } else {
      if (__any(mt > m)) {
        const float mnew = fmaxf(m, mt);
        const float alpha = __builtin_amdgcn_exp2f(m - mnew);
        m = mnew;
        lsum *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
        if (SUM == 2) {
#pragma unroll
          for (int r = 0; r < 16; ++r) lacc[r] *= alpha;
        }
      }
    }
    float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(NEGM ? st[kb][r] : st[kb][r] - m);
        st[kb][r] = p;
        if (SUM == 0) ps[0] += p;
        if (SUM == 1) ps[r & 3] += p;
      }
    if (SUM == 0) lsum += ps[0];
    if (SUM == 1) lsum += (ps[0] + ps[1]) + (ps[2] + ps[3]);
    bf16x8 pf[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        const int b = 8 * k2;
        const uint32_t a0 = pack_bf2(st[kb][b + 0], st[kb][b + 1]);
        const uint32_t a1 = pack_bf2(st[kb][b + 2], st[kb][b + 3]);
        const uint32_t b0 = pack_bf2(st[kb][b + 4], st[kb][b + 5]);
        const uint32_t b1 = pack_bf2(st[kb][b + 6], st[kb][b + 7]);
        const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
        pf[kb][k2] = make_frag(r0[0], r1[0], r0[1], r1[1]);
      }
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
          const bf16x8 vf = *(const bf16x8*)(smem + BUF * STAGE + VOFF + koff[db][kb * 2 + k2]);
          o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kb][k2], o[db], 0, 0, 0);
        }
    if (SUM == 2) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) lacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pf[kb][k2], lacc, 0, 0, 0);
    }
  };

  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  using B2 = std::integral_constant<int, 2>;
  using B3 = std::integral_constant<int, 3>;
  if (BAR2) {
    stage(B0{}, 0);
    if (1 < ntiles) stage(B1{}, 1);
    for (int it = 0; it < ntiles; it += 4) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (it + 2 < ntiles) stage(B2{}, it + 2);
      if (it + 3 < ntiles) stage(B3{}, it + 3);
      compute(B0{}, it);
      if (it + 1 < ntiles) compute(B1{}, it + 1);
      if (it + 2 >= ntiles) break;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (it + 4 < ntiles) stage(B0{}, it + 4);
      if (it + 5 < ntiles) stage(B1{}, it + 5);
      compute(B2{}, it + 2);
      if (it + 3 < ntiles) compute(B3{}, it + 3);
    }
  } else {
    stage(B0{}, 0);
    int it = 0;
    for (; it + 2 <= ntiles; it += 2) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      stage(B1{}, it + 1);
      compute(B0{}, it);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (it + 2 < ntiles) stage(B0{}, it + 2);
      compute(B1{}, it + 1);
    }
    if (ntiles & 1) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      compute(B0{}, ntiles - 1);
    }
  }

  float l;
  if (SUM == 2) l = lacc[0];
  else l = lsum + __shfl_xor(lsum, 32);
  const float inv = 1.0f / l;
  const long long q = q0 + l31;
  if (q < N) {
    bf16_t* op = O + q * ldo + h * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = db * 32 + 8 * g + 4 * hi;
        uint2 w;
        w.x = pack_bf2(o[db][g * 4 + 0] * inv, o[db][g * 4 + 1] * inv);
        w.y = pack_bf2(o[db][g * 4 + 2] * inv, o[db][g * 4 + 3] * inv);
        *(uint2*)(op + d) = w;
      }
  }
}

template <int NEGM, int SUM, int BAR2, int OCC>
static int launch(const void* Qh, const void* Kh, const void* Vt, void* O, long long N, long long Npad, int heads, long long ldo,
                  hipStream_t s) {
  constexpr int lds = BAR2 ? 65536 : 32768;
  (void)hipFuncSetAttribute((const void*)attn_exp_kernel<NEGM, SUM, BAR2, OCC>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  dim3 grid((unsigned)(Npad / 128), heads);
  hipLaunchKernelGGL((attn_exp_kernel<NEGM, SUM, BAR2, OCC>), grid, dim3(256), lds, s, (const bf16_t*)Qh, (const bf16_t*)Kh,
                     (const bf16_t*)Vt, (bf16_t*)O, N, Npad, ldo);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" void dove_set_error(const char*, ...) {}

extern "C" int attn_exp(int variant, const void* Qh, const void* Kh, const void* Vt, void* O, long long N, long long Npad, int heads,
                        long long ldo, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  switch (variant) {
    case 0: return launch<0, 0, 0, 2>(Qh, Kh, Vt, O, N, Npad, heads, ldo, s);   // = the round-1 product kernel (+ permlane max)
    case 1: return launch<1, 0, 0, 2>(Qh, Kh, Vt, O, N, Npad, heads, ldo, s);
    case 2: return launch<1, 1, 0, 2>(Qh, Kh, Vt, O, N, Npad, heads, ldo, s);
    case 3: return launch<1, 2, 0, 2>(Qh, Kh, Vt, O, N, Npad, heads, ldo, s);
    case 4: return launch<0, 2, 0, 2>(Qh, Kh, Vt, O, N, Npad, heads, ldo, s);
    case 5: return launch<1, 2, 1, 2>(Qh, Kh, Vt, O, N, Npad, heads, ldo, s);
    case 6: return launch<1, 0, 1, 2>(Qh, Kh, Vt, O, N, Npad, heads, ldo, s);
    case 7: return launch<0, 0, 1, 2>(Qh, Kh, Vt, O, N, Npad, heads, ldo, s);
    case 8: return launch<1, 1, 1, 2>(Qh, Kh, Vt, O, N, Npad, heads, ldo, s);
    case 9: return launch<2, 0, 0, 2>(Qh, Kh, Vt, O, N, Npad, heads, ldo, s);
    case 10: return launch<2, 1, 0, 2>(Qh, Kh, Vt, O, N, Npad, heads, ldo, s);
    case 11: return launch<2, 1, 1, 2>(Qh, Kh, Vt, O, N, Npad, heads, ldo, s);
    case 12: return launch<2, 2, 0, 2>(Qh, Kh, Vt, O, N, Npad, heads, ldo, s);
    case 13: return launch<2, 0, 1, 2>(Qh, Kh, Vt, O, N, Npad, heads, ldo, s);
  }
  return -1;
}

// ---------------------------------------------------------------------------------------------------------------------------
// QB2: 64 query rows per wave (two 32-row blocks), 256 rows per workgroup, 2 waves per SIMD.  Every K / V^T fragment read
// from LDS feeds two MFMAs, K/V staging and the barrier are amortised over twice the MFMA work, and the two blocks' softmax
// chains are independent so one block's VALU can sit under the other's MFMAs.  -m rides in a fifth K slice (NEGM == 2 above).
template <int SUMV, int BAR2, int FIFTH>
__global__ __launch_bounds__(256, 2) void attn_qb2_kernel(const bf16_t* __restrict__ Qh, const bf16_t* __restrict__ Kh,
                                                          const bf16_t* __restrict__ Vt, bf16_t* __restrict__ O,
                                                          long long N, long long Npad, long long ldo) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STAGE = 16384, VOFF = 8192;
  constexpr float THR = 6.0f;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int h = blockIdx.y;
  const long long q0 = (long long)blockIdx.x * 256 + wave * 64;

  bf16x8 qf[2][4];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    long long qrow = q0 + qb * 32 + l31;
    if (qrow >= Npad) qrow = Npad - 1;
    const bf16_t* qp = Qh + ((long long)h * Npad + qrow) * 64 + hi * 8;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[qb][kk] = *(const bf16x8*)(qp + kk * 16);
  }
  f32x16 o[2][2];
  float m[2] = {FIFTH ? 0.f : -1e30f, FIFTH ? 0.f : -1e30f}, lsum[2] = {0.f, 0.f};
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[qb][0][r] = 0.f; o[qb][1][r] = 0.f; }
  const bf16x8 kone = make_frag(hi ? 0u : 0x3f80u, 0u, 0u, 0u);
  bf16x8 qm[2] = {make_frag(0u, 0u, 0u, 0u), make_frag(0u, 0u, 0u, 0u)};

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
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_k, (lds_ptr_t)(smem + BUF * STAGE + (j * 256 + wave * 64) * 16), 16, vk[j],
                                               tile * (64 * 128), 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_v, (lds_ptr_t)(smem + BUF * STAGE + VOFF + (j * 256 + wave * 64) * 16), 16, vv[j],
                                               tile * (64 * 2), 0, 0);
    }
  };
  int koff[2][4];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int row = b * 32 + l31;
    const int sw = (row >> 1) & 7;
#pragma unroll
    for (int c = 0; c < 4; ++c) koff[b][c] = row * 128 + (((c * 2 + hi) ^ sw) << 4);
  }

  auto compute = [&](auto bufc, int tile) {
    constexpr int BUF = decltype(bufc)::value;
    f32x16 st[2][2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) st[qb][kb][r] = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const bf16x8 kf = *(const bf16x8*)(smem + BUF * STAGE + koff[kb][kk]);
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) st[qb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[qb][kk], st[qb][kb], 0, 0, 0);
      }
      if (FIFTH) {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) st[qb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kone, qm[qb], st[qb][kb], 0, 0, 0);
      }
    }
    const long long kv0 = (long long)tile * 64;
    if (kv0 + 64 > N) {
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const long long kv = kv0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (kv >= N) st[qb][kb][r] = -1e30f;
          }
    }
    float mt[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      float x = st[qb][0][0];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) x = fmaxf(x, st[qb][kb][r]);
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
      mt[qb] = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    const bool first = tile == 0;
    if (!FIFTH) {
      if (__any(mt[0] > m[0] || mt[1] > m[1])) {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
          const float mnew = fmaxf(m[qb], mt[qb]);
          const float alpha = __builtin_amdgcn_exp2f(m[qb] - mnew);
          m[qb] = mnew;
          lsum[qb] *= alpha;
#pragma unroll
          for (int r = 0; r < 16; ++r) { o[qb][0][r] *= alpha; o[qb][1][r] *= alpha; }
        }
      }
    } else if (first || __any(fmaxf(mt[0], mt[1]) > THR)) {          // one rare, wave-uniform branch for both query blocks
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        float delta = first ? mt[qb] : fmaxf(mt[qb], 0.f);
        const float mup = __uint_as_float((pack_bf2(m[qb] + delta, 0.f) & 0xffffu) << 16);   // keep m bf16-exact
        delta = mup - m[qb];
        qm[qb] = make_frag(hi ? 0u : (pack_bf2(-mup, 0.f) & 0xffffu), 0u, 0u, 0u);
        const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);
        m[qb] = mup;
        lsum[qb] *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          o[qb][0][r] *= alpha; o[qb][1][r] *= alpha; st[qb][0][r] -= delta; st[qb][1][r] -= delta;
        }
      }
    }
    // straight-line from here: softmax of both blocks + PV, one basic block for the scheduler
    bf16x8 pf[2][2][2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = __builtin_amdgcn_exp2f(FIFTH ? st[qb][kb][r] : st[qb][kb][r] - m[qb]);
          st[qb][kb][r] = p;
          ps[SUMV ? (r & 3) : 0] += p;
        }
      lsum[qb] += SUMV ? (ps[0] + ps[1]) + (ps[2] + ps[3]) : ps[0];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
          const int b = 8 * k2;
          const uint32_t a0 = pack_bf2(st[qb][kb][b + 0], st[qb][kb][b + 1]);
          const uint32_t a1 = pack_bf2(st[qb][kb][b + 2], st[qb][kb][b + 3]);
          const uint32_t b0 = pack_bf2(st[qb][kb][b + 4], st[qb][kb][b + 5]);
          const uint32_t b1 = pack_bf2(st[qb][kb][b + 6], st[qb][kb][b + 7]);
          const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
          const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
          pf[qb][kb][k2] = make_frag(r0[0], r1[0], r0[1], r1[1]);
        }
    }
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
          const bf16x8 vf = *(const bf16x8*)(smem + BUF * STAGE + VOFF + koff[db][kb * 2 + k2]);
#pragma unroll
          for (int qb = 0; qb < 2; ++qb) o[qb][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[qb][kb][k2], o[qb][db], 0, 0, 0);
        }
  };

  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  using B2 = std::integral_constant<int, 2>;
  using B3 = std::integral_constant<int, 3>;
  if (BAR2) {
    stage(B0{}, 0);
    if (1 < ntiles) stage(B1{}, 1);
    for (int it = 0; it < ntiles; it += 4) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (it + 2 < ntiles) stage(B2{}, it + 2);
      if (it + 3 < ntiles) stage(B3{}, it + 3);
      compute(B0{}, it);
      if (it + 1 < ntiles) compute(B1{}, it + 1);
      if (it + 2 >= ntiles) break;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (it + 4 < ntiles) stage(B0{}, it + 4);
      if (it + 5 < ntiles) stage(B1{}, it + 5);
      compute(B2{}, it + 2);
      if (it + 3 < ntiles) compute(B3{}, it + 3);
    }
  } else {
    stage(B0{}, 0);
    int it = 0;
    for (; it + 2 <= ntiles; it += 2) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      stage(B1{}, it + 1);
      compute(B0{}, it);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (it + 2 < ntiles) stage(B0{}, it + 2);
      compute(B1{}, it + 1);
    }
    if (ntiles & 1) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      compute(B0{}, ntiles - 1);
    }
  }
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const float l = lsum[qb] + __shfl_xor(lsum[qb], 32);
    const float inv = 1.0f / l;
    const long long q = q0 + qb * 32 + l31;
    if (q < N) {
      bf16_t* op = O + q * ldo + h * 64;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d = db * 32 + 8 * g + 4 * hi;
          uint2 w;
          w.x = pack_bf2(o[qb][db][g * 4 + 0] * inv, o[qb][db][g * 4 + 1] * inv);
          w.y = pack_bf2(o[qb][db][g * 4 + 2] * inv, o[qb][db][g * 4 + 3] * inv);
          *(uint2*)(op + d) = w;
        }
    }
  }
}

template <int SUMV, int BAR2, int FIFTH>
static int launch_qb2(const void* Qh, const void* Kh, const void* Vt, void* O, long long N, long long Npad, int heads, long long ldo,
                      hipStream_t s) {
  constexpr int lds = BAR2 ? 65536 : 32768;
  (void)hipFuncSetAttribute((const void*)attn_qb2_kernel<SUMV, BAR2, FIFTH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  dim3 grid((unsigned)((Npad + 255) / 256), heads);
  hipLaunchKernelGGL((attn_qb2_kernel<SUMV, BAR2, FIFTH>), grid, dim3(256), lds, s, (const bf16_t*)Qh, (const bf16_t*)Kh,
                     (const bf16_t*)Vt, (bf16_t*)O, N, Npad, ldo);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int attn_exp2(int variant, const void* Qh, const void* Kh, const void* Vt, void* O, long long N, long long Npad, int heads,
                         long long ldo, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  switch (variant) {
    case 20: return launch_qb2<0, 0, 1>(Qh, Kh, Vt, O, N, Npad, heads, ldo, s);
    case 21: return launch_qb2<1, 0, 1>(Qh, Kh, Vt, O, N, Npad, heads, ldo, s);
    case 22: return launch_qb2<0, 1, 1>(Qh, Kh, Vt, O, N, Npad, heads, ldo, s);
    case 23: return launch_qb2<1, 1, 1>(Qh, Kh, Vt, O, N, Npad, heads, ldo, s);
    case 24: return launch_qb2<0, 1, 0>(Qh, Kh, Vt, O, N, Npad, heads, ldo, s);
    case 25: return launch_qb2<0, 0, 0>(Qh, Kh, Vt, O, N, Npad, heads, ldo, s);
  }
  return -1;
}
