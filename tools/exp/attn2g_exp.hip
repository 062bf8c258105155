// EXPERIMENT (tools/archive/attn_ab.py variants 40+; never loaded by dove_amd): flash attention forward, head_dim 64, with TWO WAVE GROUPS
// ONE BARRIER PHASE APART (MI355X_MICROARCH.md "Two waves per SIMD").
//
// Workgroup = 8 waves = 256 queries (32 per wave), two waves per SIMD: group A = waves 0-3, group B = waves 4-7.  Per 64-key tile a
// wave runs a MATRIX segment M(t) = { S_t = K_t Q^T - m (8 MFMAs), O += V_{t-1} P_{t-1} (8 MFMAs) } and a VALU segment
// V(t) = { lazy max / rescale, P_t = 2^(S_t), row sums, bf16 pack }.  Within one barrier phase group A runs M while group B runs V and
// in the next phase they swap, so on every SIMD one wave feeds the matrix pipe while its partner issues the softmax VALU - the two
// streams that ADD in the product kernel (290 ns of MFMA + 311 ns of VALU per wave-tile).
//   iteration it:   phase 2it:    A: M(it)      B: V(it-1)        barrier
//                   phase 2it+1:  A: V(it)      B: M(it)          barrier
// K_t / V^T_t tiles (16 KB) are shared by all 8 waves (half the K/V re-streaming of the 4-wave kernel), staged by LDS-DMA into a
// 4-slot ring, tile it+2 issued at the start of iteration it (one K piece + one V^T piece per wave), waited for with a COUNTED
// vmcnt(2) at the end of iteration it+1, one barrier before its first reader.  Operand layouts, swizzle, lazy rescale (-m through the
// MFMA's C operand) and rounding points are those of the product kernel (dove_amd/csrc/attention.hip): results are bit-identical.
#include <stdlib.h>

#include <type_traits>

#include "../../dove_amd/csrc/common.h"

typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ bf16x8 make_frag(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_;
  u32x4_ v = {a, b, c, d};
  return __builtin_bit_cast(bf16x8, v);
}

// D = A B + C with C in other registers than D (C = -m broadcast).  volatile + trailing nops: hipcc pads nothing around an asm MFMA
// (guide 5.7 item 2): `s_nop 1` covers a VALU write of C just before, the trailing `s_nop 7, s_nop 3` the 12 wait states an
// 8-pass XDL result needs before a non-MFMA reader (the next instruction here is always the chained MFMA taking D as C: 0 needed,
// the nops make the statement safe wherever the compiler moves the following code).
template <bool SAFE>
__device__ __forceinline__ f32x16 mfma_c_in(bf16x8 a, bf16x8 b, const f32x16& c) {
  f32x16 d;
  if (SAFE) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %3\n\ts_nop 7\n\ts_nop 3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
  else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

template <int PRIO, bool SAFE>    // SAFE: padded asm MFMA (see mfma_c_in); PRIO 0: no priorities; 1: static s_setprio 1 for the younger half (waves 4-7); 2: s_setprio 1 around every M segment
__global__ __launch_bounds__(512, 1) void attn2g_kernel(const bf16_t* __restrict__ Qh, const bf16_t* __restrict__ Kh,
                                                         const bf16_t* __restrict__ Vt, bf16_t* __restrict__ O, long long N, long long Npad,
                                                         long long ldo) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STAGE = 16384, VOFF = 8192;
  constexpr float THR = 6.0f;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool grpA = wave < 4;
  const int hi = lane >> 5, l31 = lane & 31;
  const int h = blockIdx.y;
  const long long q0 = (long long)blockIdx.x * 256 + wave * 32;

  bf16x8 qf[4];
  {
    long long qrow = q0 + l31;
    if (qrow >= Npad) qrow = Npad - 1;
    const bf16_t* qp = Qh + ((long long)h * Npad + qrow) * 64 + hi * 8;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[kk] = *(const bf16x8*)(qp + kk * 16);
  }
  f32x16 o[2], negm, st[2];
  bf16x8 pf[2][2];
  float m = 0.f, lsum = 0.f;
  // st = -1e30: the VALU segment group B runs before its first matrix segment (tile -1) then produces P = 0, row sum 0, no rescale
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; negm[r] = 0.f; st[0][r] = -1e30f; st[1][r] = -1e30f; }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) pf[a][b] = make_frag(0, 0, 0, 0);

  const int ntiles = (int)((N + 63) / 64);
  // staging: 512 threads cover the 64 rows x 8 sixteen-byte chunks of a K tile (and of a V^T tile) in one instruction each
  const int srow = tid >> 3;
  const int sc_ld = (tid & 7) ^ ((srow >> 1) & 7);
  const auto srd_k = __builtin_amdgcn_make_buffer_rsrc((void*)(Kh + (long long)h * Npad * 64), (short)0, (int)(Npad * 128), 0x00020000);
  const auto srd_v = __builtin_amdgcn_make_buffer_rsrc((void*)(Vt + (long long)h * 64 * Npad), (short)0, (int)(Npad * 128), 0x00020000);
  const unsigned vk = (unsigned)((srow * 64 + sc_ld * 8) * 2);
  const unsigned vv = (unsigned)((((long long)srow) * Npad + sc_ld * 8) * 2);
  // tile t -> ring slot t & 3.  Tiles outside [0, ntiles) read out of range: the descriptor returns zeros (tile -1 zero-fills the
  // slot the very first P V product reads; tiles past the end land in slots nobody reads any more)
  auto stage = [&](int tile) {
    const int slot = __builtin_amdgcn_readfirstlane(tile & 3);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_k, (lds_ptr_t)(smem + slot * STAGE + wave * 1024), 16, vk, tile * (64 * 128), 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_v, (lds_ptr_t)(smem + slot * STAGE + VOFF + wave * 1024), 16, vv, tile * (64 * 2), 0, 0);
  };
  int koff[2][4];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int row = b * 32 + l31;
    const int sw = (row >> 1) & 7;
#pragma unroll
    for (int c = 0; c < 4; ++c) koff[b][c] = row * 128 + (((c * 2 + hi) ^ sw) << 4);
  }

  // ---- matrix segment of tile t: S_t = K_t Q^T - m from slot t & 3, O += V_{t-1} P_{t-1} from slot (t - 1) & 3.  Unconditional:
  // out-of-range tiles are zero tiles and P_{-1} = 0 ----
  auto mseg = [&](int t) {
    const int kbase = (t & 3) * STAGE, vbase = ((t - 1) & 3) * STAGE + VOFF;
    if (PRIO == 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const bf16x8 kf = *(const bf16x8*)(smem + kbase + koff[kb][kk]);
        if (kk == 0) st[kb] = mfma_c_in<SAFE>(kf, qf[kk], negm);
        else st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kk], st[kb], 0, 0, 0);
      }
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
          const bf16x8 vf = *(const bf16x8*)(smem + vbase + koff[db][kb * 2 + k2]);
          o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kb][k2], o[db], 0, 0, 0);
        }
    if (PRIO == 2) __builtin_amdgcn_s_setprio(0);
  };
  // ---- VALU segment of tile t: lazy online softmax (base 2; Q carries scale*log2e), P_t packed for the next matrix segment.
  // MASK: the clip's last tile (keys >= N are pad rows) ----
  auto vseg = [&](auto maskc, int t) {
    constexpr bool MASK = decltype(maskc)::value;
    if (MASK) {
      const long long kv0 = (long long)t * 64;
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
    const bool first = t == 0;
    if (first || __any(mt > THR)) {
      // O holds tiles < t (PV_{t-1} finished in this wave's previous matrix segment): everything at the old max is rescaled once
      const float delta = first ? mt : fmaxf(mt, 0.f);
      const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);
      m += delta;
      lsum *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        o[0][r] *= alpha; o[1][r] *= alpha; st[0][r] -= delta; st[1][r] -= delta; negm[r] = -m;
      }
    }
    float ps = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(st[kb][r]);
        st[kb][r] = p;
        ps += p;
      }
    lsum += ps;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        const int b = 8 * k2;
        pf[kb][k2] = make_frag(pack_bf2(st[kb][b + 0], st[kb][b + 1]), pack_bf2(st[kb][b + 2], st[kb][b + 3]),
                               pack_bf2(st[kb][b + 4], st[kb][b + 5]), pack_bf2(st[kb][b + 6], st[kb][b + 7]));
      }
  };
  using NoMask = std::integral_constant<bool, false>;
  using Mask = std::integral_constant<bool, true>;
  if (PRIO == 1 && !grpA) __builtin_amdgcn_s_setprio(1);       // wave-uniform: `wave` comes from readfirstlane

  // phase ends: every ds_read of the phase has returned (the slot may be re-staged after the barrier); at the end of an iteration tile
  // it+1 has landed (vmcnt(2): only tile it+2's two instructions may still be in flight)
#define PHASE_END_A() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); } while (0)
#define PHASE_END_B() do { asm volatile("s_waitcnt vmcnt(2)\n\ts_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); } while (0)
  stage(-1);
  stage(0);
  stage(1);
  asm volatile("s_waitcnt vmcnt(2)" ::: "memory");             // tiles -1 and 0 landed
  __builtin_amdgcn_s_barrier();
  // The two groups run SEPARATE loops with the same barrier sequence: one loop with `if (grpA) mseg else vseg` makes the compiler
  // reconcile the register state of the two branches with ~180 register copies per iteration.
  if (grpA) {
    int it = 0;
    for (; it < ntiles - 1; ++it) {                             // tiles whose VALU segments need no mask
      stage(it + 2);
      mseg(it);
      PHASE_END_A();
      vseg(NoMask{}, it);
      PHASE_END_B();
    }
    stage(it + 2);                                              // it = ntiles - 1: the clip's last (ragged) tile
    mseg(it);
    PHASE_END_A();
    vseg(Mask{}, it);
    PHASE_END_B();
    mseg(it + 1);                                               // it = ntiles: O += V P of the last tile
    PHASE_END_A();
  } else {
    int it = 0;
    for (; it < ntiles; ++it) {
      stage(it + 2);
      vseg(NoMask{}, it - 1);                                   // it = 0: tile -1 is the all -1e30 initial state (P = 0)
      PHASE_END_A();
      mseg(it);
      PHASE_END_B();
    }
    vseg(Mask{}, it - 1);
    PHASE_END_A();
    mseg(it);
  }
#undef PHASE_END_A
#undef PHASE_END_B
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // no LDS-DMA may outlive the workgroup's LDS allocation

  const float l = lsum + __shfl_xor(lsum, 32);
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

template <int PRIO, bool SAFE>
static int launch2g(const void* Qh, const void* Kh, const void* Vt, void* O, long long N, long long Npad, int heads, long long ldo, hipStream_t s) {
  constexpr int LDS = 4 * 16384;
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)attn2g_kernel<PRIO, SAFE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); attr = true; }
  dim3 grid((unsigned)((Npad + 255) / 256), heads);
  hipLaunchKernelGGL((attn2g_kernel<PRIO, SAFE>), grid, dim3(512), LDS, s, (const bf16_t*)Qh, (const bf16_t*)Kh, (const bf16_t*)Vt, (bf16_t*)O, N, Npad, ldo);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" void dove_set_error(const char*, ...) {}
// V^T in the quad-swapped key order of the product kernel
extern "C" int attn_exp4(int variant, const void* Qh, const void* Kh, const void* Vt, void* O, long long N, long long Npad, int heads,
                         long long ldo, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  switch (variant) {
    case 40: return launch2g<0, true>(Qh, Kh, Vt, O, N, Npad, heads, ldo, s);
    case 41: return launch2g<1, true>(Qh, Kh, Vt, O, N, Npad, heads, ldo, s);
    case 42: return launch2g<2, true>(Qh, Kh, Vt, O, N, Npad, heads, ldo, s);
    case 43: return launch2g<0, false>(Qh, Kh, Vt, O, N, Npad, heads, ldo, s);
    case 44: return launch2g<1, false>(Qh, Kh, Vt, O, N, Npad, heads, ldo, s);
  }
  return -1;
}
