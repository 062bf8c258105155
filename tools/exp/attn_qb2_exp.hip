// OUTCOME (profiles/r03_attn_qb2.log): NOT usable as written.  (i) 360 registers: hipcc keeps S / P in AGPRs and moves them through
// ~350 v_accvgpr_read/write per iteration (VALU cannot address AGPRs): 8.6 ms vs 4.3 ms for the product kernel; (ii) the results are
// WRONG (max err 5.8): with both query blocks' MFMA chains and VALU in one scheduling region the compiler places a VALU reader of the
// inline-asm MFMA's result inside its 12-wait-state window - the hazard of an asm statement it cannot see (guide 5.7 item 2; the
// product kernel is pinned against this by tests/test_isa_checks.py).  A working version needs every MFMA in asm with explicit
// "a" / "v" register classes and a hand-placed 1 MFMA : 5 VALU interleave; tools/archive/coissue.py bounds what that could gain at ~13 %
// of the attention time (478 ns per 16 MFMA + softmax mix against 540 ns now).  Kept as the record of the attempt.
//
// EXPERIMENT (tools/archive/attn_ab.py variants 50+; never loaded by dove_amd): flash attention forward, head_dim 64, ONE wave per SIMD,
// TWO query blocks per wave, skewed by half a tile so that every block of 16 MFMAs has the softmax VALU of the OTHER query block to
// interleave with IN THE SAME WAVE.
//
// Why this shape (tools/archive/coissue.py, profiles/r03_coissue_mfma_valu.log): on gfx950 an MFMA stream and a VALU stream issued by two
// DIFFERENT waves of one SIMD serialize completely (t = t_mfma + t_valu, whatever the priorities), while ONE wave that alternates
// 1 MFMA : ~7 independent VALU hides about half of the VALU time under its own MFMAs (16 MFMA + the softmax mix: 478 ns instead of
// 302 + 353).  Within one query block a tile is a dependency chain (QK^T -> softmax -> PV), so the independent VALU has to come from
// another query block:
//     step 1:  MFMA  S_B = K_j Q_B^T - m_B,  O_B += V_{j-1} P_B(j-1)        VALU  softmax(A, j)   -> P_A(j)
//     step 2:  MFMA  O_A += V_j P_A(j),      S_A = K_{j+1} Q_A^T - m_A      VALU  softmax(B, j)   -> P_B(j)
// Fragments are read from LDS per MFMA like in the product kernel (keeping K / V^T fragments in registers across the two steps pushed
// the wave past 256 VGPRs and the compiler moved S / P through AGPRs with ~110 v_accvgpr copies per tile).  8-slot LDS ring (128 KB),
// tiles staged 4 ahead.  One wave per SIMD (launch bounds 256, 1).
// Operand layouts, swizzle, lazy rescale (-m through the MFMA's C operand) and every rounding point are the product kernel's
// (dove_amd/csrc/attention.hip): results are bit-identical.
#include <stdlib.h>

#include <type_traits>

#include "../../dove_amd/csrc/common.h"

typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ bf16x8 make_frag(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_;
  u32x4_ v = {a, b, c, d};
  return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ f32x16 mfma_c_in(bf16x8 a, bf16x8 b, const f32x16& c) {
  f32x16 d;
  asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

struct QBlock {                 // per-query-block state of a wave
  bf16x8 qf[4];
  f32x16 o[2], negm, st[2];
  bf16x8 pf[2][2];
  float m, lsum;
};

// VALU half of a tile for one query block: lazy online softmax of st (already S - m), P packed into pf.  MASK: the clip's last tile.
template <bool MASK>
__device__ __forceinline__ void softmax_tile(QBlock& q, int t, long long N, int hi) {
  constexpr float THR = 6.0f;
  if (MASK) {
    const long long kv0 = (long long)t * 64;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long long kv = kv0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (kv >= N) q.st[kb][r] = -1e30f;
      }
  }
  float mt = q.st[0][0];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) mt = fmaxf(mt, q.st[kb][r]);
  {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mt), __float_as_uint(mt), false, false);
    mt = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
  }
  const bool first = t == 0;
  if (first || __any(mt > THR)) {
    const float delta = first ? mt : fmaxf(mt, 0.f);
    const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);
    q.m += delta;
    q.lsum *= alpha;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      q.o[0][r] *= alpha; q.o[1][r] *= alpha; q.st[0][r] -= delta; q.st[1][r] -= delta; q.negm[r] = -q.m;
    }
  }
  float ps = 0.f;
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = __builtin_amdgcn_exp2f(q.st[kb][r]);
      q.st[kb][r] = p;
      ps += p;
    }
  q.lsum += ps;
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
      const int b = 8 * k2;
      q.pf[kb][k2] = make_frag(pack_bf2(q.st[kb][b + 0], q.st[kb][b + 1]), pack_bf2(q.st[kb][b + 2], q.st[kb][b + 3]),
                               pack_bf2(q.st[kb][b + 4], q.st[kb][b + 5]), pack_bf2(q.st[kb][b + 6], q.st[kb][b + 7]));
    }
}
// S = K Q^T - m, K fragments from the LDS tile at `kt` (per-lane constant offsets koff)
__device__ __forceinline__ void qk_tile(QBlock& q, const char* kt, const int (&koff)[2][4]) {
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const bf16x8 kf = *(const bf16x8*)(kt + koff[kb][kk]);
      if (kk == 0) q.st[kb] = mfma_c_in(kf, q.qf[kk], q.negm);
      else q.st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, q.qf[kk], q.st[kb], 0, 0, 0);
    }
}
// O += V^T P^T, V^T fragments from the LDS tile at `vt`
__device__ __forceinline__ void pv_tile(QBlock& q, const char* vt, const int (&koff)[2][4]) {
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        const bf16x8 vf = *(const bf16x8*)(vt + koff[db][kb * 2 + k2]);
        q.o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, q.pf[kb][k2], q.o[db], 0, 0, 0);
      }
}

template <int SCHED>   // 0: compiler's schedule; 1: sched_group_barrier 1 MFMA : 6 VALU in both steps
__global__ __launch_bounds__(256, 1) void attn_qb2_kernel(const bf16_t* __restrict__ Qh, const bf16_t* __restrict__ Kh,
                                                           const bf16_t* __restrict__ Vt, bf16_t* __restrict__ O, long long N, long long Npad,
                                                           long long ldo) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STAGE = 16384, VOFF = 8192;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int h = blockIdx.y;
  const long long q0 = (long long)blockIdx.x * 256 + wave * 64;       // this wave: queries [q0, q0 + 32) = block A, [q0 + 32, q0 + 64) = block B

  QBlock A, B;
  auto init = [&](QBlock& q, long long base) {
    long long qrow = base + l31;
    if (qrow >= Npad) qrow = Npad - 1;
    const bf16_t* qp = Qh + ((long long)h * Npad + qrow) * 64 + hi * 8;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) q.qf[kk] = *(const bf16x8*)(qp + kk * 16);
#pragma unroll
    for (int r = 0; r < 16; ++r) { q.o[0][r] = 0.f; q.o[1][r] = 0.f; q.negm[r] = 0.f; q.st[0][r] = -1e30f; q.st[1][r] = -1e30f; }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) q.pf[a][b] = make_frag(0, 0, 0, 0);
    q.m = 0.f; q.lsum = 0.f;
  };
  init(A, q0);
  init(B, q0 + 32);

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
  // tile t -> ring slot t & 7 (8 x 16 KB); tiles outside [0, ntiles) read out of range = zero tiles
  auto stage = [&](int tile) {
    const int slot = __builtin_amdgcn_readfirstlane(tile & 7);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_k, (lds_ptr_t)(smem + slot * STAGE + (j * 256 + wave * 64) * 16), 16, vk[j], tile * (64 * 128), 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_v, (lds_ptr_t)(smem + slot * STAGE + VOFF + (j * 256 + wave * 64) * 16), 16, vv[j], tile * (64 * 2), 0, 0);
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
  auto ktile = [&](int t) { return (const char*)smem + (t & 7) * STAGE; };
  auto vtile = [&](int t) { return (const char*)smem + (t & 7) * STAGE + VOFF; };
  using NoMask = std::integral_constant<bool, false>;
  using Mask = std::integral_constant<bool, true>;

  // one tile j (steady state):
  //   step 1: S_B = K_j Q_B - m_B, O_B += V_{j-1} P_B(j-1)   ||  softmax(A, j)
  //   step 2: O_A += V_j P_A(j), S_A = K_{j+1} Q_A - m_A      ||  softmax(B, j)
  auto tile = [&](auto maskc, int j) {
    constexpr bool MASK = decltype(maskc)::value;
    stage(j + 4);                                                // slot (j + 4) & 7: tile j - 4, long consumed
    qk_tile(B, ktile(j), koff);
    pv_tile(B, vtile(j - 1), koff);
    softmax_tile<MASK>(A, j, N, hi);
    if (SCHED == 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 6, 0); }
    }
    pv_tile(A, vtile(j), koff);
    qk_tile(A, ktile(j + 1), koff);
    softmax_tile<MASK>(B, j, N, hi);
    if (SCHED == 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 6, 0); }
    }
    // tile j + 2 has landed (the two newest tiles, j + 3 and j + 4 = 8 instructions, may still be in flight)
    asm volatile("s_waitcnt vmcnt(8)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };

  // prologue: tile -1 (the first O_B update reads V_{-1}: a zero tile, P_B(-1) = 0), tiles 0..3 staged; S_A(0) computed
  stage(-1);
  stage(0);
  stage(1);
  stage(2);
  stage(3);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");             // tiles -1, 0, 1 landed
  __builtin_amdgcn_s_barrier();
  qk_tile(A, ktile(0), koff);
  int j = 0;
  for (; j < ntiles - 1; ++j) tile(NoMask{}, j);
  tile(Mask{}, j);                                              // the last tile: masked; its QK(A, ntiles) runs on a zero tile and is never used
  pv_tile(B, vtile(ntiles - 1), koff);                          // O_B += V_{ntiles-1} P_B(ntiles-1)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  auto store = [&](QBlock& q, long long base) {
    const float l = q.lsum + __shfl_xor(q.lsum, 32);
    const float inv = 1.0f / l;
    const long long qi = base + l31;
    if (qi < N) {
      bf16_t* op = O + qi * ldo + h * 64;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d = db * 32 + 8 * g + 4 * hi;
          uint2 w;
          w.x = pack_bf2(q.o[db][g * 4 + 0] * inv, q.o[db][g * 4 + 1] * inv);
          w.y = pack_bf2(q.o[db][g * 4 + 2] * inv, q.o[db][g * 4 + 3] * inv);
          *(uint2*)(op + d) = w;
        }
    }
  };
  store(A, q0);
  store(B, q0 + 32);
}

template <int SCHED>
static int launch_qb2(const void* Qh, const void* Kh, const void* Vt, void* O, long long N, long long Npad, int heads, long long ldo, hipStream_t s) {
  constexpr int LDS = 8 * 16384;
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)attn_qb2_kernel<SCHED>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); attr = true; }
  dim3 grid((unsigned)((Npad + 255) / 256), heads);
  hipLaunchKernelGGL(attn_qb2_kernel<SCHED>, grid, dim3(256), LDS, s, (const bf16_t*)Qh, (const bf16_t*)Kh, (const bf16_t*)Vt, (bf16_t*)O, N, Npad, ldo);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" void dove_set_error(const char*, ...) {}
// V^T in the quad-swapped key order of the product kernel
extern "C" int attn_exp5(int variant, const void* Qh, const void* Kh, const void* Vt, void* O, long long N, long long Npad, int heads,
                         long long ldo, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  switch (variant) {
    case 50: return launch_qb2<0>(Qh, Kh, Vt, O, N, Npad, heads, ldo, s);
    case 51: return launch_qb2<1>(Qh, Kh, Vt, O, N, Npad, heads, ldo, s);
  }
  return -1;
}
