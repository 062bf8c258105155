// EXPERIMENT (tools/archive/attn_ab.py, variants 30+): software-pipelined flash attention.  Within ONE wave the QK^T MFMAs of tile j+1 are
// issued under the exponentials of tile j and the PV MFMAs of tile j under the row sums of tile j and the max-reduction of tile
// j+1; the interleave is pinned with sched_barrier fences (the compiler otherwise orders the phases serially).  Operand layout and
// numerics contract are those of dove_attention_fwd_bf16 (lazy rescale, -m through the MFMA C operand).
#include <stdlib.h>

#include <type_traits>

#include "../../dove_amd/csrc/common.h"

typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ bf16x8 make_frag(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  u32x4 v = {a, b, c, d};
  return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ f32x16 mfma_c_in(bf16x8 a, bf16x8 b, const f32x16& c) {
  f32x16 d;
  asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
#define FENCE() __builtin_amdgcn_sched_barrier(0)

// VAR bit 0: row sums in phase A (with the exps) instead of phase B; bit 1: no fences (compiler order, for comparison)
template <int VAR>
__global__ __launch_bounds__(256, 2) void attn_pipe_kernel(const bf16_t* __restrict__ Qh, const bf16_t* __restrict__ Kh,
                                                           const bf16_t* __restrict__ Vt, bf16_t* __restrict__ O, long long N,
                                                           long long Npad, long long ldo) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STAGE = 16384, VOFF = 8192, K0OFF = 4 * STAGE;
  constexpr float THR = 6.0f;
  constexpr bool SUMA = VAR & 1, NOFENCE = VAR & 2;
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
  f32x16 o[2], negm, stA[2], stB[2];
  float m = 0.f, lsum = 0.f, mt_cur = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; negm[r] = 0.f; }

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
  auto stage_k = [&](int ldsoff, int tile) {       // K tile -> smem + ldsoff (wave-uniform)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_k, (lds_ptr_t)(smem + ldsoff + (j * 256 + wave * 64) * 16), 16, vk[j], tile * (64 * 128), 0, 0);
  };
  auto stage_v = [&](int ldsoff, int tile) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_v, (lds_ptr_t)(smem + ldsoff + (j * 256 + wave * 64) * 16), 16, vv[j], tile * (64 * 2), 0, 0);
  };
  // unit u = {K tile u+1, V^T tile u} in ring buffer u & 3: what the pipelined body of tile u consumes
  auto stage_unit = [&](int u) {
    const int base = (u & 3) * STAGE;
    if (u + 1 < ntiles) stage_k(base, u + 1);
    stage_v(base + VOFF, u);
  };
  int koff[2][4];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int row = b * 32 + l31;
    const int sw = (row >> 1) & 7;
#pragma unroll
    for (int c = 0; c < 4; ++c) koff[b][c] = row * 128 + (((c * 2 + hi) ^ sw) << 4);
  }
  auto rowmax = [&](const f32x16 (&st)[2]) -> float {
    float mt = st[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mt = fmaxf(mt, st[kb][r]);
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mt), __float_as_uint(mt), false, false);
    return fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
  };
  auto rescale = [&](f32x16 (&cur)[2], bool first) {
    if (first || __any(mt_cur > THR)) {
      const float delta = first ? mt_cur : fmaxf(mt_cur, 0.f);
      const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);
      m += delta;
      lsum *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        o[0][r] *= alpha; o[1][r] *= alpha; cur[0][r] -= delta; cur[1][r] -= delta; negm[r] = -m;
      }
    }
  };
  // P^T fragments of one 32-key block from its 16 exponentiated scores
  auto pfrag = [&](const f32x16& p, bf16x8 (&pf)[2]) {
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
      const int b = 8 * k2;
      const uint32_t a0 = pack_bf2(p[b + 0], p[b + 1]), a1 = pack_bf2(p[b + 2], p[b + 3]);
      const uint32_t b0 = pack_bf2(p[b + 4], p[b + 5]), b1 = pack_bf2(p[b + 6], p[b + 7]);
      const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
      const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
      pf[k2] = make_frag(r0[0], r1[0], r0[1], r1[1]);
    }
  };

  // ---- pipelined body of tile j (< ntiles - 1): consumes cur = S(j) - m, produces nxt = S(j+1) - m and its row max ----
  auto body = [&](int j, f32x16 (&cur)[2], f32x16 (&nxt)[2]) {
    const int kbase = (j & 3) * STAGE;
    rescale(cur, j == 0);
    if (!NOFENCE) FENCE();
    auto kfrag = [&](int kb, int kk) { return *(const bf16x8*)(smem + kbase + koff[kb][kk]); };
    auto vfrag = [&](int db, int c) { return *(const bf16x8*)(smem + kbase + VOFF + koff[db][c]); };
    bf16x8 ka[4], kb4[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) ka[kk] = kfrag(0, kk);
    float ps = 0.f;
    auto exps = [&](int kb, int kk) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = kk * 4 + e;
        const float p = __builtin_amdgcn_exp2f(cur[kb][r]);
        cur[kb][r] = p;
        if (SUMA) ps += p;
      }
    };
    // phase A: 8 QK^T MFMAs of tile j+1, four exponentials of tile j behind each
    nxt[0] = mfma_c_in(ka[0], qf[0], negm);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) kb4[kk] = kfrag(1, kk);        // second key block's fragments: in flight under the first chain
    exps(0, 0);
    if (!NOFENCE) FENCE();
#pragma unroll
    for (int kk = 1; kk < 4; ++kk) {
      nxt[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[kk], qf[kk], nxt[0], 0, 0, 0);
      exps(0, kk);
      if (!NOFENCE) FENCE();
    }
    bf16x8 v0[2], v1[2];
    nxt[1] = mfma_c_in(kb4[0], qf[0], negm);
    v0[0] = vfrag(0, 0); v0[1] = vfrag(1, 0);                     // V^T fragments of tile j for the first PV MFMAs
    exps(1, 0);
    if (!NOFENCE) FENCE();
    nxt[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb4[1], qf[1], nxt[1], 0, 0, 0);
    v1[0] = vfrag(0, 1); v1[1] = vfrag(1, 1);
    exps(1, 1);
    if (!NOFENCE) FENCE();
    nxt[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb4[2], qf[2], nxt[1], 0, 0, 0);
    exps(1, 2);
    if (!NOFENCE) FENCE();
    nxt[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb4[3], qf[3], nxt[1], 0, 0, 0);
    exps(1, 3);
    bf16x8 pf0[2], pf1[2];
    pfrag(cur[0], pf0);
    if (!NOFENCE) FENCE();
    // phase B: 8 PV MFMAs of tile j; behind them the second half of P, the row sum of tile j and the row max of tile j+1
    float mt = nxt[0][0];
    bf16x8 v2[2], v3[2];
    o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0[0], pf0[0], o[0], 0, 0, 0);
    v2[0] = vfrag(0, 2); v2[1] = vfrag(1, 2);
    pfrag(cur[1], pf1);
    if (!NOFENCE) FENCE();
    o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0[1], pf0[0], o[1], 0, 0, 0);
    v3[0] = vfrag(0, 3); v3[1] = vfrag(1, 3);
    if (!SUMA) {
#pragma unroll
      for (int r = 0; r < 8; ++r) ps += cur[0][r];
    }
    if (!NOFENCE) FENCE();
    o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1[0], pf0[1], o[0], 0, 0, 0);
    if (!SUMA) {
#pragma unroll
      for (int r = 8; r < 16; ++r) ps += cur[0][r];
    }
    if (!NOFENCE) FENCE();
    o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1[1], pf0[1], o[1], 0, 0, 0);
    if (!SUMA) {
#pragma unroll
      for (int r = 0; r < 8; ++r) ps += cur[1][r];
    }
    if (!NOFENCE) FENCE();
    o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v2[0], pf1[0], o[0], 0, 0, 0);
    if (!SUMA) {
#pragma unroll
      for (int r = 8; r < 16; ++r) ps += cur[1][r];
    }
    if (!NOFENCE) FENCE();
    o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v2[1], pf1[0], o[1], 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) mt = fmaxf(mt, nxt[0][r]);
    if (!NOFENCE) FENCE();
    o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v3[0], pf1[1], o[0], 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) mt = fmaxf(mt, nxt[1][r]);
    if (!NOFENCE) FENCE();
    o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v3[1], pf1[1], o[1], 0, 0, 0);
    lsum += ps;
    {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mt), __float_as_uint(mt), false, false);
      mt_cur = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    if (!NOFENCE) FENCE();
  };
  // ---- last tile: no successor; keys past N are masked here (their K rows are zero: they only ever polluted the row max upwards) ----
  auto last = [&](int j, f32x16 (&cur)[2]) {
    rescale(cur, j == 0);
    const long long kv0 = (long long)j * 64;
    if (kv0 + 64 > N) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const long long kv = kv0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (kv >= N) cur[kb][r] = -1e30f;
        }
    }
    float ps = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) { const float p = __builtin_amdgcn_exp2f(cur[kb][r]); cur[kb][r] = p; ps += p; }
    lsum += ps;
    bf16x8 pf[2][2];
    pfrag(cur[0], pf[0]);
    pfrag(cur[1], pf[1]);
    const int vb = (j & 3) * STAGE + VOFF;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
          const bf16x8 vfr = *(const bf16x8*)(smem + vb + koff[db][kb * 2 + k2]);
          o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr, pf[kb][k2], o[db], 0, 0, 0);
        }
  };

  stage_k(K0OFF, 0);
  stage_unit(0);
  if (1 < ntiles) stage_unit(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // S(0): the first tile's K sits in its own 8 KB slot
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const bf16x8 kfr = *(const bf16x8*)(smem + K0OFF + koff[kb][kk]);
      if (kk == 0) stA[kb] = mfma_c_in(kfr, qf[kk], negm);
      else stA[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr, qf[kk], stA[kb], 0, 0, 0);
    }
  mt_cur = rowmax(stA);
  const int nlast = ntiles - 1;
  for (int j = 0; j < nlast; j += 2) {                           // two tiles per barrier
    if (j + 2 < ntiles) stage_unit(j + 2);
    if (j + 3 < ntiles) stage_unit(j + 3);
    body(j, stA, stB);
    if (j + 1 < nlast) body(j + 1, stB, stA);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (nlast & 1) last(nlast, stB); else last(nlast, stA);

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

template <int VAR>
static int launch_pipe(const void* Qh, const void* Kh, const void* Vt, void* O, long long N, long long Npad, int heads, long long ldo, hipStream_t s) {
  constexpr int lds = 4 * 16384 + 8192;
  (void)hipFuncSetAttribute((const void*)attn_pipe_kernel<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  dim3 grid((unsigned)(Npad / 128), heads);
  hipLaunchKernelGGL((attn_pipe_kernel<VAR>), grid, dim3(256), lds, s, (const bf16_t*)Qh, (const bf16_t*)Kh, (const bf16_t*)Vt, (bf16_t*)O, N, Npad, ldo);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" void dove_set_error(const char*, ...) {}
extern "C" int attn_exp3(int variant, const void* Qh, const void* Kh, const void* Vt, void* O, long long N, long long Npad, int heads,
                         long long ldo, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  switch (variant) {
    case 30: return launch_pipe<0>(Qh, Kh, Vt, O, N, Npad, heads, ldo, s);   // fenced interleave, sums in phase B
    case 31: return launch_pipe<1>(Qh, Kh, Vt, O, N, Npad, heads, ldo, s);   // fenced, sums with the exps
    case 32: return launch_pipe<2>(Qh, Kh, Vt, O, N, Npad, heads, ldo, s);   // same source order, compiler schedule
  }
  return -1;
}
