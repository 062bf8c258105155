// Flash-attention forward for the CogVideoX DiT joint [text ; video] self-attention (gfx950).
//   head_dim 64, bf16 MFMA 32x32x16, fp32 online softmax, non-causal, no mask, N not a tile multiple.
// Replaces F.scaled_dot_product_attention inside diffusers' CogVideoXAttnProcessor2_0, reached from
// /root/reference/inference_script.py:483-489.
//
// Operands come head-major from dove_qkv_post_bf16: Q' [H][Npad][64] (already multiplied by
// scale*log2e), K' [H][Npad][64], V^T [H][64][Npad]; pad rows/columns are zero.
// Workgroup = 4 waves = 128 query rows (32 per wave); KV tiles of 64 keys are staged K and V^T alike with
// 16-byte buffer_load ... lds into XOR-swizzled LDS (per-thread constant offsets, the tile index rides in
// soffset), double-buffered, one barrier per tile, loop unrolled x2 so every ds_read address is a per-lane
// constant + immediate.
// Swapped products keep the softmax lane-local (guide T12): S^T = K Q^T puts one query column in each
// lane (row max/sum = in-lane + one cross-half shuffle), P^T is packed with v_cvt_pk_bf16_f32 and re-laid
// to the MFMA B layout with v_permlane32_swap, and O^T = V^T P^T accumulates with the same query-per-lane
// ownership.  The O rescale is skipped (wave-uniformly) on tiles where no query's running max moved.
// v1 of this kernel was VALU-bound (PMC: 31 VALU instructions per MFMA, software bf16 rounding + 64-bit
// address math); see profiles/r01_pmc_halo_attn.txt.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "../../include/dove_hip.h"

typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ bf16x8 make_frag(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
  u32x4 v = {a, b, c, d};
  return __builtin_bit_cast(bf16x8, v);
}

// QB = query blocks (of 32 rows) per wave.  QB = 2: every K / V^T fragment read from LDS feeds two MFMAs and the two
// independent softmax chains give the scheduler VALU work to hide under the other block's MFMAs.
template <int QB>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const bf16_t* __restrict__ Qh, const bf16_t* __restrict__ Kh,
                                                          const bf16_t* __restrict__ Vt, bf16_t* __restrict__ O,
                                                          long long N, long long Npad, long long ldo) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STAGE = 16384, VOFF = 8192;
  constexpr int QROWS = 32 * QB;                 // query rows per wave
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int h = blockIdx.y;
  const long long q0 = (long long)blockIdx.x * (4 * QROWS) + wave * QROWS;

  bf16x8 qf[QB][4];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    long long qrow = q0 + qb * 32 + l31;
    if (qrow >= Npad) qrow = Npad - 1;           // rows past the padded end are never stored; keep the load in range
    const bf16_t* qp = Qh + ((long long)h * Npad + qrow) * 64 + hi * 8;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[qb][kk] = *(const bf16x8*)(qp + kk * 16);
  }

  f32x16 o[QB][2];
  float m[QB], lsum[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    m[qb] = -1e30f;
    lsum[qb] = 0.f;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[qb][db][r] = 0.f;
  }

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

  // per-lane constant fragment offsets; the V^T tile uses the same (row, chunk) pattern, so its reads are koff + VOFF (immediate)
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
    // ---- S^T[kv][q] = K Q^T : each K fragment feeds QB MFMAs ----
    f32x16 st[QB][2];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) st[qb][kb][r] = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const bf16x8 kf = *(const bf16x8*)(smem + BUF * STAGE + koff[kb][kk]);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) st[qb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[qb][kk], st[qb][kb], 0, 0, 0);
      }
    const long long kv0 = (long long)tile * 64;
    if (kv0 + 64 > N) {
#pragma unroll
      for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const long long kv = kv0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (kv >= N) st[qb][kb][r] = -1e30f;
          }
    }
    // ---- online softmax per query block (base 2; Q carries scale*log2e) ----
    bf16x8 pf[QB][2][2];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      float mt = st[qb][0][0];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) mt = fmaxf(mt, st[qb][kb][r]);
      mt = fmaxf(mt, __shfl_xor(mt, 32));
      if (__any(mt > m[qb])) {   // wave-uniform: rescale only when some query's running max moved
        const float mnew = fmaxf(m[qb], mt);
        const float alpha = __builtin_amdgcn_exp2f(m[qb] - mnew);
        m[qb] = mnew;
        lsum[qb] *= alpha;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[qb][db][r] *= alpha;
      }
      float ps = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = __builtin_amdgcn_exp2f(st[qb][kb][r] - m[qb]);
          st[qb][kb][r] = p;
          ps += p;
        }
      lsum[qb] += ps;
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
    // ---- O^T[d][q] += V^T P^T : each V^T fragment feeds QB MFMAs ----
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
          const bf16x8 vf = *(const bf16x8*)(smem + BUF * STAGE + VOFF + koff[db][kb * 2 + k2]);
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) o[qb][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[qb][kb][k2], o[qb][db], 0, 0, 0);
        }
  };

  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
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

#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
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

extern "C" int dove_attention_fwd_bf16(const void* Qh, const void* Kh, const void* Vt, void* O, long long N,
                                        long long Npad, int heads, int head_dim, long long ldo, void* stream) {
  DOVE_CHECK_ARG(Qh && Kh && Vt && O, "attention_fwd: null pointer");
  DOVE_CHECK_ARG(head_dim == 64, "attention_fwd: head_dim must be 64 (got %d)", head_dim);
  DOVE_CHECK_ARG(N > 0 && Npad % 128 == 0 && Npad >= N && Npad - N < 128, "attention_fwd: Npad must be N rounded up to 128");
  DOVE_CHECK_ARG(Npad * 128 < (1ll << 31), "attention_fwd: sequence too long for 31-bit buffer offsets");
  DOVE_CHECK_ARG(ldo >= (long long)heads * 64 && ldo % 4 == 0, "attention_fwd: bad ldo");
  // QB = 2 (64 queries per wave) measured SLOWER on MI355X (805 vs 845 TFLOP/s at N = 18226: 256 VGPRs + spills), so the
  // one-block-per-wave instantiation is the only one dispatched.
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 32768);
    attr_set = true;
  }
  dim3 grid((unsigned)(Npad / 128), heads);
  hipLaunchKernelGGL(attn_fwd_kernel<1>, grid, dim3(256), 32768, (hipStream_t)stream, (const bf16_t*)Qh,
                     (const bf16_t*)Kh, (const bf16_t*)Vt, (bf16_t*)O, N, Npad, ldo);
  DOVE_CHECK_LAUNCH("dove_attention_fwd_bf16");
  return DOVE_OK;
}
