// MXFP8 attention for the DiT (BASELINE configs[4]: "fp8 MFMA attention/FFN path"; the FFN / projection half is mxfp8.hip).
// Same online-softmax flash attention as attention.hip - head_dim 64, swapped products so the softmax is lane-local - with both
// matrix products on the block-scaled v_mfma_scale_f32_32x32x64_f8f6f4: one instruction covers the whole head dimension of
// S^T = K Q^T (K = 64) and a whole 64-key tile of O^T = V^T P^T, i.e. 4 MX MFMAs per (32 queries x 64 keys) instead of 16 bf16 ones.
//
// Operands (written by dove_qkv_post_mxfp8 from the fused QKV projection, same pre-processing as dove_qkv_post_bf16):
//   Q8 [H][Npad][64] e4m3 of q * qscale * 8, fixed block scale 2^-3;  K8 [H][Npad][64] e4m3 of k, fixed scale 2^0
//     (q and k are LayerNorm(64) outputs: their dynamic range is a few binades by construction, e4m3's relative precision does
//      not need a data-dependent scale there - and constant scales cost no loads in the kernel);
//   V8t [H][64][Npad] e4m3 with ONE E8M0 scale per (d row, 32 consecutive keys) - OCP MX along the contraction dimension; inside a
//     32-key block the keys are stored quad-wise as [0,2,4,6,1,3,5,7] (the order the QK^T MFMA leaves P in, so P needs no permute) -
//     Vs [H][Npad/64][64][2] bytes (tile, d, 32-key block);
//   P is quantised in registers per (query, 64-key tile): p' = 2^(s - m - e), e = ceil(max_tile(s - m)) - 8, so the tile's
//     largest probability lands in (128, 256] and the E8M0 scale 2^e restores it inside the MFMA.  A scale per tile (not one
//     fixed scale against the running max) matters at N = 18k keys: a flat tail of keys at 2^-12 of the max carries more
//     mass than the max itself and would flush to zero under a fixed scale.
// Operand register layout of the 8-bit 32x32x64 MFMA (probed, tools/archive/mxprobe.py): lane (row = l & 31, h = l >> 5) holds K bytes
// [16h, 16h+16) in registers 0-3 and [32+16h, 32+16h+16) in registers 4-7; the scale of 32-block b comes from lanes with h = b.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "../../include/dove_hip.h"

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef int v8i __attribute__((ext_vector_type(8)));

namespace {
__device__ __forceinline__ unsigned e8m0_amax(float amax) {      // smallest power of two s with amax / s <= 448 (mxfp8.hip)
  if (!(amax > 0.f)) return 0u;
  const unsigned u = __float_as_uint(amax * (1.0f / 448.0f));
  int e = (int)((u >> 23) & 0xff);
  if (u & 0x7fffffu) e += 1;
  return (unsigned)(e < 0 ? 0 : (e > 254 ? 254 : e));
}
__device__ __forceinline__ unsigned pack4_fp8(float a, float b, float c, float d) {
  unsigned r = 0;
  r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, r, false);
  r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
  return r;
}
// four probabilities -> four e4m3 bytes of p / scale (scale = 2^e as a float): the MX block scale is folded into the conversion
__device__ __forceinline__ unsigned pack4_fp8_scaled(float a, float b, float c, float d, float scale) {
  typedef short v2s __attribute__((ext_vector_type(2)));
  v2s r = {0, 0};
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, a, b, scale, false);
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, c, d, scale, true);
  return __builtin_bit_cast(unsigned, r);
}
// D = A B + C with C != D (the builtin ties C to D and would copy 16 registers): S - m straight out of the matrix core.
// `s_nop 1`: a VALU write of C (rare rescale path) needs two wait states before an MFMA reads it (attention.hip).
__device__ __forceinline__ f32x16 mx_mfma_c_in(v8i a, v8i b, const f32x16& c, unsigned sa, unsigned sb) {
  f32x16 d;
  asm("s_nop 1\n\tv_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %3, %4, %5 op_sel_hi:[0,0,0]"
      : "=&v"(d) : "v"(a), "v"(b), "v"(c), "v"(sa), "v"(sb));
  return d;
}
}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// qkv_post for the fp8 attention: one lane = one (token, head) 64-vector for q / k (LayerNorm, RoPE, scale, e4m3); the v part of
// a wave (64 tokens) goes through LDS so that lane d owns row d of V^T for those 64 keys = two MX blocks: block maxima, scales
// and a full 64-byte line per lane.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void qkv_post_mx_kernel(const bf16_t* __restrict__ qkv, long long N, long long Npad, int heads,
                                                          int text_len, const float* __restrict__ gq, const float* __restrict__ bq,
                                                          const float* __restrict__ gk, const float* __restrict__ bk,
                                                          const float* __restrict__ cosT, const float* __restrict__ sinT, float qscale,
                                                          float eps, unsigned char* __restrict__ Q8, unsigned char* __restrict__ K8,
                                                          unsigned char* __restrict__ V8t, unsigned char* __restrict__ Vs) {
  __shared__ float vt[4][64][65];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = blockIdx.y;
  const int which = blockIdx.z;  // 0 q, 1 k, 2 v
  const long long n0 = ((long long)blockIdx.x * 4 + wave) * 64;          // first token of this wave = one 64-key tile
  const long long n = n0 + lane;
  if (n0 >= Npad) return;
  const int D = heads * 64;
  float f[64];
  if (n < N) {
    const bf16_t* src = qkv + n * (3LL * D) + (long long)which * D + h * 64;
#pragma unroll
    for (int i = 0; i < 8; ++i) unpack8(*(const uint4*)(src + i * 8), f + i * 8);
  } else {
#pragma unroll
    for (int d = 0; d < 64; ++d) f[d] = 0.f;                             // pad rows / keys are exact zeros
  }
  if (which == 2) {
#pragma unroll
    for (int d = 0; d < 64; ++d) vt[wave][lane][d] = f[d];
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // lane = d: the 64 keys of this tile for row d of V^T
    float v[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) v[j] = vt[wave][j][lane];
    unsigned char* dst = V8t + ((long long)h * 64 + lane) * Npad + n0;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      float amax = 0.f;
#pragma unroll
      for (int j = 0; j < 32; ++j) amax = fmaxf(amax, fabsf(v[b * 32 + j]));
      const unsigned e = e8m0_amax(amax);
      const float inv = __uint_as_float((254u - e) << 23);
      u32x4 o[2];
#pragma unroll
      for (int q4 = 0; q4 < 8; ++q4) {
        // stored quad q4 of the 32-key block holds key quad (q4 & 3) * 2 + (q4 >> 2): the order in which the QK^T MFMA leaves the
        // probabilities in a lane (keys 8i + 4h + j in register i of lane half h), so that P needs no cross-lane exchange
        const float* p = v + b * 32 + ((q4 & 3) * 2 + (q4 >> 2)) * 4;
        o[q4 >> 2][q4 & 3] = pack4_fp8(p[0] * inv, p[1] * inv, p[2] * inv, p[3] * inv);
      }
      *(u32x4*)(dst + b * 32) = o[0];
      *(u32x4*)(dst + b * 32 + 16) = o[1];
      Vs[(((long long)h * (Npad >> 6) + (n0 >> 6)) * 64 + lane) * 2 + b] = (unsigned char)e;
    }
    return;
  }
  if (n >= N) {                                                          // zero rows of the padded tail
    unsigned char* dst = (which == 0 ? Q8 : K8) + ((long long)h * Npad + n) * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) *(u32x4*)(dst + i * 16) = u32x4{0u, 0u, 0u, 0u};
    return;
  }
  const float* gam = which == 0 ? gq : gk;
  const float* bet = which == 0 ? bq : bk;
  float s = 0.f;
#pragma unroll
  for (int d = 0; d < 64; ++d) s += f[d];
  const float mean = s * (1.0f / 64.0f);
  float vv = 0.f;
#pragma unroll
  for (int d = 0; d < 64; ++d) { const float t = f[d] - mean; vv += t * t; }
  const float rstd = rsqrtf(vv * (1.0f / 64.0f) + eps);
#pragma unroll
  for (int d = 0; d < 64; ++d) f[d] = (f[d] - mean) * rstd * gam[d] + bet[d];
  if (n >= text_len && cosT) {
    const float* cr = cosT + (n - text_len) * 64;
    const float* sr = sinT + (n - text_len) * 64;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const float a = f[2 * i], b = f[2 * i + 1];
      f[2 * i] = a * cr[2 * i] - b * sr[2 * i];
      f[2 * i + 1] = b * cr[2 * i + 1] + a * sr[2 * i + 1];
    }
  }
  const float sc = which == 0 ? qscale * 8.0f : 1.0f;                    // Q carries softmax scale * log2(e) * 2^3 (block scale 2^-3)
  unsigned char* dst = (which == 0 ? Q8 : K8) + ((long long)h * Npad + n) * 64;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    u32x4 o;
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const float* p = f + i * 16 + q4 * 4;
      o[q4] = pack4_fp8(p[0] * sc, p[1] * sc, p[2] * sc, p[3] * sc);
    }
    *(u32x4*)(dst + i * 16) = o;
  }
}

extern "C" int dove_qkv_post_mxfp8(const void* qkv, long long N, long long Npad, int heads, int head_dim, int text_len, const float* gq,
                                   const float* bq, const float* gk, const float* bk, const float* cosT, const float* sinT, float qscale,
                                   float eps, void* Q8, void* K8, void* V8t, void* Vs, void* stream) {
  DOVE_CHECK_ARG(qkv && Q8 && K8 && V8t && Vs && gq && bq && gk && bk, "qkv_post_mxfp8: null pointer");
  DOVE_CHECK_ARG(head_dim == 64, "qkv_post_mxfp8: head_dim must be 64 (got %d)", head_dim);
  DOVE_CHECK_ARG(N > 0 && Npad >= N && Npad % 128 == 0, "qkv_post_mxfp8: Npad must be a multiple of 128 and >= N");
  DOVE_CHECK_ARG((cosT == nullptr) == (sinT == nullptr), "qkv_post_mxfp8: cos/sin must both be given or both be null");
  dim3 grid((unsigned)((Npad + 255) / 256), heads, 3);                   // every row of the padded buffers is (re)written
  hipLaunchKernelGGL(qkv_post_mx_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, N, Npad, heads, text_len, gq, bq, gk, bk,
                     cosT, sinT, qscale, eps, (unsigned char*)Q8, (unsigned char*)K8, (unsigned char*)V8t, (unsigned char*)Vs);
  DOVE_CHECK_LAUNCH("dove_qkv_post_mxfp8");
  return DOVE_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// attention: 4 waves x 32 queries; K and V^T tiles of 64 keys = 4 KB each, one LDS-DMA instruction per thread and tile; two tiles
// per barrier (4 stages of 8 KB)
// ------------------------------------------------------------------------------------------------------------------
template <int NW>   // waves per workgroup sharing each K / V^T tile (waves 0-3 stage)
__global__ __launch_bounds__(NW * 64, 2) void attn_fwd_mx_kernel(const unsigned char* __restrict__ Q8, const unsigned char* __restrict__ K8,
                                                             const unsigned char* __restrict__ V8t, const unsigned char* __restrict__ Vs,
                                                             bf16_t* __restrict__ O, long long N, long long Npad, long long ldo, int qblocks) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STAGE = 8192, VOFF = 4096;
  constexpr float THR = 6.0f;                    // rescale when a score exceeds the running max by 2^6
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  // qblocks > 0: 1-D grid, each XCD walks a contiguous range of the head-major tile list (see attention.hip); 0: 2-D grid
  int h, qb;
  if (qblocks > 0) {
    const unsigned t = xcd_remap(blockIdx.x, gridDim.x);
    h = (int)(t / (unsigned)qblocks);
    qb = (int)(t - (unsigned)h * (unsigned)qblocks);
  } else {
    h = blockIdx.y;
    qb = blockIdx.x;
  }
  const long long q0 = (long long)qb * (NW * 32) + wave * 32;

  v8i qf;
  {
    long long qrow = q0 + l31;
    if (qrow >= Npad) qrow = Npad - 1;
    const unsigned char* qp = Q8 + ((long long)h * Npad + qrow) * 64;
    const u32x4 lo = *(const u32x4*)(qp + 16 * hi), hh = *(const u32x4*)(qp + 32 + 16 * hi);
    qf = v8i{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hh[0], (int)hh[1], (int)hh[2], (int)hh[3]};
  }
  f32x16 o[2];
  f32x16 negm;
  float m = 0.f, lsum = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; negm[r] = 0.f; }

  const int ntiles = (int)((N + 63) / 64);
  // staging: thread t moves 16 bytes of row t >> 2 (K: key row, V^T: d row); source chunk XOR-swizzled against (row >> 2) & 3
  const int srow = tid >> 2;
  const int sc_ld = (tid & 3) ^ ((srow >> 2) & 3);
  const auto srd_k = __builtin_amdgcn_make_buffer_rsrc((void*)(K8 + (long long)h * Npad * 64), (short)0, (int)(Npad * 64), 0x00020000);
  const auto srd_v = __builtin_amdgcn_make_buffer_rsrc((void*)(V8t + (long long)h * 64 * Npad), (short)0, (int)(Npad * 64), 0x00020000);
  const auto srd_s = __builtin_amdgcn_make_buffer_rsrc((void*)(Vs + (long long)h * (Npad >> 6) * 128), (short)0, (int)((Npad >> 6) * 128), 0x00020000);
  const unsigned vk = (unsigned)(srow * 64 + sc_ld * 16);
  const unsigned vv = (unsigned)((long long)srow * Npad + sc_ld * 16);
  auto stage = [&](auto bufc, int tile) {
    constexpr int BUF = decltype(bufc)::value;
    if (NW > 4 && wave >= 4) return;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_k, (lds_ptr_t)(smem + BUF * STAGE + wave * 1024), 16, vk, tile * 4096, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_v, (lds_ptr_t)(smem + BUF * STAGE + VOFF + wave * 1024), 16, vv, tile * 64, 0, 0);
  };
  // fragment offsets: row block b (keys 32b.. for K, d rows 32b.. for V^T), this lane's two 16-byte chunks h and 2 + h
  int koff[2][2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int row = b * 32 + l31;
    const int sw = (row >> 2) & 3;
#pragma unroll
    for (int j = 0; j < 2; ++j) koff[b][j] = row * 64 + (((2 * j + hi) ^ sw) << 4);
  }
  auto frag = [&](int base, int b) -> v8i {
    const u32x4 lo = *(const u32x4*)(smem + base + koff[b][0]), hh = *(const u32x4*)(smem + base + koff[b][1]);
    return v8i{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hh[0], (int)hh[1], (int)hh[2], (int)hh[3]};
  };
  // V scale bytes of (tile, d = 32 db + l31, block hi): prefetched one tile pair ahead
  unsigned vs_cur[2][2] = {{0u, 0u}, {0u, 0u}}, vs_nxt[2][2] = {{0u, 0u}, {0u, 0u}};
  auto load_vs = [&](unsigned (&dst)[2][2], int tile) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int db = 0; db < 2; ++db)
        dst[t][db] = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(srd_s, (int)(((db * 32 + l31) * 2 + hi)), (tile + t) * 128, 0);
  };
  const unsigned sQ = 124u, sK = 127u;                         // E8M0: 2^-3 (Q carries a factor 8), 2^0

  auto compute = [&](auto bufc, int tile, const unsigned (&vs)[2]) {
    constexpr int BUF = decltype(bufc)::value;
    // ---- (S - m)^T[kv][q] = K Q^T - m : one MX MFMA per 32-key block covers the whole head dimension, the shift rides in C ----
    f32x16 st[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) st[kb] = mx_mfma_c_in(frag(BUF * STAGE, kb), qf, negm, sK, sQ);
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
    // ---- lazy online softmax (base 2), as attention.hip: m moves only when a score exceeds it by 2^THR ----
    float mt = st[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mt = fmaxf(mt, st[kb][r]);
    {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mt), __float_as_uint(mt), false, false);
      mt = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    const bool first = tile == 0;
    if (first || __any(mt > THR)) {
      const float delta = first ? mt : fmaxf(mt, 0.f);
      const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);
      m += delta;
      mt -= delta;
      lsum *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        o[0][r] *= alpha; o[1][r] *= alpha; st[0][r] -= delta; st[1][r] -= delta; negm[r] = -m;
      }
    }
    // per (query, tile) block scale 2^e of P: the tile's largest p / 2^e lands in (128, 256]
    const int esh = (int)fmaxf(ceilf(mt), -100.f) - 8;          // in [-108, THR - 8]
    const unsigned sP = (unsigned)(127 + esh);
    const float pscale = __uint_as_float(sP << 23);
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
    // P^T as the B operand: slot 16h + 4i + j of each 32-key block = this lane's register i = key 8i + 4h + j; V8t stores its keys in
    // that order (qkv_post_mx), the contraction does not care
    v8i pf;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      unsigned w0 = pack4_fp8_scaled(st[kb][0], st[kb][1], st[kb][2], st[kb][3], pscale);
      unsigned w1 = pack4_fp8_scaled(st[kb][4], st[kb][5], st[kb][6], st[kb][7], pscale);
      unsigned w2 = pack4_fp8_scaled(st[kb][8], st[kb][9], st[kb][10], st[kb][11], pscale);
      unsigned w3 = pack4_fp8_scaled(st[kb][12], st[kb][13], st[kb][14], st[kb][15], pscale);
      pf[kb * 4 + 0] = (int)w0; pf[kb * 4 + 1] = (int)w1; pf[kb * 4 + 2] = (int)w2; pf[kb * 4 + 3] = (int)w3;
    }
    // ---- O^T[d][q] += V^T P^T : one MX MFMA per 32-row block of d covers the 64 keys ----
#pragma unroll
    for (int db = 0; db < 2; ++db)
      o[db] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(frag(BUF * STAGE + VOFF, db), pf, o[db], 0, 0, 0, vs[db], 0, sP);
  };

  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  using B2 = std::integral_constant<int, 2>;
  using B3 = std::integral_constant<int, 3>;
  stage(B0{}, 0);
  if (1 < ntiles) stage(B1{}, 1);
  load_vs(vs_nxt, 0);
  for (int it = 0; it < ntiles; it += 4) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2; ++t) { vs_cur[t][0] = vs_nxt[t][0]; vs_cur[t][1] = vs_nxt[t][1]; }
    if (it + 2 < ntiles) { stage(B2{}, it + 2); load_vs(vs_nxt, it + 2); }
    if (it + 3 < ntiles) stage(B3{}, it + 3);
    compute(B0{}, it, vs_cur[0]);
    if (it + 1 < ntiles) compute(B1{}, it + 1, vs_cur[1]);
    if (it + 2 >= ntiles) break;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2; ++t) { vs_cur[t][0] = vs_nxt[t][0]; vs_cur[t][1] = vs_nxt[t][1]; }
    if (it + 4 < ntiles) { stage(B0{}, it + 4); load_vs(vs_nxt, it + 4); }
    if (it + 5 < ntiles) stage(B1{}, it + 5);
    compute(B2{}, it + 2, vs_cur[0]);
    if (it + 3 < ntiles) compute(B3{}, it + 3, vs_cur[1]);
  }

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

extern "C" int dove_attention_fwd_mxfp8(const void* Q8, const void* K8, const void* V8t, const void* Vs, void* O, long long N, long long Npad,
                                        int heads, int head_dim, long long ldo, void* stream) {
  DOVE_CHECK_ARG(Q8 && K8 && V8t && Vs && O, "attention_fwd_mxfp8: null pointer");
  DOVE_CHECK_ARG(head_dim == 64, "attention_fwd_mxfp8: head_dim must be 64 (got %d)", head_dim);
  DOVE_CHECK_ARG(N > 0 && Npad % 128 == 0 && Npad >= N && Npad - N < 128, "attention_fwd_mxfp8: Npad must be N rounded up to 128");
  DOVE_CHECK_ARG(Npad * 64 < (1ll << 31), "attention_fwd_mxfp8: sequence too long for 31-bit buffer offsets");
  DOVE_CHECK_ARG(ldo >= (long long)heads * 64 && ldo % 4 == 0, "attention_fwd_mxfp8: bad ldo");
  constexpr int LDS = 4 * 8192;
  constexpr int NW = 4;
  const int qblocks = (int)((Npad + NW * 32 - 1) / (NW * 32));
  DOVE_CHECK_ARG((long long)qblocks * heads < (1ll << 31), "attention_fwd_mxfp8: grid too large");
  hipLaunchKernelGGL(attn_fwd_mx_kernel<NW>, dim3((unsigned)(qblocks * heads)), dim3(NW * 64), LDS, (hipStream_t)stream, (const unsigned char*)Q8,
                     (const unsigned char*)K8, (const unsigned char*)V8t, (const unsigned char*)Vs, (bf16_t*)O, N, Npad, ldo, qblocks);
  DOVE_CHECK_LAUNCH("dove_attention_fwd_mxfp8");
  return DOVE_OK;
}

#ifdef DOVE_TIMING_BUILD
extern "C" int dove_attention_fwd_mxfp8_nw(const void* Q8, const void* K8, const void* V8t, const void* Vs, void* O, long long N, long long Npad,
                                           int heads, long long ldo, int nw, void* stream) {
  constexpr int LDS = 4 * 8192;
  const int w = nw == 14 ? 4 : nw;
  const int qb = (int)((Npad + w * 32 - 1) / (w * 32));
  dim3 grid((unsigned)qb, heads);
#define MXL(W, G, QB) hipLaunchKernelGGL(attn_fwd_mx_kernel<W>, G, dim3(W * 64), LDS, (hipStream_t)stream, (const unsigned char*)Q8, (const unsigned char*)K8, \
                                         (const unsigned char*)V8t, (const unsigned char*)Vs, (bf16_t*)O, N, Npad, ldo, QB)
  if (nw == 4) MXL(4, grid, 0); else if (nw == 6) MXL(6, grid, 0); else if (nw == 8) MXL(8, grid, 0);
  else if (nw == 14) MXL(4, dim3((unsigned)(qb * heads)), qb);      // the product mapping: XCD-contiguous 1-D grid
  else return -1;
  DOVE_CHECK_LAUNCH("dove_attention_fwd_mxfp8_nw");
  return DOVE_OK;
}
#endif
