// Operators of the T5 text encoder (CogVideoX's `text_encoder`, a T5-v1.1-XXL encoder) that the DiT-side kernels do not cover.
// DOVE reaches it only for a non-empty prompt: `pipe.text_encoder(prompt_token_ids)[0]` at
// /root/reference/inference_script.py:429-444 (226 tokens, no attention mask); the default run uses the cached
// empty-prompt embedding and never calls it.  Linears run on dove_conv_igemm_bf16.
//   dove_rmsnorm_bf16          T5LayerNorm: y = w * x * rsqrt(mean(x^2) + eps), fp32 statistics, no mean subtraction, no bias
//   dove_gated_gelu_bf16       T5DenseGatedActDense: gelu_new(wi_0 x) * (wi_1 x) on the fused [M][2F] projection
//   dove_attention_bias_bf16   T5Attention: softmax(q k^T + position_bias) v  - NO 1/sqrt(d) scaling, additive fp32 bias
//                              [H][N][N] (relative-position buckets, shared by all layers), short sequences (N <= 1024)
#include "common.h"
#include "../../include/dove_hip.h"

template <int NIT>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, long long rows, int D,
                                                      float eps, const float* __restrict__ w) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long row = (long long)blockIdx.x * 4 + wave;
  if (row >= rows) return;
  const bf16_t* xr = x + row * D;
  float f[NIT][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int c0 = (i * 64 + lane) * 8;
    if (c0 < D) {
      unpack8(*(const uint4*)(xr + c0), f[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += f[i][e] * f[i][e];
    }
  }
  const float rstd = rsqrtf(wave_sum(s) / (float)D + eps);
  bf16_t* yr = y + row * D;
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int c0 = (i * 64 + lane) * 8;
    if (c0 < D) {
      const f32x4 w0 = *(const f32x4*)(w + c0), w1 = *(const f32x4*)(w + c0 + 4);
      float o[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { o[e] = f[i][e] * rstd * w0[e]; o[4 + e] = f[i][4 + e] * rstd * w1[e]; }
      *(uint4*)(yr + c0) = pack8(o);
    }
  }
}

extern "C" int dove_rmsnorm_bf16(const void* x, void* y, long long rows, int D, float eps, const float* weight, void* stream) {
  DOVE_CHECK_ARG(x && y && weight, "rmsnorm: null pointer");
  DOVE_CHECK_ARG(D % 8 == 0 && D > 0 && D <= 4096, "rmsnorm: D (%d) must be a multiple of 8, <= 4096", D);
  DOVE_CHECK_ARG(rows > 0, "rmsnorm: empty input");
  const unsigned grid = (unsigned)((rows + 3) / 4);
  hipStream_t s = (hipStream_t)stream;
  const int nit = (D / 8 + 63) / 64;
#define RMS_LAUNCH(N) hipLaunchKernelGGL((rmsnorm_kernel<N>), dim3(grid), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, rows, D, eps, weight)
  switch (nit) {
    case 1: RMS_LAUNCH(1); break;
    case 2: RMS_LAUNCH(2); break;
    case 3: RMS_LAUNCH(3); break;
    case 4: RMS_LAUNCH(4); break;
    case 5: RMS_LAUNCH(5); break;
    case 6: RMS_LAUNCH(6); break;
    case 7: RMS_LAUNCH(7); break;
    default: RMS_LAUNCH(8); break;
  }
#undef RMS_LAUNCH
  DOVE_CHECK_LAUNCH("dove_rmsnorm_bf16");
  return DOVE_OK;
}

__global__ __launch_bounds__(256) void gated_gelu_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, long long M, int F) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const int per_row = F >> 3;
  const long long row = t / per_row;
  if (row >= M) return;
  const int c0 = (int)(t - row * per_row) * 8;
  float a[8], b[8], o[8];
  unpack8(*(const uint4*)(x + row * 2 * F + c0), a);
  unpack8(*(const uint4*)(x + row * 2 * F + F + c0), b);
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = bf2f(f2bf(gelu_tanh_f(a[e]))) * b[e];      // the reference rounds act(wi_0 x) to bf16 before the product
  *(uint4*)(y + row * F + c0) = pack8(o);
}

extern "C" int dove_gated_gelu_bf16(const void* x, void* y, long long M, int F, void* stream) {
  DOVE_CHECK_ARG(x && y, "gated_gelu: null pointer");
  DOVE_CHECK_ARG(M > 0 && F > 0 && F % 8 == 0, "gated_gelu: F (%d) must be a positive multiple of 8", F);
  const long long n = M * (F >> 3);
  hipLaunchKernelGGL(gated_gelu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                     (bf16_t*)y, M, F);
  DOVE_CHECK_LAUNCH("dove_gated_gelu_bf16");
  return DOVE_OK;
}

// one wave per (query, head): lanes own keys j = lane, lane + 64, ... for the scores and one output dim for P V
__global__ __launch_bounds__(256) void attn_bias_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                        const bf16_t* __restrict__ v, long long ld, const float* __restrict__ bias,
                                                        bf16_t* __restrict__ out, long long ldo, int N) {
  extern __shared__ float psm[];                       // [4 waves][N]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = blockIdx.y;
  const int i = blockIdx.x * 4 + wave;
  if (i >= N) return;                                  // waves are independent: no block-level sync below
  float* p = psm + wave * N;
  float qf[64];
  const bf16_t* qp = q + (long long)i * ld + h * 64;
#pragma unroll
  for (int c = 0; c < 8; ++c) unpack8(*(const uint4*)(qp + c * 8), qf + c * 8);
  const float* brow = bias + ((long long)h * N + i) * N;
  float mx = -3.0e38f;
  for (int j = lane; j < N; j += 64) {
    const bf16_t* kp = k + (long long)j * ld + h * 64;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float kf[8];
      unpack8(*(const uint4*)(kp + c * 8), kf);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += qf[c * 8 + e] * kf[e];
    }
    s = bf2f(f2bf(s)) + bf2f(f2bf(brow[j]));           // bf16 scores + bf16 position bias, like the reference's bf16 modules
    s = bf2f(f2bf(s));
    p[j] = s;
    mx = fmaxf(mx, s);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float sum = 0.f;
  for (int j = lane; j < N; j += 64) {
    const float e = __expf(p[j] - mx);
    p[j] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own LDS writes are visible to all its lanes
  float acc = 0.f;
  for (int j = 0; j < N; ++j) acc += bf2f(f2bf(p[j] * inv)) * bf2f(v[(long long)j * ld + h * 64 + lane]);   // softmax output cast to bf16
  out[(long long)i * ldo + h * 64 + lane] = f2bf(acc);
}

extern "C" int dove_attention_bias_bf16(const void* q, const void* k, const void* v, long long ld, const float* bias, void* out,
                                         long long ldo, int N, int heads, int head_dim, void* stream) {
  DOVE_CHECK_ARG(q && k && v && bias && out, "attention_bias: null pointer");
  DOVE_CHECK_ARG(head_dim == 64, "attention_bias: head_dim must be 64 (got %d)", head_dim);
  DOVE_CHECK_ARG(N > 0 && N <= 1024 && heads > 0, "attention_bias: N (%d) must be in 1..1024", N);
  DOVE_CHECK_ARG(ld % 8 == 0 && ldo >= (long long)heads * 64, "attention_bias: bad leading dimension");
  dim3 grid((unsigned)((N + 3) / 4), (unsigned)heads);
  hipLaunchKernelGGL(attn_bias_kernel, grid, dim3(256), (size_t)4 * N * sizeof(float), (hipStream_t)stream, (const bf16_t*)q,
                     (const bf16_t*)k, (const bf16_t*)v, ld, bias, (bf16_t*)out, ldo, N);
  DOVE_CHECK_LAUNCH("dove_attention_bias_bf16");
  return DOVE_OK;
}
