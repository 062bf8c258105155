// Streaming GroupNorm-apply + SiLU variants (within-run A/B; never loaded by dove_amd): how close to the HBM rate can a
// 2 B read + 2 B write per element pass get on MI355X, and what does it take.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef unsigned short bf16_t;
__device__ __forceinline__ float bf2f(uint32_t h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ uint32_t pk(float a, float b) {
  typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
  v2bf r = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint4 pack8(const float* f) { return uint4{pk(f[0], f[1]), pk(f[2], f[3]), pk(f[4], f[5]), pk(f[6], f[7])}; }
__device__ __forceinline__ float silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f)); }

// flat: the tensor is n16 16-byte chunks; chunk i holds channels (i & (cpp-1))*8 .. +8 of pixel i >> cpp_log
template <int U, bool NT, bool COPY>
__global__ __launch_bounds__(256) void gn_flat(const uint4* __restrict__ x, uint4* __restrict__ y, long long n16, int cpp_log,
                                               const float* __restrict__ sc, const float* __restrict__ sh) {
  const long long stride = (long long)gridDim.x * 256;
  const int q = threadIdx.x & ((1 << cpp_log) - 1);          // stride is a multiple of cpp: a thread's channel chunk is fixed
  float s[8], h[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s[e] = sc[q * 8 + e]; h[e] = sh[q * 8 + e]; }
  for (long long i0 = (long long)blockIdx.x * 256 + threadIdx.x; i0 < n16; i0 += U * stride) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const long long i = i0 + u * stride; if (i < n16) v[u] = x[i]; }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long i = i0 + u * stride;
      if (i < n16) {
        uint4 o = v[u];
        if (!COPY) {
          float f[8];
          unpack8(v[u], f);
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] = silu(f[e] * s[e] + h[e]);
          o = pack8(f);
        }
        typedef unsigned int u4v __attribute__((ext_vector_type(4)));
        if (NT) __builtin_nontemporal_store(u4v{o.x, o.y, o.z, o.w}, (u4v*)(y + i)); else y[i] = o;
      }
    }
  }
}

extern "C" int gn_exp(int variant, const void* x, void* y, long long n16, int cpp_log, const float* sc, const float* sh, int blocks, void* stream) {
  hipStream_t s = (hipStream_t)stream;
#define L(U, NT, CP) hipLaunchKernelGGL((gn_flat<U, NT, CP>), dim3(blocks), dim3(256), 0, s, (const uint4*)x, (uint4*)y, n16, cpp_log, sc, sh)
  switch (variant) {
    case 0: L(4, false, true); break;
    case 1: L(1, false, false); break;
    case 2: L(2, false, false); break;
    case 3: L(4, false, false); break;
    case 4: L(8, false, false); break;
    case 5: L(4, true, false); break;
    case 6: L(8, true, false); break;
    case 7: L(8, false, true); break;
    case 8: L(8, true, true); break;
    default: return -1;
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
