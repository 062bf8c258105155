// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the DOVE hot path.
// wave = 64 lanes everywhere; bf16 is carried as raw uint16_t bit patterns.
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) short bf16x8;   // one MFMA A/B fragment (8 bf16 = 4 VGPRs)
typedef __attribute__((ext_vector_type(16))) float f32x16;  // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define DOVE_OK 0
#define DOVE_EINVAL (-1)
#define DOVE_ELAUNCH (-2)

extern "C" void dove_set_error(const char* fmt, ...);

#define DOVE_CHECK_ARG(cond, ...)        \
  do {                                   \
    if (!(cond)) {                       \
      dove_set_error(__VA_ARGS__);       \
      return DOVE_EINVAL;                \
    }                                    \
  } while (0)

#define DOVE_CHECK_LAUNCH(name)                                           \
  do {                                                                    \
    hipError_t e__ = hipGetLastError();                                   \
    if (e__ != hipSuccess) {                                              \
      dove_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return DOVE_ELAUNCH;                                                \
    }                                                                     \
  } while (0)

// hipFuncSetAttribute (dynamic LDS above 64 KB) applies to the CURRENT device: one flag per call site AND device, so a process that
// drives several GPUs (dove_create(device) invites it) raises the limit on each of them
// One limit for every per-device table of the library (this flag, igemm.hip's zero page and CU count).
constexpr int DOVE_MAX_DEVICES = 32;
struct PerDeviceOnce {
  std::atomic<bool> done[DOVE_MAX_DEVICES] = {};
  // Usage:  if (auto once = flag.guard()) { hipFuncSetAttribute(...); }
  // The guard tests true for EVERY caller until one of them has LEFT the block (its destructor publishes), i.e. until the attribute is known
  // to be set on this device: a second host thread that arrives while the first is still inside hipFuncSetAttribute sets the attribute itself
  // (twice is harmless) instead of launching a > 64 KB-LDS kernel before the limit is raised (ranks-as-threads on one GPU:
  // tests/test_graph_gpu.py drives the library that way).
  struct Guard {
    std::atomic<bool>* flag;                 // nullptr: already published (or no table slot for this device: always set, never publish)
    bool todo;
    explicit operator bool() const { return todo; }
    ~Guard() { if (todo && flag) flag->store(true, std::memory_order_release); }
  };
  Guard guard() {
    int d = 0;
    (void)hipGetDevice(&d);
    if (d < 0 || d >= DOVE_MAX_DEVICES) return Guard{nullptr, true};
    return Guard{&done[d], !done[d].load(std::memory_order_acquire)};
  }
};

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even, NaN preserved (same rounding as torch's float->bfloat16)
__device__ __forceinline__ bf16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

// two fp32 -> packed bf16x2 with ONE v_cvt_pk_bf16_f32 (round-to-nearest-even in hardware on gfx950)
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) float f32x2_t;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}

__device__ __forceinline__ uint4 pack8(const float* f) {
  uint4 v;
  v.x = pack_bf2(f[0], f[1]); v.y = pack_bf2(f[2], f[3]);
  v.z = pack_bf2(f[4], f[5]); v.w = pack_bf2(f[6], f[7]);
  return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// x * sigmoid(x) with the hardware reciprocal (1 ulp): an IEEE fp32 division is ~10 VALU instructions per element, and the result
// is rounded to bf16 right after
__device__ __forceinline__ float silu_f(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}

// 0.5*x*(1+tanh(u)) == x*sigmoid(2u), u = sqrt(2/pi)*(x + 0.044715 x^3)
__device__ __forceinline__ float gelu_tanh_f(float x) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  const float u = k0 * (x + k1 * x * x * x);
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u * -2.8853900817779268f));
}

// async 16-byte global -> LDS copy: LDS destination is wave-uniform base + lane*16, the global
// source address is per-lane (cdna_hip_programming.md section 5)
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// XCD-aware, bijective block remap: the dispatcher places block b on XCD b%8; give every XCD a
// contiguous range of logical tiles so neighbouring tiles share one L2 (guide T1).
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
  const unsigned q = nwg >> 3, r = nwg & 7u, xcd = bid & 7u, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
