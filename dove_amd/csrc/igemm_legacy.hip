// The generic implicit-GEMM kernel (v1): per-lane 64-bit gather addresses, zero page for padding, global_load_lds staging.  It carries
// what no later kernel does - the nearest-x2-upsample-fused convs too small for the LDS-halo tile (test-sized clips only; no production
// shape: tests/test_abi.py pins that) - and is kept out of igemm.hip so that file holds the product kernels only.
#include "igemm_args.h"

template <int BN, int BK>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmArgs a) {
  constexpr int BM = 128;
  constexpr int CPR = BK / 8;                  // 16-byte chunks per tile row
  constexpr int CPR_LOG = (BK == 64) ? 3 : 2;
  constexpr int RPG = 256 / CPR;               // tile rows covered by one 256-lane glds pass
  constexpr int NA = BM / RPG;                 // A passes per K-step
  constexpr int B_SLOTS = BN * CPR;
  constexpr int NB = (B_SLOTS + 255) / 256;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
  constexpr int WN = (BN >= 128) ? 2 : 1, WM = 4 / WN;
  constexpr int PT = (BM / WM) / 32, CT = (BN / WN) / 32;
  constexpr int KK = BK / 16;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  unsigned rest = xcd_remap(blockIdx.x, gridDim.x);
  const int tn = rest % a.tiles_n; rest /= a.tiles_n;
  const int twi = rest % a.tiles_w; rest /= a.tiles_w;
  const int thi = rest % a.tiles_h;
  const int t = rest / a.tiles_h;
  const int n0 = tn * BN;
  const int TWm = (1 << a.tw_log2) - 1;
  const int oh0 = thi * (128 >> a.tw_log2), ow0 = twi << a.tw_log2;
  const int H_eff = a.H_in << a.up, W_eff = a.W_in << a.up;

  // ---- per-thread staging geometry (fixed for the whole K loop) ----
  const int cs = tid & (CPR - 1);
  const int rsub = tid >> CPR_LOG;
  const int c = (BK == 64) ? (cs ^ ((rsub >> 1) & 7)) : (cs ^ ((rsub >> 2) & 3));
  int ih0[NA], iw0[NA];
  bool mval[NA];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int m = j * RPG + rsub;
    const int oh = oh0 + (m >> a.tw_log2), ow = ow0 + (m & TWm);
    mval[j] = (oh < a.H_out) && (ow < a.W_out);
    ih0[j] = oh * a.stride - a.pad_h;
    iw0[j] = ow * a.stride - a.pad_w;
  }
  const long long frame_elems = (long long)a.H_in * a.W_in * a.Cin;
  const int kc_per_tap = a.Cin / BK;
  const int nk = a.kt * a.kh * a.kw * kc_per_tap;

  auto stage = [&](int buf, int tap, int kc) {
    const int dw = tap % a.kw;
    const int dh = (tap / a.kw) % a.kh;
    const int dt = tap / (a.kw * a.kh);
    const bf16_t* fp = igemm_src_frame(a, t, dt, frame_elems);
    const int k0 = kc * BK + c * 8;
    char* As = smem + buf * STAGE;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int ih = ih0[j] + dh, iw = iw0[j] + dw;
      const bool ok = mval[j] && ((unsigned)ih < (unsigned)H_eff) && ((unsigned)iw < (unsigned)W_eff);
      const long long off = ((long long)(ih >> a.up) * a.W_in + (iw >> a.up)) * a.Cin + k0;
      const bf16_t* src = ok ? (fp + off) : a.zero;
      glds16(src, As + (j * 256 + wave * 64) * 16);
    }
    char* Bs = As + A_BYTES;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      if (j * 256 + wave * 64 < B_SLOTS) {
        const int row = j * RPG + rsub;
        const bf16_t* src = a.w + ((long long)tap * a.Cout_pad + n0 + row) * a.Cin + k0;
        glds16(src, Bs + (j * 256 + wave * 64) * 16);
      }
    }
  };

  f32x16 acc[CT][PT];
#pragma unroll
  for (int i = 0; i < CT; ++i)
#pragma unroll
    for (int p = 0; p < PT; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][p][r] = 0.f;

  const int hi = lane >> 5, l31 = lane & 31;

  stage(0, 0, 0);
  int tap_n = 0, kc_n = 1;  // coordinates of the NEXT K-step to stage
  if (kc_n == kc_per_tap) { kc_n = 0; tap_n = 1; }

  for (int it = 0; it < nk; ++it) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (it + 1 < nk) {
      stage((it + 1) & 1, tap_n, kc_n);
      if (++kc_n == kc_per_tap) { kc_n = 0; ++tap_n; }
    }
    const char* As = smem + (it & 1) * STAGE;
    const char* Bs = As + A_BYTES;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      const int chunk = kk * 2 + hi;
      bf16x8 xf[PT], wf[CT];
#pragma unroll
      for (int p = 0; p < PT; ++p) {
        const int row = wm * (BM / WM) + p * 32 + l31;
        const int sc = (BK == 64) ? (chunk ^ ((row >> 1) & 7)) : (chunk ^ ((row >> 2) & 3));
        xf[p] = *(const bf16x8*)(As + row * (BK * 2) + sc * 16);
      }
#pragma unroll
      for (int i = 0; i < CT; ++i) {
        const int row = wn * (BN / WN) + i * 32 + l31;
        const int sc = (BK == 64) ? (chunk ^ ((row >> 1) & 7)) : (chunk ^ ((row >> 2) & 3));
        wf[i] = *(const bf16x8*)(Bs + row * (BK * 2) + sc * 16);
      }
#pragma unroll
      for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int p = 0; p < PT; ++p)
          acc[i][p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], xf[p], acc[i][p], 0, 0, 0);
    }
  }

  // ---- epilogue: lane holds pixel (lane&31), channels 8*g + 4*(lane>>5) + {0..3} of each 32x32 tile ----
#pragma unroll
  for (int p = 0; p < PT; ++p) {
    const int m = wm * (BM / WM) + p * 32 + l31;
    const int oh = oh0 + (m >> a.tw_log2), ow = ow0 + (m & TWm);
    if (!((oh < a.H_out) && (ow < a.W_out))) continue;
    const long long pix = ((long long)t * a.H_out + oh) * a.W_out + ow;
    const float* gate = a.gate ? (a.gate + (pix < a.gate_split ? 0 : a.Cout_pad)) : nullptr;
#pragma unroll
    for (int i = 0; i < CT; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cb = n0 + wn * (BN / WN) + i * 32 + 8 * g + 4 * hi;
        if (cb >= a.Cout_st) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][p][g * 4 + e];
        if (a.bias) {
          const f32x4 b = *(const f32x4*)(a.bias + cb);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += b[e];
        }
        if (a.act == 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_tanh_f(v[e]);
        }
        if (a.resid) {
          const uint2 rr = *(const uint2*)(a.resid + pix * a.ldr + cb);
          float r[4] = {__uint_as_float(rr.x << 16), __uint_as_float(rr.x & 0xffff0000u),
                        __uint_as_float(rr.y << 16), __uint_as_float(rr.y & 0xffff0000u)};
          if (gate) {
            const f32x4 gg = *(const f32x4*)(gate + cb);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = r[e] + gg[e] * v[e];
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += r[e];
          }
        }
        uint2 o;
        o.x = pack_bf2(v[0], v[1]);
        o.y = pack_bf2(v[2], v[3]);
        *(uint2*)(a.out + pix * a.ldo + cb) = o;
      }
    }
  }
}


template <int BN, int BK>
static int launch_v1(const IgemmArgs& a, unsigned grid, hipStream_t s) {
  constexpr int lds = 2 * (128 * BK * 2 + BN * BK * 2);
  static PerDeviceOnce attr_set;
  if (auto once_ = attr_set.guard()) (void)hipFuncSetAttribute((const void*)igemm_kernel<BN, BK>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL((igemm_kernel<BN, BK>), dim3(grid), dim3(256), lds, s, a);
  DOVE_CHECK_LAUNCH("dove_conv_igemm_bf16");
  return DOVE_OK;
}
int launch_igemm_legacy(const IgemmArgs& a, unsigned grid, int BN, bool bk64, hipStream_t s) {
  if (BN == 128) return bk64 ? launch_v1<128, 64>(a, grid, s) : launch_v1<128, 32>(a, grid, s);
  if (BN == 64) return bk64 ? launch_v1<64, 64>(a, grid, s) : launch_v1<64, 32>(a, grid, s);
  return bk64 ? launch_v1<32, 64>(a, grid, s) : launch_v1<32, 32>(a, grid, s);
}
