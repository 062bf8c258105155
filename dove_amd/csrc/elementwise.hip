// Small HBM-bound glue kernels of the DOVE hot path (gfx950): layout changes between the reference's
// [B,C,T,H,W] boundary tensors and the internal channels-last bf16 layout, Downsample3D's temporal
// average pool, the VAE posterior sample, get_velocity, DiT patchify/unpatchify and the M=1 linears
// (timestep embedding MLP and AdaLN modulation vectors, constant for a fixed sr_noise_step).
// Reference call sites: /root/reference/inference_script.py:407-409 (encode+sample), :483-493
// (transformer + get_velocity), :500-501 (decode + range map).
#include "common.h"
#include "../../include/dove_hip.h"

__device__ __forceinline__ float load_any(const void* p, long long i, int dt) {
  return dt == DOVE_F32 ? ((const float*)p)[i] : bf2f(((const bf16_t*)p)[i]);
}
__device__ __forceinline__ void store_any(void* p, long long i, int dt, float v) {
  if (dt == DOVE_F32) ((float*)p)[i] = v;
  else ((bf16_t*)p)[i] = f2bf(v);
}

// ---- [C,T,H,W] (fp32|bf16) -> [T,H,W,Cp] bf16, channels >= C zero-filled, y = x*scale + shift ----
__global__ void cl_from_ncthw_kernel(const void* __restrict__ x, int dt, int C, long long npix, int Cp,
                                     float scale, float shift, bf16_t* __restrict__ y) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npix) return;
  bf16_t* yr = y + p * Cp;
  for (int c0 = 0; c0 < Cp; c0 += 8) {
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = c0 + e;
      f[e] = c < C ? load_any(x, (long long)c * npix + p, dt) * scale + shift : 0.f;
    }
    *(uint4*)(yr + c0) = pack8(f);
  }
}

extern "C" int dove_cl_from_ncthw(const void* x, int dtype, int C, long long npix, int Cp, float scale, float shift,
                                   void* y, void* stream) {
  DOVE_CHECK_ARG(x && y, "cl_from_ncthw: null pointer");
  DOVE_CHECK_ARG(dtype == DOVE_F32 || dtype == DOVE_BF16, "cl_from_ncthw: bad dtype %d", dtype);
  DOVE_CHECK_ARG(Cp % 8 == 0 && Cp >= C && C > 0 && npix > 0, "cl_from_ncthw: need Cp %% 8 == 0, Cp >= C");
  hipLaunchKernelGGL(cl_from_ncthw_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                     dtype, C, npix, Cp, scale, shift, (bf16_t*)y);
  DOVE_CHECK_LAUNCH("dove_cl_from_ncthw");
  return DOVE_OK;
}

// ---- [C,T,H,W] -> [T,H,W,Cp] bf16 with the 3x3 spatial neighbourhood unrolled into channels: y[..][(dy*3+dx)*C + c] =
//      x[c][t][h+dy-1][w+dx-1] (zero outside the frame), channels >= 9C zero.  encoder.conv_in (3 -> 128, 3x3x3) then runs as a
//      (3,1,1) conv with K = 3 x 32 instead of 27 x 32 (29 of 32 input channels were padding) ----
__global__ void cl_im2col3x3_kernel(const void* __restrict__ x, int dt, int C, int T, int H, int W, int Cp, float scale, float shift,
                                    bf16_t* __restrict__ y) {
  const long long npix = (long long)T * H * W;
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npix) return;
  const int w = (int)(p % W), h = (int)((p / W) % H);
  bf16_t* yr = y + p * Cp;
  for (int c0 = 0; c0 < Cp; c0 += 8) {
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = c0 + e, tap = k / C, c = k - tap * C;
      const int dy = tap / 3 - 1, dx = tap % 3 - 1;
      const bool in = tap < 9 && h + dy >= 0 && h + dy < H && w + dx >= 0 && w + dx < W;
      f[e] = in ? load_any(x, (long long)c * npix + p + (long long)dy * W + dx, dt) * scale + shift : 0.f;
    }
    *(uint4*)(yr + c0) = pack8(f);
  }
}
extern "C" int dove_cl_im2col3x3_from_ncthw(const void* x, int dtype, int C, int T, int H, int W, int Cp, float scale, float shift, void* y,
                                            void* stream) {
  DOVE_CHECK_ARG(x && y, "cl_im2col3x3: null pointer");
  DOVE_CHECK_ARG(dtype == DOVE_F32 || dtype == DOVE_BF16, "cl_im2col3x3: bad dtype %d", dtype);
  DOVE_CHECK_ARG(C > 0 && Cp % 8 == 0 && Cp >= 9 * C && T > 0 && H > 0 && W > 0, "cl_im2col3x3: need Cp %% 8 == 0 and Cp >= 9 C");
  const long long npix = (long long)T * H * W;
  hipLaunchKernelGGL(cl_im2col3x3_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, dtype, C, T, H, W, Cp,
                     scale, shift, (bf16_t*)y);
  DOVE_CHECK_LAUNCH("dove_cl_im2col3x3_from_ncthw");
  return DOVE_OK;
}

// ---- [T,H,W,ld] bf16 -> [C,T,H,W] (fp32|bf16), y = clamp(x*scale + shift, lo, hi) ----
__global__ void ncthw_from_cl_kernel(const bf16_t* __restrict__ x, long long ld, int C, long long npix, float scale,
                                     float shift, float lo, float hi, void* __restrict__ y, int dt) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npix) return;
  const bf16_t* xr = x + p * ld;
  for (int c0 = 0; c0 < C; c0 += 4) {
    const uint2 v = *(const uint2*)(xr + c0);
    const float f[4] = {__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                        __uint_as_float(v.y & 0xffff0000u)};
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (c0 + e < C) store_any(y, (long long)(c0 + e) * npix + p, dt, fminf(fmaxf(f[e] * scale + shift, lo), hi));
  }
}

extern "C" int dove_ncthw_from_cl(const void* x, long long ld, int C, long long npix, float scale, float shift,
                                   float lo, float hi, void* y, int dtype, void* stream) {
  DOVE_CHECK_ARG(x && y, "ncthw_from_cl: null pointer");
  DOVE_CHECK_ARG(dtype == DOVE_F32 || dtype == DOVE_BF16, "ncthw_from_cl: bad dtype %d", dtype);
  DOVE_CHECK_ARG(ld % 4 == 0 && ld >= ((C + 3) / 4) * 4 && C > 0 && npix > 0, "ncthw_from_cl: ld must be a multiple of 4 covering C");
  hipLaunchKernelGGL(ncthw_from_cl_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, ld, C, npix, scale, shift, lo, hi, y, dtype);
  DOVE_CHECK_LAUNCH("dove_ncthw_from_cl");
  return DOVE_OK;
}

// ---- tap-split decoder.conv_out, second half: sum the 9 spatially shifted partial planes (see include/dove_hip.h) ----
// One workgroup = 8 x 64 output pixels of one frame: the 10 x 66 partial-plane records they touch are read ONCE, coalesced, into LDS
// (record stride ldp + 1 floats: conflict-free for the per-lane reads below), then every thread sums its 9 x C values out of LDS.
// (A direct gather - 27 scattered 4-byte loads per pixel - ran at 0.33 TB/s.)
__global__ __launch_bounds__(256) void conv_out_gather_kernel(const float* __restrict__ p, int ldp, int T, int H, int W, int C,
                                                              const float* __restrict__ bias, float scale, float shift, float lo, float hi,
                                                              void* __restrict__ y, int dt, int ldy) {
  extern __shared__ __attribute__((aligned(16))) float tile[];
  constexpr int TH = 8, TW = 64, RW = TW + 2, NREC = (TH + 2) * RW;
  const int lrec = ldp + 1, f4n = ldp >> 2;
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH, t = blockIdx.z;
  for (int i = threadIdx.x; i < NREC * f4n; i += 256) {
    const int rec = i / f4n, f4 = i - rec * f4n;
    const int ry = rec / RW, rx = rec - ry * RW;
    const int iy = y0 + ry - 1, ix = x0 + rx - 1;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};                                        // zero padding at the frame border
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *(const f32x4*)(p + (((long long)t * H + iy) * W + ix) * ldp + f4 * 4);
    float* d = tile + rec * lrec + f4 * 4;
    d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
  }
  __syncthreads();
  const int lx = threadIdx.x & 63;
  const long long npix = (long long)T * H * W;
#pragma unroll
  for (int h2 = 0; h2 < 2; ++h2) {
    const int rr = (threadIdx.x >> 6) + 4 * h2;
    const int oy = y0 + rr, ox = x0 + lx;
    if (oy >= H || ox >= W) continue;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const float* q = tile + ((rr + dy) * RW + lx + dx) * lrec + (dy * 3 + dx) * C;
        for (int c = 0; c < C; ++c) acc[c] += q[c];
      }
    const long long pix = ((long long)t * H + oy) * W + ox;
    if (ldy > 0) {                                                         // channels-last bf16 [T][H][W][ldy] (a spatial tile: blended before the layout change)
      bf16_t* yp = (bf16_t*)y + pix * ldy;
      for (int c = 0; c < ldy; ++c) yp[c] = c < C ? f2bf(acc[c] + (bias ? bias[c] : 0.f)) : (bf16_t)0;
      continue;
    }
    for (int c = 0; c < C; ++c) {
      const float v = bf2f(f2bf(acc[c] + (bias ? bias[c] : 0.f)));          // the conv's bf16 output rounding
      store_any(y, (long long)c * npix + pix, dt, fminf(fmaxf(v * scale + shift, lo), hi));
    }
  }
}
static int conv_out_gather_launch(const float* p, long long ldp, int T, int H, int W, int C, const float* bias, float scale, float shift,
                                  float lo, float hi, void* y, int dtype, int ldy, void* stream) {
  DOVE_CHECK_ARG(p && y, "conv_out_gather: null pointer");
  DOVE_CHECK_ARG(C >= 1 && C <= 4 && ldp >= 9 * C && ldp <= 36 && ldp % 4 == 0 && T > 0 && H > 0 && W > 0,
                 "conv_out_gather: need 1 <= C <= 4 and 9 C <= ldp <= 36, ldp %% 4 == 0");
  const int lds = 10 * 66 * ((int)ldp + 1) * 4;
  static PerDeviceOnce attr;
  if (auto once_ = attr.guard()) { (void)hipFuncSetAttribute((const void*)conv_out_gather_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 10 * 66 * 37 * 4); }
  dim3 grid((unsigned)((W + 63) / 64), (unsigned)((H + 7) / 8), (unsigned)T);
  hipLaunchKernelGGL(conv_out_gather_kernel, grid, dim3(256), lds, (hipStream_t)stream, p, (int)ldp, T, H, W, C, bias, scale, shift, lo, hi, y, dtype, ldy);
  DOVE_CHECK_LAUNCH("dove_conv_out_gather");
  return DOVE_OK;
}
extern "C" int dove_conv_out_gather(const float* p, long long ldp, int T, int H, int W, int C, const float* bias, float scale, float shift,
                                    float lo, float hi, void* y, int dtype, void* stream) {
  DOVE_CHECK_ARG(dtype == DOVE_F32 || dtype == DOVE_BF16, "conv_out_gather: bad dtype %d", dtype);
  return conv_out_gather_launch(p, ldp, T, H, W, C, bias, scale, shift, lo, hi, y, dtype, 0, stream);
}
extern "C" int dove_conv_out_gather_cl(const float* p, long long ldp, int T, int H, int W, int C, const float* bias, void* y, int ldy, void* stream) {
  DOVE_CHECK_ARG(ldy >= C && ldy <= 32 && ldy % 4 == 0, "conv_out_gather_cl: need C <= ldy <= 32, ldy %% 4 == 0 (got %d)", ldy);
  DOVE_CHECK_ARG(T <= 65535, "conv_out_gather_cl: more than 65535 frames (%d) in one call", T);
  return conv_out_gather_launch(p, ldp, T, H, W, C, bias, 1.0f, 0.0f, 0.0f, 0.0f, y, DOVE_BF16, ldy, stream);
}

// ---- Downsample3D temporal pool: odd T keeps frame 0 and averages pairs (1,2),(3,4)..; even T pairs (0,1).. ----
__host__ __device__ inline int odd_to(int T) { return (T & 1) ? 1 + (T - 1) / 2 : T / 2; }
__global__ void avgpool_time_kernel(const bf16_t* __restrict__ x, int T, long long frame8, bf16_t* __restrict__ y) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // 8-element chunk within a frame
  if (i >= frame8) return;
  const int To = odd_to(T);
  const int inst = blockIdx.y / To, to = blockIdx.y - inst * To;   // instance of nb (each [T][frame] -> [To][frame], back to back)
  const int odd = T & 1;
  const uint4* xs = (const uint4*)x + (long long)inst * T * frame8;
  uint4* ys = (uint4*)y + (long long)inst * To * frame8;
  if (odd && to == 0) {
    ys[i] = xs[i];
    return;
  }
  const int t0 = odd ? 1 + 2 * (to - 1) : 2 * to;
  float a[8], b[8];
  unpack8(xs[(long long)t0 * frame8 + i], a);
  unpack8(xs[(long long)(t0 + 1) * frame8 + i], b);
#pragma unroll
  for (int e = 0; e < 8; ++e) a[e] = 0.5f * (a[e] + b[e]);
  ys[(long long)to * frame8 + i] = pack8(a);
}

extern "C" int dove_avgpool_time_nb_bf16(const void* x, int nb, int T, long long frame_elems, void* y, void* stream) {
  DOVE_CHECK_ARG(x && y, "avgpool_time: null pointer");
  DOVE_CHECK_ARG(T >= 2 && frame_elems % 8 == 0 && frame_elems > 0, "avgpool_time: need T >= 2 and frame_elems %% 8 == 0");
  const int To = odd_to(T);
  DOVE_CHECK_ARG(nb >= 1 && (long long)nb * To <= 65535, "avgpool_time: bad instance count %d", nb);
  const long long f8 = frame_elems / 8;
  hipLaunchKernelGGL(avgpool_time_kernel, dim3((unsigned)((f8 + 255) / 256), nb * To), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, T, f8, (bf16_t*)y);
  DOVE_CHECK_LAUNCH("dove_avgpool_time_bf16");
  return DOVE_OK;
}
extern "C" int dove_avgpool_time_bf16(const void* x, int T, long long frame_elems, void* y, void* stream) {
  return dove_avgpool_time_nb_bf16(x, 1, T, frame_elems, y, stream);
}

// ---- DiagonalGaussianDistribution: split channels-last moments into [2L,T,h,w] params, and sample ----
__global__ void posterior_kernel(const bf16_t* __restrict__ mom, long long ld, int L, long long npix,
                                 const void* __restrict__ noise, int ndt, void* __restrict__ out, int odt) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npix) return;
  for (int c = 0; c < L; ++c) {
    const float mean = bf2f(mom[p * ld + c]);
    float lv = bf2f(mom[p * ld + L + c]);
    lv = fminf(fmaxf(lv, -30.f), 20.f);
    const float eps = load_any(noise, (long long)c * npix + p, ndt);
    store_any(out, (long long)c * npix + p, odt, mean + __expf(0.5f * lv) * eps);
  }
}

extern "C" int dove_posterior_sample(const void* moments, long long ld, int latent_channels, long long npix,
                                      const void* noise, int noise_dtype, void* out, int out_dtype, void* stream) {
  DOVE_CHECK_ARG(moments && noise && out, "posterior_sample: null pointer");
  DOVE_CHECK_ARG(ld >= 2 * latent_channels && npix > 0, "posterior_sample: bad ld");
  hipLaunchKernelGGL(posterior_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)moments, ld, latent_channels, npix, noise, noise_dtype, out, out_dtype);
  DOVE_CHECK_LAUNCH("dove_posterior_sample");
  return DOVE_OK;
}

// ---- out = a*x + b*y (CogVideoXDPMScheduler.get_velocity / add_noise) ----
__global__ void axpby_kernel(const void* x, const void* y, void* out, int dt, long long n, float a, float b) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  store_any(out, i, dt, a * load_any(x, i, dt) + b * load_any(y, i, dt));
}

extern "C" int dove_axpby(const void* x, const void* y, void* out, int dtype, long long n, float a, float b,
                           void* stream) {
  DOVE_CHECK_ARG(x && y && out && n > 0, "axpby: null pointer / empty");
  DOVE_CHECK_ARG(dtype == DOVE_F32 || dtype == DOVE_BF16, "axpby: bad dtype %d", dtype);
  hipLaunchKernelGGL(axpby_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, out,
                     dtype, n, a, b);
  DOVE_CHECK_LAUNCH("dove_axpby");
  return DOVE_OK;
}

// ---- CogVideoXPatchEmbed gather: [T,C,h,w] -> tokens [(T/pt)*(h/p)*(w/p)][C*pt*p*p], feature order (c,t,ph,pw) ----
__global__ void patchify_kernel(const void* __restrict__ x, int dt, int T, int C, int H, int W, int pt, int p,
                                bf16_t* __restrict__ tok, long long ld, int dir, void* __restrict__ y) {
  const int gh = H / p, gw = W / p;
  const int F = C * pt * p * p;
  const long long total = (long long)(T / pt) * gh * gw * F;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int f = (int)(i % F);
  const long long n = i / F;
  const int pw = f % p, ph = (f / p) % p, tt = (f / (p * p)) % pt, c = f / (p * p * pt);
  const int iw = (int)(n % gw), ih = (int)((n / gw) % gh), it = (int)(n / ((long long)gw * gh));
  const long long src = (((long long)(it * pt + tt) * C + c) * H + ih * p + ph) * W + iw * p + pw;
  if (dir == 0) tok[n * ld + f] = f2bf(load_any(x, src, dt));
  else store_any(y, src, dt, bf2f(tok[n * ld + f]));
}

extern "C" int dove_patchify(const void* x, int dtype, int T, int C, int H, int W, int pt, int p, void* tokens,
                              long long ld, void* stream) {
  DOVE_CHECK_ARG(x && tokens, "patchify: null pointer");
  DOVE_CHECK_ARG(T % pt == 0 && H % p == 0 && W % p == 0 && ld >= (long long)C * pt * p * p, "patchify: shape not divisible by patch");
  const long long total = (long long)(T / pt) * (H / p) * (W / p) * C * pt * p * p;
  hipLaunchKernelGGL(patchify_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                     dtype, T, C, H, W, pt, p, (bf16_t*)tokens, ld, 0, nullptr);
  DOVE_CHECK_LAUNCH("dove_patchify");
  return DOVE_OK;
}

extern "C" int dove_unpatchify(const void* tokens, long long ld, int T, int C, int H, int W, int pt, int p, void* y,
                                int dtype, void* stream) {
  DOVE_CHECK_ARG(y && tokens, "unpatchify: null pointer");
  DOVE_CHECK_ARG(T % pt == 0 && H % p == 0 && W % p == 0 && ld >= (long long)C * pt * p * p, "unpatchify: shape not divisible by patch");
  const long long total = (long long)(T / pt) * (H / p) * (W / p) * C * pt * p * p;
  hipLaunchKernelGGL(patchify_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     nullptr, dtype, T, C, H, W, pt, p, (bf16_t*)const_cast<void*>(tokens), ld, 1, y);
  DOVE_CHECK_LAUNCH("dove_unpatchify");
  return DOVE_OK;
}

// ---- M = 1 linear: y[j] = b[j] + sum_k W[j][k] * act(x[k]); one wave per output row ----
__global__ __launch_bounds__(256) void gemv_kernel(const bf16_t* __restrict__ W, const float* __restrict__ b,
                                                   const float* __restrict__ x, int in_f, int out_f, int act,
                                                   float* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= out_f) return;
  const bf16_t* wr = W + (long long)j * in_f;
  float acc = 0.f;
  for (int k0 = lane * 8; k0 < in_f; k0 += 512) {
    float w[8];
    unpack8(*(const uint4*)(wr + k0), w);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float xv = x[k0 + e];
      if (act == 1) xv = silu_f(xv);
      acc += w[e] * xv;
    }
  }
  acc = wave_sum(acc);
  if (lane == 0) y[j] = acc + (b ? b[j] : 0.f);
}

extern "C" int dove_gemv_bf16(const void* W, const float* bias, const float* x, int in_features, int out_features,
                               int act_in, float* y, void* stream) {
  DOVE_CHECK_ARG(W && x && y, "gemv: null pointer");
  DOVE_CHECK_ARG(in_features % 8 == 0 && in_features > 0 && out_features > 0, "gemv: in_features must be a multiple of 8");
  hipLaunchKernelGGL(gemv_kernel, dim3((out_features + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)W,
                     bias, x, in_features, out_features, act_in, y);
  DOVE_CHECK_LAUNCH("dove_gemv_bf16");
  return DOVE_OK;
}

// ---- diffusers VAE tiling: same-shaped spatial tiles of a channels-last clip -> ONE batch (tile-major), 16-byte chunks ----
//   out [nb][nt][th][tw][C] <- x [T][H][W][C] frames [t0, t0 + nt), tile n at (oy[n], ox[n]).
// im2col_cin > 0: x is the im2col'ed clip of dove_cl_im2col3x3_from_ncthw (channel (dy*3+dx)*cin + c = pixel (y+dy-1, x+dx-1)); a tile must
// see ZERO padding at ITS border (diffusers runs the whole network per tile), so the channels of taps that reach outside the tile are zeroed
// at the tile's border pixels - the tile then equals the im2col of the cropped tile, bit for bit.
struct TileOrigins { int n; short oy[64], ox[64]; };
__global__ void tile_gather_kernel(const bf16_t* __restrict__ x, int H, int W, int C8, int t0, int nt, int th, int tw, TileOrigins org,
                                   int cin, bf16_t* __restrict__ out) {
  const long long per = (long long)nt * th * tw * C8, total = per * org.n;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(i / per);
    long long r = i - (long long)n * per;
    const int c = (int)(r % C8); r /= C8;
    const int xx = (int)(r % tw); r /= tw;
    const int yy = (int)(r % th);
    const int t = (int)(r / th);
    uint4 v = ((const uint4*)x)[(((long long)(t0 + t) * H + org.oy[n] + yy) * W + org.ox[n] + xx) * C8 + c];
    if (cin > 0 && (yy == 0 || yy == th - 1 || xx == 0 || xx == tw - 1)) {
      uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int ch = c * 8 + e;
        if (ch >= 9 * cin) continue;
        const int tap = ch / cin, dy = tap / 3, dx = tap - dy * 3;
        const bool outside = (yy == 0 && dy == 0) || (yy == th - 1 && dy == 2) || (xx == 0 && dx == 0) || (xx == tw - 1 && dx == 2);
        if (outside) w[e >> 1] &= (e & 1) ? 0x0000ffffu : 0xffff0000u;
      }
      v = uint4{w[0], w[1], w[2], w[3]};
    }
    ((uint4*)out)[i] = v;
  }
}
extern "C" int dove_tile_gather_bf16(const void* x, int H, int W, int C, int t0, int nt, int th, int tw, int nb, const int* oy, const int* ox,
                                     int im2col_cin, void* out, void* stream) {
  DOVE_CHECK_ARG(x && out && oy && ox, "tile_gather: null pointer");
  DOVE_CHECK_ARG(C > 0 && C % 8 == 0 && H > 0 && W > 0 && H < 32768 && W < 32768, "tile_gather: need C %% 8 == 0 and H, W < 32768");
  DOVE_CHECK_ARG(nb >= 1 && nb <= 64, "tile_gather: 1 <= nb <= 64 tiles per call (got %d)", nb);
  DOVE_CHECK_ARG(nt > 0 && t0 >= 0 && th > 0 && tw > 0 && th <= H && tw <= W, "tile_gather: bad tile shape %d x %d x %d", nt, th, tw);
  DOVE_CHECK_ARG(im2col_cin >= 0 && 9 * im2col_cin <= C, "tile_gather: im2col_cin %d does not fit %d channels", im2col_cin, C);
  TileOrigins org; org.n = nb;
  for (int n = 0; n < nb; ++n) {
    DOVE_CHECK_ARG(oy[n] >= 0 && ox[n] >= 0 && oy[n] + th <= H && ox[n] + tw <= W, "tile_gather: tile %d at (%d, %d) leaves the %d x %d frame", n, oy[n], ox[n], H, W);
    org.oy[n] = (short)oy[n]; org.ox[n] = (short)ox[n];
  }
  const long long n16 = (long long)nb * nt * th * tw * (C / 8);
  const unsigned grid = (unsigned)(n16 + 255 < 256ll * 8192 ? (n16 + 255) / 256 : 8192);
  hipLaunchKernelGGL(tile_gather_kernel, dim3(grid ? grid : 1), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, H, W, C / 8, t0, nt, th, tw, org,
                     im2col_cin, (bf16_t*)out);
  DOVE_CHECK_LAUNCH("dove_tile_gather_bf16");
  return DOVE_OK;
}

// ---- diffusers VAE tiling: blend_v / blend_h on channels-last tiles, in place on b ----
//   axis 0 (rows):  b[t, y, w, :] = a[t, Ha - extent + y, w, :] * (1 - y/extent) + b[t, y, w, :] * (y/extent),  y < extent
//   axis 1 (cols):  b[t, h, x, :] = a[t, h, Wa - extent + x, :] * (1 - x/extent) + b[t, h, x, :] * (x/extent),  x < extent
__global__ void blend_edge_kernel(const bf16_t* __restrict__ a, bf16_t* __restrict__ b, int T, int Ha, int Wa, int Hb,
                                  int Wb, int ld, int extent, int axis) {
  const int c4 = ld / 4;                                     // 8-byte granules (decoder tiles have ld = 4)
  const int L = axis == 0 ? Wb : Hb;                         // length of the untouched spatial axis
  const long long total = (long long)T * extent * L * c4;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % c4);
  long long r = i / c4;
  const int l = (int)(r % L); r /= L;
  const int e = (int)(r % extent);
  const int t = (int)(r / extent);
  long long ia, ib;
  if (axis == 0) {
    ia = (((long long)t * Ha + (Ha - extent + e)) * Wa + l) * ld + c * 4;
    ib = (((long long)t * Hb + e) * Wb + l) * ld + c * 4;
  } else {
    ia = (((long long)t * Ha + l) * Wa + (Wa - extent + e)) * ld + c * 4;
    ib = (((long long)t * Hb + l) * Wb + e) * ld + c * 4;
  }
  const uint2 va = *(const uint2*)(a + ia), vb = *(const uint2*)(b + ib);
  const float fa[4] = {__uint_as_float(va.x << 16), __uint_as_float(va.x & 0xffff0000u), __uint_as_float(va.y << 16), __uint_as_float(va.y & 0xffff0000u)};
  const float fb[4] = {__uint_as_float(vb.x << 16), __uint_as_float(vb.x & 0xffff0000u), __uint_as_float(vb.y << 16), __uint_as_float(vb.y & 0xffff0000u)};
  const float wb = (float)e / (float)extent, wa = 1.0f - wb;
  uint2 o;
  o.x = pack_bf2(fa[0] * wa + fb[0] * wb, fa[1] * wa + fb[1] * wb);
  o.y = pack_bf2(fa[2] * wa + fb[2] * wb, fa[3] * wa + fb[3] * wb);
  *(uint2*)(b + ib) = o;
}

extern "C" int dove_blend_edge_bf16(const void* a, void* b, int T, int Ha, int Wa, int Hb, int Wb, int ld, int extent,
                                     int axis, void* stream) {
  DOVE_CHECK_ARG(a && b, "blend_edge: null pointer");
  DOVE_CHECK_ARG(ld % 4 == 0 && ld > 0, "blend_edge: ld (%d) must be a multiple of 4", ld);
  DOVE_CHECK_ARG(axis == 0 || axis == 1, "blend_edge: axis must be 0 (rows) or 1 (cols)");
  DOVE_CHECK_ARG(T > 0 && extent >= 0, "blend_edge: bad shape");
  if (axis == 0) DOVE_CHECK_ARG(Wa == Wb && extent <= Ha && extent <= Hb, "blend_edge: row blend needs equal widths, extent <= heights");
  else DOVE_CHECK_ARG(Ha == Hb && extent <= Wa && extent <= Wb, "blend_edge: col blend needs equal heights, extent <= widths");
  if (extent == 0) return DOVE_OK;
  const long long total = (long long)T * extent * (axis == 0 ? Wb : Hb) * (ld / 4);
  hipLaunchKernelGGL(blend_edge_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)a, (bf16_t*)b, T, Ha, Wa, Hb, Wb, ld, extent, axis);
  DOVE_CHECK_LAUNCH("dove_blend_edge_bf16");
  return DOVE_OK;
}

// ---- script-level pre-processing (ref :192-235, :670-679): pad F to 8N+1 (repeat last frame), pad H,W to x16 with zeros
//      (bottom/right), bilinear x`up` (align_corners=False, torch semantics), x/255*2-1, [F,H,W,3] u8 -> [3,F',H',W'] ----
__global__ void preprocess_kernel(const uint8_t* __restrict__ src, int F0, int H0, int W0, int Fp, int Hp, int Wp, int up,
                                  void* __restrict__ out, int odt) {
  const int Ho = Hp * up, Wo = Wp * up;
  const long long npix = (long long)Fp * Ho * Wo;
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npix) return;
  const int ox = (int)(p % Wo);
  const int oy = (int)((p / Wo) % Ho);
  const int f = (int)(p / ((long long)Wo * Ho));
  const int fs = f < F0 ? f : F0 - 1;
  const float inv = 1.0f / (float)up;
  float sy = ((float)oy + 0.5f) * inv - 0.5f, sx = ((float)ox + 0.5f) * inv - 0.5f;
  sy = sy < 0.f ? 0.f : sy;
  sx = sx < 0.f ? 0.f : sx;
  const int y0 = (int)sy, x0 = (int)sx;
  const int y1 = y0 + (y0 < Hp - 1 ? 1 : 0), x1 = x0 + (x0 < Wp - 1 ? 1 : 0);
  const float ly = sy - (float)y0, lx = sx - (float)x0;
  const uint8_t* fr = src + (long long)fs * H0 * W0 * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    auto px = [&](int y, int x) -> float { return (y < H0 && x < W0) ? (float)fr[((long long)y * W0 + x) * 3 + c] : 0.f; };
    const float top = px(y0, x0) * (1.f - lx) + px(y0, x1) * lx;
    const float bot = px(y1, x0) * (1.f - lx) + px(y1, x1) * lx;
    const float v = top * (1.f - ly) + bot * ly;
    store_any(out, (long long)c * npix + p, odt, v / 255.0f * 2.0f - 1.0f);
  }
}

extern "C" int dove_preprocess_u8(const void* frames, int F0, int H0, int W0, int pad_f, int pad_h, int pad_w, int upscale,
                                   void* out, int out_dtype, void* stream) {
  DOVE_CHECK_ARG(frames && out, "preprocess: null pointer");
  DOVE_CHECK_ARG(F0 > 0 && H0 > 0 && W0 > 0 && pad_f >= 0 && pad_h >= 0 && pad_w >= 0 && upscale >= 1, "preprocess: bad shape");
  DOVE_CHECK_ARG(out_dtype == DOVE_F32 || out_dtype == DOVE_BF16, "preprocess: bad dtype %d", out_dtype);
  const long long npix = (long long)(F0 + pad_f) * (H0 + pad_h) * upscale * (W0 + pad_w) * upscale;
  hipLaunchKernelGGL(preprocess_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const uint8_t*)frames, F0, H0, W0, F0 + pad_f, H0 + pad_h, W0 + pad_w, upscale, out, out_dtype);
  DOVE_CHECK_LAUNCH("dove_preprocess_u8");
  return DOVE_OK;
}

// ---- post-processing (ref :238-246 crop, :124/143/168 uint8): [3,F,H,W] in [0,1] -> [Fo,Ho,Wo,3] u8 = trunc(clamp(x*255)) ----
__global__ void postprocess_kernel(const void* __restrict__ vid, int dt, int F, int H, int W, int Fo, int Ho, int Wo,
                                   uint8_t* __restrict__ out) {
  const long long npix = (long long)Fo * Ho * Wo;
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npix) return;
  const int x = (int)(p % Wo);
  const int y = (int)((p / Wo) % Ho);
  const int f = (int)(p / ((long long)Wo * Ho));
  const long long plane = (long long)F * H * W;
  const long long si = ((long long)f * H + y) * W + x;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float v = fminf(fmaxf(load_any(vid, c * plane + si, dt) * 255.0f, 0.f), 255.f);
    out[p * 3 + c] = (uint8_t)v;
  }
}

extern "C" int dove_postprocess_u8(const void* video, int dtype, int F, int H, int W, int Fo, int Ho, int Wo, void* out,
                                    void* stream) {
  DOVE_CHECK_ARG(video && out, "postprocess: null pointer");
  DOVE_CHECK_ARG(dtype == DOVE_F32 || dtype == DOVE_BF16, "postprocess: bad dtype %d", dtype);
  DOVE_CHECK_ARG(Fo > 0 && Ho > 0 && Wo > 0 && Fo <= F && Ho <= H && Wo <= W, "postprocess: crop must fit inside the video");
  const long long npix = (long long)Fo * Ho * Wo;
  hipLaunchKernelGGL(postprocess_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)stream, video,
                     dtype, F, H, W, Fo, Ho, Wo, (uint8_t*)out);
  DOVE_CHECK_LAUNCH("dove_postprocess_u8");
  return DOVE_OK;
}
