// HBM-bound normalisation kernels of the DOVE hot path (gfx950): 16-byte vector loads, wave-shuffle
// and LDS tree reductions, fp32 statistics (fp64 final combine), bf16 storage.
//   * GroupNorm(32) statistics over one frame-batch [T,H,W,C]    (encoder GN, decoder SpatialNorm3D)
//   * GroupNorm apply (+ latent-conditioned scale/shift of SpatialNorm3D) + SiLU
//   * LayerNorm + AdaLN-Zero modulation, text / video rows with different (shift, scale)
//   * per-head QK LayerNorm + 3D RoPE + softmax pre-scale + head-major re-layout (Q, K, V^T)
// Reference call sites: /root/reference/inference_script.py:408,500 (VAE), :483-489 (DiT); the
// arithmetic itself lives in diffusers.
#include "common.h"
#include "../../include/dove_hip.h"

// ------------------------------------------------------------------------------------------------
// GroupNorm statistics: partial (sum, sumsq) per block and group, then an fp64 combine.
// Thread layout: cpp = C/8 chunk columns x (256/cpp) pixel lanes; deterministic (no atomics).
// Grid = (blocks per frame, frames): a block's pixels and their summation order depend on the FRAME SIZE only, so the partial
// rows of two pieces of a frame-batch (dove_amd.dist: a batch split over a rank pair) are exactly the rows of the whole batch and
// the fp64 combine of either set gives the same statistics.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_partial_kernel(const bf16_t* __restrict__ xall, long long npix, int C,
                                                         int cpp_log, float* __restrict__ partial_all) {
  const bf16_t* __restrict__ x = xall + (long long)blockIdx.y * npix * C;      // npix = pixels of ONE frame
  float* __restrict__ partial = partial_all + (long long)blockIdx.y * gridDim.x * 64;
  __shared__ float red[256 * 17];
  __shared__ float chan[2 * 2048];
  const int tid = threadIdx.x;
  const int cpp = 1 << cpp_log;
  const int q = tid & (cpp - 1), sub = tid >> cpp_log;
  const int nsub = 256 >> cpp_log;
  float s[8], ss[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s[e] = 0.f; ss[e] = 0.f; }
  for (long long p = (long long)blockIdx.x * nsub + sub; p < npix; p += (long long)gridDim.x * nsub) {
    const uint4 v = *(const uint4*)(x + p * C + q * 8);
    float f[8];
    unpack8(v, f);
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] += f[e]; ss[e] += f[e] * f[e]; }
  }
  // red[(q*nsub + sub)*17 + k], k = 0..15 : (sum e0..7, sumsq e0..7); 17-stride breaks bank conflicts
  float* r = red + (q * nsub + sub) * 17;
#pragma unroll
  for (int e = 0; e < 8; ++e) { r[e] = s[e]; r[8 + e] = ss[e]; }
  __syncthreads();
  // per-channel block sums: 2*C outputs
  for (int o = tid; o < 2 * C; o += 256) {
    const int ch = o >> 1, which = o & 1;
    const int qq = ch >> 3, e = ch & 7;
    float acc = 0.f;
    for (int k = 0; k < nsub; ++k) acc += red[(qq * nsub + k) * 17 + which * 8 + e];
    chan[o] = acc;
  }
  __syncthreads();
  const int cpg = C / 32;
  if (tid < 64) {
    const int g = tid >> 1, which = tid & 1;
    float acc = 0.f;
    for (int k = 0; k < cpg; ++k) acc += chan[(g * cpg + k) * 2 + which];
    partial[(long long)blockIdx.x * 64 + tid] = acc;
  }
}

template <typename PT>   // float: per-block partials; double: the first-level sums of gn_reduce_rows_kernel
__global__ __launch_bounds__(256) void gn_finalize_kernel(const PT* __restrict__ partial_all, int nblocks, double count,
                                                          float eps, float* __restrict__ stats_all) {
  // blockIdx.x = instance (the batched forms: nb independent tensors, each with its own rows and its own statistics)
  const PT* __restrict__ partial = partial_all + (long long)blockIdx.x * nblocks * 64;
  float* __restrict__ stats = stats_all + blockIdx.x * 64;
  const int tid = threadIdx.x, j = tid & 63, part = tid >> 6;  // 4 strided partial sums per (group, which)
  double acc = 0.0;
#pragma unroll 8
  for (int b = part; b < nblocks; b += 4) acc += (double)partial[(long long)b * 64 + j];
  __shared__ double sh[256];
  sh[tid] = acc;
  __syncthreads();
  if (tid < 64) sh[tid] = (sh[tid] + sh[tid + 64]) + (sh[tid + 128] + sh[tid + 192]);
  __syncthreads();
  if (tid < 32) {
    const double mean = sh[tid * 2] / count;
    double var = sh[tid * 2 + 1] / count - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[tid * 2] = (float)mean;
    stats[tid * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

// raw fp64 (sum, sumsq) per group from the per-block partials (distributed pieces add these before finalising)
template <typename PT>
__global__ __launch_bounds__(256) void gn_sums_kernel(const PT* __restrict__ partial, int nblocks, double* __restrict__ sums) {
  const int tid = threadIdx.x, j = tid & 63, part = tid >> 6;
  double acc = 0.0;
#pragma unroll 8
  for (int b = part; b < nblocks; b += 4) acc += (double)partial[(long long)b * 64 + j];
  __shared__ double sh[256];
  sh[tid] = acc;
  __syncthreads();
  if (tid < 64) sums[tid] = (sh[tid] + sh[tid + 64]) + (sh[tid + 128] + sh[tid + 192]);
}
__global__ void gn_finalize_sums_kernel(const double* __restrict__ sums, double count, float eps, float* __restrict__ stats) {
  const int g = threadIdx.x;
  if (count <= 0.0) count = sums[64];      // the 65-double message of a rank pair carries the element count behind the sums
  if (g < 32) {
    const double mean = sums[g * 2] / count;
    double var = sums[g * 2 + 1] / count - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[g * 2] = (float)mean;
    stats[g * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

// partial rows written by a conv epilogue (one per tile x wave) -> 256 rows for gn_finalize_kernel; fixed order.  The first-level
// sums stay fp64 (they used to be rounded to fp32 here): sums of fp32 tile partials are then exact in practice at every level, so
// a frame-batch split over a rank pair (dove_amd.dist, whose pieces have other row counts) finalises to the same (mean, rstd)
__global__ __launch_bounds__(256) void gn_reduce_rows_kernel(const float* __restrict__ partial_all, long long rows,
                                                             double* __restrict__ out_all) {
  const float* __restrict__ partial = partial_all + (long long)blockIdx.y * rows * 64;     // blockIdx.y = instance
  double* __restrict__ out = out_all + (long long)blockIdx.y * gridDim.x * 64;
  const int tid = threadIdx.x, j = tid & 63, part = tid >> 6;
  double acc = 0.0;
  for (long long r = (long long)blockIdx.x * 4 + part; r < rows; r += (long long)gridDim.x * 4) acc += (double)partial[r * 64 + j];
  __shared__ double sh[256];
  sh[tid] = acc;
  __syncthreads();
  if (tid < 64) out[blockIdx.x * 64 + tid] = (sh[tid] + sh[tid + 64]) + (sh[tid + 128] + sh[tid + 192]);
}

extern "C" int dove_groupnorm_finalize_partials_nb(const float* partial, long long rows, int nb, double count, float eps, void* ws,
                                                   size_t ws_bytes, float* stats, void* stream) {
  DOVE_CHECK_ARG(partial && ws && stats, "groupnorm_finalize_partials: null pointer");
  DOVE_CHECK_ARG(rows > 0 && count > 0 && nb >= 1 && nb <= 65535, "groupnorm_finalize_partials: empty input");
  hipStream_t s = (hipStream_t)stream;
  if (rows <= 1024) {
    hipLaunchKernelGGL(gn_finalize_kernel<float>, dim3(nb), dim3(256), 0, s, partial, (int)rows, count, eps, stats);
  } else {
    DOVE_CHECK_ARG(ws_bytes >= (size_t)nb * 256 * 64 * sizeof(double), "groupnorm_finalize_partials: scratch of %zu bytes, %d instances need %zu",
                   ws_bytes, nb, (size_t)nb * 256 * 64 * sizeof(double));
    hipLaunchKernelGGL(gn_reduce_rows_kernel, dim3(256, nb), dim3(256), 0, s, partial, rows, (double*)ws);
    DOVE_CHECK_LAUNCH("dove_groupnorm_finalize_partials(reduce)");
    hipLaunchKernelGGL(gn_finalize_kernel<double>, dim3(nb), dim3(256), 0, s, (const double*)ws, 256, count, eps, stats);
  }
  DOVE_CHECK_LAUNCH("dove_groupnorm_finalize_partials");
  return DOVE_OK;
}
extern "C" int dove_groupnorm_finalize_partials(const float* partial, long long rows, double count, float eps, void* ws,
                                                float* stats, void* stream) {
  return dove_groupnorm_finalize_partials_nb(partial, rows, 1, count, eps, ws, (size_t)256 * 64 * sizeof(double), stats, stream);
}

extern "C" int dove_groupnorm_sums_from_partials(const float* partial, long long rows, void* ws, double* sums, void* stream) {
  DOVE_CHECK_ARG(partial && ws && sums && rows > 0, "groupnorm_sums_from_partials: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (rows <= 1024) {
    hipLaunchKernelGGL(gn_sums_kernel<float>, dim3(1), dim3(256), 0, s, partial, (int)rows, sums);
  } else {
    hipLaunchKernelGGL(gn_reduce_rows_kernel, dim3(256), dim3(256), 0, s, partial, rows, (double*)ws);
    DOVE_CHECK_LAUNCH("dove_groupnorm_sums_from_partials(reduce)");
    hipLaunchKernelGGL(gn_sums_kernel<double>, dim3(1), dim3(256), 0, s, (const double*)ws, 256, sums);
  }
  DOVE_CHECK_LAUNCH("dove_groupnorm_sums_from_partials");
  return DOVE_OK;
}

extern "C" int dove_groupnorm_finalize_sums(const double* sums, double count, float eps, float* stats, void* stream) {
  DOVE_CHECK_ARG(sums && stats, "groupnorm_finalize_sums: bad arguments");
  hipLaunchKernelGGL(gn_finalize_sums_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, count, eps, stats);
  DOVE_CHECK_LAUNCH("dove_groupnorm_finalize_sums");
  return DOVE_OK;
}

// launches gn_partial_kernel over [frames][frame_pix][C]; returns the number of partial rows written (0 after an argument error)
static int gn_partial_launch(const void* x, long long npix, long long frame_pix, int C, void* partial_ws, int ws_blocks, hipStream_t s,
                             const char* who) {
  if (!(C >= 32 && C <= 2048 && (C & (C - 1)) == 0)) { dove_set_error("%s: C (%d) must be a power of two in [32,2048]", who, C); return 0; }
  if (!(npix > 0 && ws_blocks > 0)) { dove_set_error("%s: empty input", who); return 0; }
  if (frame_pix <= 0) frame_pix = npix;
  if (npix % frame_pix) { dove_set_error("%s: npix (%lld) is not a multiple of frame_pix (%lld)", who, npix, frame_pix); return 0; }
  const long long frames = npix / frame_pix;
  if (frames > 65535 || frames > ws_blocks) { dove_set_error("%s: %lld frames exceed the scratch rows (%d)", who, frames, ws_blocks); return 0; }
  int cpp_log = 0;
  while ((1 << cpp_log) < C / 8) ++cpp_log;
  const int nsub = 256 >> cpp_log;
  // blocks per frame: a function of the frame size alone (>= 32 pixels per thread, at most 256 blocks) - see gn_partial_kernel;
  // only a scratch buffer too small for frames x bpf rows lowers it (the result stays deterministic, but no longer split-invariant)
  long long bpf = (frame_pix + (long long)nsub * 32 - 1) / ((long long)nsub * 32);
  if (bpf > 256) bpf = 256;
  if (bpf * frames > ws_blocks) {
    // a lowered block count would change the summation order with the number of frames in the call: a frame-batch split over a rank
    // pair (dove_amd.dist) or a batch of tiles would silently stop reducing to the statistics of the single call - refuse instead
    dove_set_error("%s: %lld frames x %lld blocks per frame exceed the %d scratch rows", who, frames, bpf, ws_blocks);
    return 0;
  }
  hipLaunchKernelGGL(gn_partial_kernel, dim3((unsigned)bpf, (unsigned)frames), dim3(256), 0, s, (const bf16_t*)x, frame_pix, C, cpp_log,
                     (float*)partial_ws);
  return (int)(bpf * frames);
}

extern "C" int dove_groupnorm_stats_bf16(const void* x, long long npix, long long frame_pix, int C, float eps, void* partial_ws,
                                          int ws_blocks, float* stats, void* stream) {
  DOVE_CHECK_ARG(x && partial_ws && stats, "groupnorm_stats: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const int rows = gn_partial_launch(x, npix, frame_pix, C, partial_ws, ws_blocks, s, "groupnorm_stats");
  if (!rows) return DOVE_EINVAL;
  DOVE_CHECK_LAUNCH("dove_groupnorm_stats_bf16(partial)");
  hipLaunchKernelGGL(gn_finalize_kernel<float>, dim3(1), dim3(256), 0, s, (const float*)partial_ws, rows,
                     (double)npix * (double)(C / 32), eps, stats);
  DOVE_CHECK_LAUNCH("dove_groupnorm_stats_bf16(finalize)");
  return DOVE_OK;
}

/* nb instances [nb][npix][C] back to back (npix = pixels of ONE instance): stats [nb][32][2].  The partial rows are per (frame, fixed share
 * of the frame), i.e. exactly the rows nb separate calls would write, as long as the scratch holds nb * frames * blocks-per-frame rows. */
extern "C" int dove_groupnorm_stats_nb_bf16(const void* x, int nb, long long npix, long long frame_pix, int C, float eps, void* partial_ws,
                                             int ws_blocks, float* stats, void* stream) {
  DOVE_CHECK_ARG(x && partial_ws && stats && nb >= 1, "groupnorm_stats: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const int rows = gn_partial_launch(x, npix * nb, frame_pix > 0 ? frame_pix : npix, C, partial_ws, ws_blocks, s, "groupnorm_stats");
  if (!rows) return DOVE_EINVAL;
  DOVE_CHECK_LAUNCH("dove_groupnorm_stats_bf16(partial)");
  DOVE_CHECK_ARG(rows % nb == 0, "groupnorm_stats: %d partial rows do not split over %d instances", rows, nb);
  hipLaunchKernelGGL(gn_finalize_kernel<float>, dim3(nb), dim3(256), 0, s, (const float*)partial_ws, rows / nb,
                     (double)npix * (double)(C / 32), eps, stats);
  DOVE_CHECK_LAUNCH("dove_groupnorm_stats_bf16(finalize)");
  return DOVE_OK;
}

extern "C" int dove_groupnorm_sums_bf16(const void* x, long long npix, long long frame_pix, int C, void* partial_ws, int ws_blocks,
                                        double* sums, void* stream) {
  DOVE_CHECK_ARG(x && partial_ws && sums, "groupnorm_sums: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const int rows = gn_partial_launch(x, npix, frame_pix, C, partial_ws, ws_blocks, s, "groupnorm_sums");
  if (!rows) return DOVE_EINVAL;
  DOVE_CHECK_LAUNCH("dove_groupnorm_sums_bf16(partial)");
  hipLaunchKernelGGL(gn_sums_kernel<float>, dim3(1), dim3(256), 0, s, (const float*)partial_ws, rows, sums);
  DOVE_CHECK_LAUNCH("dove_groupnorm_sums_bf16(sums)");
  return DOVE_OK;
}

// ------------------------------------------------------------------------------------------------
// GroupNorm apply (+ SpatialNorm3D conditioning) + SiLU.
//   y = silu( ((x - mean_g) * rstd_g * gamma_c + beta_c) [ * Y[z(p)][c] + B[z(p)][c] ] )
// `yb` is the [Tz, hz, wz, 2C] table conv_y(zq) || conv_b(zq) computed on the LATENT grid by the igemm
// kernel; the nearest-neighbour resize of zq to f's (T,H,W) becomes a gather index (never materialised).
// ------------------------------------------------------------------------------------------------
struct GnApplyArgs {
  const bf16_t* x; bf16_t* y; const float* stats; const float* gamma; const float* beta;
  const bf16_t* yb;
  int T, H, W, C, cpp_log;       // T = frames of ONE instance; the grid covers nb * T frames
  int hz, wz, sshift;
  int tmap[32];                  // per-instance frame map into that instance's Tz latent frames
  int act;
  int Tz;                        // latent frames per instance (yb is [nb * Tz, hz, wz, 2C])
};

// grid = (rows of H, T): each block walks image rows, threads = (C/8 channel chunks) x (256/(C/8)) pixels
__global__ __launch_bounds__(256) void gn_apply_kernel(const GnApplyArgs a) {
  const int tid = threadIdx.x;
  const int cpp = 1 << a.cpp_log;
  const int q = tid & (cpp - 1), sub = tid >> a.cpp_log;
  const int nsub = 256 >> a.cpp_log;
  const int cpg = a.C / 32;
  const int t = blockIdx.y;                                   // global frame: instance b = t / T, frame t - b T of it
  const int b = t / a.T;
  const float* __restrict__ stats = a.stats + b * 64;         // every instance has its own GroupNorm scope
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ch = q * 8 + e, g = ch / cpg;
    const float mean = stats[g * 2], rstd = stats[g * 2 + 1];
    sc[e] = rstd * a.gamma[ch];
    sh[e] = a.beta[ch] - mean * sc[e];
  }
  const int tz = b * a.Tz + a.tmap[t - b * a.T];
  for (int h = blockIdx.x; h < a.H; h += gridDim.x) {
    const long long rowbase = ((long long)t * a.H + h) * a.W;
    const long long zrow = ((long long)tz * a.hz + (h >> a.sshift)) * a.wz;
    // four pixels per thread and trip: all loads of a trip are issued before the first SiLU (a lone 16-byte load per trip
    // left the kernel latency-bound at 5.2 TB/s)
    for (int w0 = sub; w0 < a.W; w0 += 4 * nsub) {
      uint4 xv[4], yv[4], bv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int w = w0 + u * nsub;
        if (w < a.W) {
          xv[u] = *(const uint4*)(a.x + (rowbase + w) * a.C + q * 8);
          if (a.yb) {
            const bf16_t* yb = a.yb + (zrow + (w >> a.sshift)) * (2 * a.C) + q * 8;
            yv[u] = *(const uint4*)yb;
            bv[u] = *(const uint4*)(yb + a.C);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int w = w0 + u * nsub;
        if (w < a.W) {
          float f[8];
          unpack8(xv[u], f);
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] = f[e] * sc[e] + sh[e];
          if (a.yb) {
            float fy[8], fb[8];
            unpack8(yv[u], fy);
            unpack8(bv[u], fb);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = f[e] * fy[e] + fb[e];
          }
          if (a.act) {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = silu_f(f[e]);
          }
          *(uint4*)(a.y + (rowbase + w) * a.C + q * 8) = pack8(f);
        }
      }
    }
  }
}

extern "C" int dove_groupnorm_apply_nb_bf16(const void* x, void* y, int nb, int T, int H, int W, int C, const float* stats,
                                             const float* gamma, const float* beta, int silu, const void* yb, int Tz, int hz,
                                             int wz, int sshift, const int* tmap, void* stream) {
  DOVE_CHECK_ARG(x && y && stats && gamma && beta, "groupnorm_apply: null pointer");
  DOVE_CHECK_ARG(nb >= 1 && (long long)nb * T <= 65535, "groupnorm_apply: bad instance count %d", nb);
  DOVE_CHECK_ARG(C >= 32 && C <= 2048 && (C & (C - 1)) == 0, "groupnorm_apply: C (%d) must be a power of two in [32,2048]", C);
  DOVE_CHECK_ARG(T > 0 && T <= 32 && H > 0 && W > 0, "groupnorm_apply: need 0 < T <= 32 frames per batch (got %d), H, W > 0", T);
  GnApplyArgs a;
  a.x = (const bf16_t*)x; a.y = (bf16_t*)y; a.stats = stats; a.gamma = gamma; a.beta = beta; a.yb = (const bf16_t*)yb;
  a.T = T; a.H = H; a.W = W; a.C = C; a.act = silu;
  int cpp_log = 0;
  while ((1 << cpp_log) < C / 8) ++cpp_log;
  a.cpp_log = cpp_log;
  a.hz = hz; a.wz = wz; a.sshift = sshift; a.Tz = Tz;
  for (int i = 0; i < 32; ++i) a.tmap[i] = 0;
  if (yb) {
    DOVE_CHECK_ARG(tmap, "groupnorm_apply: spatial norm needs a frame map");
    DOVE_CHECK_ARG(sshift >= 0 && sshift <= 8 && (hz << sshift) >= H && (wz << sshift) >= W, "groupnorm_apply: latent grid too small");
    for (int i = 0; i < T; ++i) a.tmap[i] = tmap[i];
  }
  hipLaunchKernelGGL(gn_apply_kernel, dim3(H < 4096 ? H : 4096, nb * T), dim3(256), 0, (hipStream_t)stream, a);
  DOVE_CHECK_LAUNCH("dove_groupnorm_apply_bf16");
  return DOVE_OK;
}
extern "C" int dove_groupnorm_apply_bf16(const void* x, void* y, int T, int H, int W, int C, const float* stats,
                                          const float* gamma, const float* beta, int silu, const void* yb, int hz,
                                          int wz, int sshift, const int* tmap, void* stream) {
  return dove_groupnorm_apply_nb_bf16(x, y, 1, T, H, W, C, stats, gamma, beta, silu, yb, 0, hz, wz, sshift, tmap, stream);
}

// ------------------------------------------------------------------------------------------------
// LayerNorm over the last dim D (affine) followed by AdaLN modulation y = LN(x)*(1+scale)+shift,
// with (shift, scale) chosen by row class: rows < split use mod[0] (text), else mod[1] (video).
// mod layout: [2 classes][2 (shift, scale)][D] fp32; mod == nullptr -> plain LayerNorm.
// One wave per row, row kept in registers (D <= 4096), two-pass mean/variance in fp32.
// ------------------------------------------------------------------------------------------------
template <int NIT>
__global__ __launch_bounds__(256) void ln_mod_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                     long long rows, int D, float eps,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const float* __restrict__ mod, long long split) {
  // combined affine per row class in LDS: y = (x-mean)*rstd * A[cls][c] + B[cls][c],
  //   A = gamma*(1+scale), B = beta*(1+scale)+shift  (32 rows per block amortise the parameter reads)
  extern __shared__ __attribute__((aligned(16))) float lnp[];
  float* A = lnp;            // [2][D]
  float* Bv = lnp + 2 * D;   // [2][D]
  for (int c = threadIdx.x; c < D; c += 256) {
    const float g = gamma[c], b = beta[c];
#pragma unroll
    for (int cls = 0; cls < 2; ++cls) {
      float sc = 0.f, sh = 0.f;
      if (mod) { sh = mod[(long long)cls * 2 * D + c]; sc = mod[(long long)cls * 2 * D + D + c]; }
      A[cls * D + c] = g * (1.0f + sc);
      Bv[cls * D + c] = b * (1.0f + sc) + sh;
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = 0; r < 8; ++r) {
    const long long row = (long long)blockIdx.x * 32 + wave * 8 + r;
    if (row >= rows) break;
    const bf16_t* xr = x + row * D;
    float f[NIT][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int c0 = (i * 64 + lane) * 8;
      if (c0 < D) {
        unpack8(*(const uint4*)(xr + c0), f[i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += f[i][e];
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) f[i][e] = 0.f;
      }
    }
    const float mean = wave_sum(s) / (float)D;
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int c0 = (i * 64 + lane) * 8;
      if (c0 < D) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = f[i][e] - mean; v += d * d; }
      }
    }
    const float rstd = rsqrtf(wave_sum(v) / (float)D + eps);
    const int cls = row < split ? 0 : 1;
    bf16_t* yr = y + row * D;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int c0 = (i * 64 + lane) * 8;
      if (c0 < D) {
        const f32x4 a0 = *(const f32x4*)(A + cls * D + c0), a1 = *(const f32x4*)(A + cls * D + c0 + 4);
        const f32x4 b0 = *(const f32x4*)(Bv + cls * D + c0), b1 = *(const f32x4*)(Bv + cls * D + c0 + 4);
        float o[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = (f[i][e] - mean) * rstd * a0[e] + b0[e];
          o[4 + e] = (f[i][4 + e] - mean) * rstd * a1[e] + b1[e];
        }
        *(uint4*)(yr + c0) = pack8(o);
      }
    }
  }
}

extern "C" int dove_layernorm_modulate_bf16(const void* x, void* y, long long rows, int D, float eps,
                                             const float* gamma, const float* beta, const float* mod,
                                             long long split, void* stream) {
  DOVE_CHECK_ARG(x && y && gamma && beta, "layernorm_modulate: null pointer");
  DOVE_CHECK_ARG(D % 8 == 0 && D > 0 && D <= 4096, "layernorm_modulate: D (%d) must be a multiple of 8, <= 4096", D);
  DOVE_CHECK_ARG(rows > 0, "layernorm_modulate: empty input");
  const int nit = (D + 511) / 512;
  const unsigned grid = (unsigned)((rows + 31) / 32);
  const size_t lds = (size_t)4 * D * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
#define LN_LAUNCH(N)                                                                                          \
  hipLaunchKernelGGL((ln_mod_kernel<N>), dim3(grid), dim3(256), lds, s, (const bf16_t*)x, (bf16_t*)y, rows, D, eps, \
                     gamma, beta, mod, split)
  switch (nit) {
    case 1: LN_LAUNCH(1); break;
    case 2: LN_LAUNCH(2); break;
    case 3: LN_LAUNCH(3); break;
    case 4: LN_LAUNCH(4); break;
    case 5: LN_LAUNCH(5); break;
    case 6: LN_LAUNCH(6); break;
    case 7: LN_LAUNCH(7); break;
    default: LN_LAUNCH(8); break;
  }
#undef LN_LAUNCH
  DOVE_CHECK_LAUNCH("dove_layernorm_modulate_bf16");
  return DOVE_OK;
}

// ------------------------------------------------------------------------------------------------
// qkv_post: split the fused QKV projection [N, 3*D] into head-major attention operands.
//   Q' [H][Npad][64] = RoPE(LN64(q)) * (softmax_scale * log2 e)      (rows >= text_len get RoPE)
//   K' [H][Npad][64] = RoPE(LN64(k))
//   V^T[H][64][Npad] = v transposed (so the PV MFMA reads its A operand like K)
// Eight lanes own one (token, head) 64-vector, 16 bytes each: every load / store instruction of a wave covers whole 128-B lines of
// 8 tokens (one lane per vector touched 64 different lines per instruction and ran at 2.6 TB/s: round 3).  LayerNorm statistics are
// reduced over the 8 lanes by three xor-shuffles; the interleaved-pair rotation stays inside a lane's 8 values.
// ------------------------------------------------------------------------------------------------
constexpr int QK_HB = 8;                                       // heads per thread: the token's cos / sin rows (fp32: 4x the bytes of the data they
                                                               // rotate) are fetched once per QK_HB heads, not once per head
__global__ __launch_bounds__(256) void qk_post_kernel(const bf16_t* __restrict__ qkv, long long N, long long Npad, int heads, int text_len,
                                                      const float* __restrict__ gq, const float* __restrict__ bq,
                                                      const float* __restrict__ gk, const float* __restrict__ bk,
                                                      const float* __restrict__ cosT, const float* __restrict__ sinT, float qscale, float eps,
                                                      bf16_t* __restrict__ Qh, bf16_t* __restrict__ Kh, float* __restrict__ norm2) {
  __shared__ unsigned smax[4][QK_HB];
  const int c = threadIdx.x & 7;                               // 8-value chunk of the 64-vector
  const int h0 = blockIdx.y * QK_HB, which = blockIdx.z;       // 0 q, 1 k
  const long long n = (long long)blockIdx.x * 32 + (threadIdx.x >> 3);
  const int D = heads * 64;
  const bool live = n < N;                                     // all 8 lanes of a token agree: the shuffles below stay inside live groups
  const long long nn = live ? n : N - 1;                       // dead lanes compute on a valid row and store nothing
  const float* gam = (which == 0 ? gq : gk) + c * 8;
  const float* bet = (which == 0 ? bq : bk) + c * 8;
  const f32x4 g0 = *(const f32x4*)gam, g1 = *(const f32x4*)(gam + 4), b0 = *(const f32x4*)bet, b1 = *(const f32x4*)(bet + 4);
  const bool rope = nn >= text_len && cosT;
  f32x4 c0 = {1.f, 1.f, 1.f, 1.f}, c1 = c0, s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
  if (rope) {
    const float* cr = cosT + (nn - text_len) * 64 + c * 8;
    const float* sr = sinT + (nn - text_len) * 64 + c * 8;
    c0 = *(const f32x4*)cr; c1 = *(const f32x4*)(cr + 4); s0 = *(const f32x4*)sr; s1 = *(const f32x4*)(sr + 4);
  }
  const float sc = which == 0 ? qscale : 1.0f;
  const bf16_t* src = qkv + nn * (3LL * D) + (long long)which * D + c * 8;
  bf16_t* dst = (which == 0 ? Qh : Kh) + nn * 64 + c * 8;
  uint4 raw[QK_HB];
#pragma unroll
  for (int j = 0; j < QK_HB; ++j) raw[j] = h0 + j < heads ? *(const uint4*)(src + (h0 + j) * 64) : uint4{0u, 0u, 0u, 0u};
#pragma unroll
  for (int j = 0; j < QK_HB; ++j) {
    float f[8];
    unpack8(raw[j], f);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s += f[e];
    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
    const float mean = s * (1.0f / 64.0f);
    float v = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float t = f[e] - mean; v += t * t; }
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
    const float rstd = rsqrtf(v * (1.0f / 64.0f) + eps);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = (f[e] - mean) * rstd * (e < 4 ? g0[e] : g1[e - 4]) + (e < 4 ? b0[e] : b1[e - 4]);
    if (rope) {
      // diffusers rounds LN output to the model dtype before apply_rotary_emb's fp32 math; keep fp32 here
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float a = f[2 * i], b = f[2 * i + 1];
        const float ca = i < 2 ? c0[2 * i] : c1[2 * i - 4], cb = i < 2 ? c0[2 * i + 1] : c1[2 * i - 3];
        const float sa = i < 2 ? s0[2 * i] : s1[2 * i - 4], sb = i < 2 ? s0[2 * i + 1] : s1[2 * i - 3];
        f[2 * i] = a * ca - b * sa;
        f[2 * i + 1] = b * cb + a * sb;
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] *= sc;
    const uint4 packed = pack8(f);
    if (live && h0 + j < heads) *(uint4*)(dst + (long long)(h0 + j) * Npad * 64) = packed;
    if (norm2) {
      // squared norm of the STORED (bf16-rounded, scaled, rotated) row: what bounds the attention scores of this head (dove_attention_fwd_bf16)
      float r[8];
      unpack8(packed, r);
      float q = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) q += r[e] * r[e];
      q += __shfl_xor(q, 1); q += __shfl_xor(q, 2); q += __shfl_xor(q, 4);          // the row (8 lanes)
      // the wave's 8 rows: the maximum of the BIT PATTERNS (a sum of squares is non-negative: same order) so that a NaN row - larger than
      // +inf as an unsigned - survives every stage up to the attention kernel, whose `b <= 40` test then picks the running maximum
      unsigned qb = __float_as_uint(q);
      qb = max(qb, (unsigned)__shfl_xor((int)qb, 8)); qb = max(qb, (unsigned)__shfl_xor((int)qb, 16)); qb = max(qb, (unsigned)__shfl_xor((int)qb, 32));
      if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6][j] = qb;
    }
  }
  if (norm2) {                                                 // one atomic per workgroup, head and operand (dead rows repeat row N - 1)
    __syncthreads();
    const int j = threadIdx.x;
    if (j < QK_HB && h0 + j < heads) {
      atomicMax((unsigned*)(norm2 + (h0 + j) * 2 + which), max(max(smax[0][j], smax[1][j]), max(smax[2][j], smax[3][j])));
    }
  }
}

// V^T: a wave's 64 tokens x 64 dims go through LDS so that lane d owns row d of V^T (128 contiguous bytes per lane instead of 64
// two-byte stores Npad apart); the loads are the same 8-lanes-per-token full-line pattern as above.
__global__ __launch_bounds__(256) void v_post_kernel(const bf16_t* __restrict__ qkv, long long N, long long Npad, int heads,
                                                     bf16_t* __restrict__ Vt, int v_swap) {
  __shared__ float vt[4][64][65];   // one slice per wave (row stride 65: conflict-free both ways)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = blockIdx.y;
  const long long n0 = ((long long)blockIdx.x * 4 + wave) * 64;
  if (n0 >= N) return;
  const int D = heads * 64;
  const int t8 = lane >> 3, c = lane & 7;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const long long n = n0 + i * 8 + t8;
    float f[8];
    if (n < N) unpack8(*(const uint4*)(qkv + n * (3LL * D) + 2LL * D + h * 64 + c * 8), f);
    else {
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = 0.f;                           // keys past N inside the pad stay zero
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) vt[wave][i * 8 + t8][c * 8 + e] = f[e];
  }
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if ((Npad & 7) == 0) {
    bf16_t* dst = Vt + ((long long)h * 64 + lane) * Npad + n0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float o[8];
      // natural: slots 8j .. 8j+7 hold keys 8j .. 8j+7; quad-swapped (what dove_attention_fwd_bf16 reads): every 16 keys are stored
      // [0-3, 8-11, 4-7, 12-15] so that a 16-byte V^T fragment holds the keys the QK^T MFMA left in the same lane half
      const int g16 = (j >> 1) * 16, odd = (j & 1) * 4;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = v_swap ? vt[wave][g16 + odd + (e & 3) + (e >> 2) * 8][lane] : vt[wave][j * 8 + e][lane];
      if (n0 + j * 8 + 8 <= Npad) *(uint4*)(dst + j * 8) = pack8(o);
    }
  } else {                                                              // row stride not 16-byte aligned (rank-local packing): natural order,
    bf16_t* dst = Vt + ((long long)h * 64 + lane) * Npad + n0;          // two-byte stores of the tokens that exist
    for (int j = 0; j < 64 && n0 + j < N; ++j) dst[j] = f2bf(vt[wave][j][lane]);
  }
}

extern "C" int dove_qkv_post_bf16(const void* qkv, long long N, long long Npad, int heads, int head_dim, int text_len,
                                   const float* gq, const float* bq, const float* gk, const float* bk,
                                   const float* cosT, const float* sinT, float qscale, float eps, void* Qh, void* Kh,
                                   void* Vt, int v_order, float* norm2, void* stream) {
  DOVE_CHECK_ARG(qkv && Qh && Kh && Vt && gq && bq && gk && bk, "qkv_post: null pointer");
  DOVE_CHECK_ARG(head_dim == 64, "qkv_post: head_dim must be 64 (got %d)", head_dim);
  // Npad is only the row stride of the head-major outputs here (the attention kernel is what wants a multiple of 128);
  // dove_amd.dist packs rank-local rows with Npad == N so a head group is one contiguous all-to-all chunk
  DOVE_CHECK_ARG(N > 0 && Npad >= N, "qkv_post: Npad must be >= N");
  DOVE_CHECK_ARG((cosT == nullptr) == (sinT == nullptr), "qkv_post: cos/sin must both be given or both be null");
  DOVE_CHECK_ARG(v_order == 0 || (v_order == 1 && Npad % 16 == 0), "qkv_post: v_order 1 (quad-swapped V^T) needs Npad %% 16 == 0");
  if (norm2) {
    const hipError_t me = hipMemsetAsync(norm2, 0, sizeof(float) * 2 * heads, (hipStream_t)stream);
    if (me != hipSuccess) { dove_set_error("qkv_post: clearing norm2 failed: %s", hipGetErrorString(me)); return DOVE_ELAUNCH; }
  }
  hipLaunchKernelGGL(qk_post_kernel, dim3((unsigned)((N + 31) / 32), (unsigned)((heads + QK_HB - 1) / QK_HB), 2), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, N, Npad, heads,
                     text_len, gq, bq, gk, bk, cosT, sinT, qscale, eps, (bf16_t*)Qh, (bf16_t*)Kh, norm2);
  hipLaunchKernelGGL(v_post_kernel, dim3((unsigned)((N + 255) / 256), heads), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, N, Npad, heads,
                     (bf16_t*)Vt, v_order);
  DOVE_CHECK_LAUNCH("dove_qkv_post_bf16");
  return DOVE_OK;
}

// natural <-> quad-swapped key order of V^T rows, in place (an involution): every 32-byte group [q0 q1 q2 q3] -> [q0 q2 q1 q3]
__global__ __launch_bounds__(256) void vt_quad_swap_kernel(uint4* __restrict__ vt, long long ngroups) {
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  if (g >= ngroups) return;
  const uint4 a = vt[2 * g], b = vt[2 * g + 1];
  vt[2 * g] = uint4{a.x, a.y, b.x, b.y};
  vt[2 * g + 1] = uint4{a.z, a.w, b.z, b.w};
}
extern "C" int dove_vt_quad_swap_bf16(void* Vt, long long rows, long long Npad, void* stream) {
  DOVE_CHECK_ARG(Vt && rows > 0 && Npad > 0 && Npad % 16 == 0, "vt_quad_swap: Npad must be a positive multiple of 16");
  const long long ng = rows * (Npad / 16);
  hipLaunchKernelGGL(vt_quad_swap_kernel, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (uint4*)Vt, ng);
  DOVE_CHECK_LAUNCH("dove_vt_quad_swap_bf16");
  return DOVE_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Sequence/head-parallel (Ulysses) DiT, receive side of the Q' / K' / V^T all-to-all (dove_amd.dist.dit_forward_ulysses): the three
// receive buffers hold, per source rank i, that rank's rows of MY heads - [hloc][c_i][64] for Q', K' and [hloc][64][c_i] (natural key
// order) for V^T - and the attention kernel wants [hloc][Npad][64] / [hloc][64][Npad] with the V^T keys quad-swapped and the pad
// columns zero.  One launch instead of 3 x world slice copies + a pad clear + an in-place swap per layer.
// ------------------------------------------------------------------------------------------------------------------
struct UlyPlaceArgs {
  const bf16_t *rq, *rk, *rv;
  bf16_t *Qh, *Kh, *Vt;
  int world, hloc;
  long long N, Npad;
  long long bound[17];        // rows [bound[i], bound[i+1]) come from rank i
  long long off[17];          // element offset of rank i's block in a receive buffer
  int extra;                  // 1: every source block carries ONE extra row per head (q, k) / column (v) behind its counts[i] rows - the
  float* norm2_out;           //    K block's holds that rank's (max |q|^2, max |k|^2) of the head as two floats -> norm2_out [hloc][2] = max over ranks
};
__device__ __forceinline__ int uly_rank(const UlyPlaceArgs& a, long long n) {
  int i = 0;
  while (i + 1 < a.world && n >= a.bound[i + 1]) ++i;
  return i;
}
__global__ __launch_bounds__(256) void ulysses_place_kernel(const UlyPlaceArgs a) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.y == 3) {                                        // the score bound that rode in the K blocks' extra rows
    if (t >= a.hloc * 2) return;
    const int h = (int)(t >> 1), e = (int)(t & 1);
    // the same maximum of bit patterns as qk_post_kernel's atomicMax on one GPU: a NaN norm from ANY rank (larger than +inf as an unsigned)
    // reaches the attention kernel, which then runs that head on the running maximum - fmaxf would have dropped it
    unsigned m = 0u;
    for (int i = 0; i < a.world; ++i) {
      const long long ci = a.bound[i + 1] - a.bound[i];
      m = max(m, ((const unsigned*)(a.rk + a.off[i] + ((long long)h * (ci + 1) + ci) * 64))[e]);
    }
    ((unsigned*)a.norm2_out)[t] = m;
    return;
  }
  if (blockIdx.y < 2) {                                         // Q', K': one 16-byte chunk (8 of the 64 d) of one row
    const long long total = (long long)a.hloc * a.N * 8;
    if (t >= total) return;
    const int c = (int)(t & 7);
    const long long n = (t >> 3) % a.N;
    const int h = (int)((t >> 3) / a.N);
    const int i = uly_rank(a, n);
    const long long ci = a.bound[i + 1] - a.bound[i] + a.extra;
    const bf16_t* src = (blockIdx.y == 0 ? a.rq : a.rk) + a.off[i] + ((long long)h * ci + (n - a.bound[i])) * 64 + c * 8;
    bf16_t* dst = (blockIdx.y == 0 ? a.Qh : a.Kh) + ((long long)h * a.Npad + n) * 64 + c * 8;
    *(uint4*)dst = *(const uint4*)src;
  } else {                                                      // V^T: one quad (4 keys) of one (head, d) row, written at its swapped place
    const long long quads = a.Npad >> 2, total = (long long)a.hloc * 64 * quads;
    if (t >= total) return;
    const long long qd = t % quads;
    const long long row = t / quads;                            // h * 64 + d
    const int h = (int)(row >> 6), d = (int)(row & 63);
    bf16_t v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const long long n = qd * 4 + e;
      v[e] = 0;
      if (n < a.N) {
        const int i = uly_rank(a, n);
        const long long ci = a.bound[i + 1] - a.bound[i] + a.extra;
        v[e] = a.rv[a.off[i] + ((long long)h * 64 + d) * ci + (n - a.bound[i])];
      }
    }
    const int q = (int)(qd & 3);
    const long long pq = (qd & ~3ll) | (q == 1 ? 2 : (q == 2 ? 1 : q));   // [q0 q1 q2 q3] -> [q0 q2 q1 q3] inside every 16 keys
    uint2 w;
    w.x = (uint32_t)v[0] | ((uint32_t)v[1] << 16);
    w.y = (uint32_t)v[2] | ((uint32_t)v[3] << 16);
    *(uint2*)(a.Vt + row * a.Npad + pq * 4) = w;
  }
}
extern "C" int dove_ulysses_place_bf16(const void* rq, const void* rk, const void* rv, const long long* counts, int world, int hloc, long long N,
                                       long long Npad, void* Qh, void* Kh, void* Vt, float* norm2_out, void* stream) {
  DOVE_CHECK_ARG(rq && rk && rv && counts && Qh && Kh && Vt, "ulysses_place: null pointer");
  DOVE_CHECK_ARG(world >= 1 && world <= 16 && hloc >= 1 && N > 0 && Npad >= N && Npad % 16 == 0, "ulysses_place: bad shape (world %d, hloc %d)", world, hloc);
  UlyPlaceArgs a;
  a.rq = (const bf16_t*)rq; a.rk = (const bf16_t*)rk; a.rv = (const bf16_t*)rv;
  a.Qh = (bf16_t*)Qh; a.Kh = (bf16_t*)Kh; a.Vt = (bf16_t*)Vt;
  a.world = world; a.hloc = hloc; a.N = N; a.Npad = Npad;
  long long b = 0, o = 0;
  for (int i = 0; i < 17; ++i) { a.bound[i] = N; a.off[i] = 0; }
  for (int i = 0; i < world; ++i) {
    DOVE_CHECK_ARG(counts[i] >= 0, "ulysses_place: negative row count");
    a.bound[i] = b; a.off[i] = o;
    b += counts[i]; o += (counts[i] + (norm2_out ? 1 : 0)) * hloc * 64;
  }
  a.extra = norm2_out ? 1 : 0;
  a.norm2_out = norm2_out;
  a.bound[world] = b;
  DOVE_CHECK_ARG(b == N, "ulysses_place: the ranks' row counts add up to %lld, not N = %lld", b, N);
  const long long tq = (long long)hloc * N * 8, tv = (long long)hloc * 64 * (Npad >> 2);
  const long long tmax = tq > tv ? tq : tv;
  dim3 grid((unsigned)((tmax + 255) / 256), norm2_out ? 4 : 3);
  hipLaunchKernelGGL(ulysses_place_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
  DOVE_CHECK_LAUNCH("dove_ulysses_place_bf16");
  return DOVE_OK;
}
