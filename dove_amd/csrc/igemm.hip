// Implicit-GEMM convolution / linear layer on bf16 MFMA for gfx950.
//
// One kernel serves every dense contraction of the DOVE hot path:
//   * CogVideoXCausalConv3d 3x3x3 (front-only temporal halo from the previous frame-batch's
//     `conv_cache` or the replicated first frame)         -- vae.encode / decode_latents
//     (/root/reference/inference_script.py:408,500)
//   * Downsample3D's Conv2d 3x3 stride 2 with (0,1,0,1) zero pad, Upsample3D's Conv2d 3x3 with the
//     nearest x2 (space, optionally time) upsample folded into the input addressing
//   * 1x1x1 convs (resnet shortcut, SpatialNorm conv_y/conv_b on the latent)
//   * every Linear of the DiT (token-major [N, C] tensors are a 1 x 1 x N "image")
//     (/root/reference/inference_script.py:483-489)
//
// Layout: activations channels-last [T, H, W, C] bf16; weights pre-packed [tap][Cout_pad][Cin] bf16
// (K contiguous for both MFMA operands).  Tile: 128 output pixels x BN output channels per 256-thread
// workgroup (4 waves), K-step BK input channels of one tap.  Both operand tiles are staged with
// 16-byte global_load_lds (per-lane gather addresses do the im2col; out-of-image pixels read a zero
// page), XOR-swizzled on the source side so ds_read_b128 fragment reads are bank-conflict free, and
// double-buffered with one barrier per K-step.  MFMA operands are swapped (A = weights, B = pixels) so
// each lane ends up with 4 consecutive output channels of one pixel -> 8-byte epilogue stores.
// Epilogue: +bias, GELU(tanh), residual add, row-class-dependent gate (AdaLN-Zero), bf16 store.
#include "common.h"
#include "../../include/dove_hip.h"

struct IgemmArgs {
  const bf16_t* x;
  const bf16_t* cache;
  const bf16_t* w;
  const float* bias;
  const bf16_t* resid;
  const float* gate;
  bf16_t* out;
  const bf16_t* zero;
  int T_out, H_out, W_out;
  int T_in, H_in, W_in;
  int Cin, Cout_pad, Cout_st;
  int kt, kh, kw, stride, pad_h, pad_w, up, tmode, act;
  long long ldo, ldr;
  long long gate_split;
  int tw_log2, tiles_w, tiles_h, tiles_n;
};

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
    const bf16_t* fp;
    if (a.kt > 1) {
      const int fv = t + dt - (a.kt - 1);
      if (fv >= 0) fp = a.x + fv * frame_elems;
      else if (a.cache) fp = a.cache + (a.kt - 1 + fv) * frame_elems;
      else fp = a.x;
    } else {
      const int tin = a.tmode == 0 ? t : (a.tmode == 1 ? (t >> 1) : (t == 0 ? 0 : 1 + ((t - 1) >> 1)));
      fp = a.x + tin * frame_elems;
    }
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

// ---- host side -------------------------------------------------------------------------------
static bf16_t* g_zero_page[16] = {nullptr};

static const bf16_t* zero_page() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  if (!g_zero_page[dev]) {
    void* p = nullptr;
    if (hipMalloc(&p, 4096) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, 4096) != hipSuccess) return nullptr;
    g_zero_page[dev] = (bf16_t*)p;
  }
  return g_zero_page[dev];
}

template <int BN, int BK>
static int launch_igemm(const IgemmArgs& a, unsigned grid, hipStream_t s) {
  constexpr int lds = 2 * (128 * BK * 2 + BN * BK * 2);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)igemm_kernel<BN, BK>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((igemm_kernel<BN, BK>), dim3(grid), dim3(256), lds, s, a);
  DOVE_CHECK_LAUNCH("dove_conv_igemm_bf16");
  return DOVE_OK;
}

extern "C" int dove_conv_igemm_bf16(const dove_conv_desc* d, void* stream) {
  DOVE_CHECK_ARG(d && d->x && d->w && d->out, "conv_igemm: null pointer");
  DOVE_CHECK_ARG(d->cin % 32 == 0 && d->cin > 0, "conv_igemm: Cin (%d) must be a positive multiple of 32 (pad on pack)", d->cin);
  DOVE_CHECK_ARG(d->cout_pad % 32 == 0 && d->cout_pad > 0, "conv_igemm: Cout_pad (%d) must be a multiple of 32", d->cout_pad);
  DOVE_CHECK_ARG(d->cout_store % 4 == 0 && d->cout_store <= d->cout_pad && d->cout_store > 0, "conv_igemm: bad cout_store %d", d->cout_store);
  DOVE_CHECK_ARG(d->ldo % 4 == 0 && d->ldo >= d->cout_store, "conv_igemm: ldo (%lld) must be a multiple of 4 and >= cout_store", d->ldo);
  DOVE_CHECK_ARG(!d->resid || (d->ldr % 4 == 0 && d->ldr >= d->cout_store), "conv_igemm: bad ldr %lld", d->ldr);
  DOVE_CHECK_ARG(!d->gate || d->resid, "conv_igemm: gate needs resid");
  DOVE_CHECK_ARG(d->kt >= 1 && d->kt <= 3 && d->kh >= 1 && d->kw >= 1, "conv_igemm: bad kernel size");
  DOVE_CHECK_ARG(d->stride == 1 || d->stride == 2, "conv_igemm: stride must be 1 or 2");
  DOVE_CHECK_ARG(d->up == 0 || d->up == 1, "conv_igemm: up must be 0/1");
  DOVE_CHECK_ARG(d->kt == 1 || (d->tmode == 0 && d->t_in == d->t_out), "conv_igemm: causal temporal taps need t_in == t_out and tmode 0");
  DOVE_CHECK_ARG(d->t_out > 0 && d->h_out > 0 && d->w_out > 0 && d->t_in > 0 && d->h_in > 0 && d->w_in > 0, "conv_igemm: empty tensor");
  const bf16_t* zp = zero_page();
  DOVE_CHECK_ARG(zp, "conv_igemm: could not allocate the zero page");
  IgemmArgs a;
  a.x = (const bf16_t*)d->x; a.cache = (const bf16_t*)d->cache; a.w = (const bf16_t*)d->w;
  a.bias = d->bias; a.resid = (const bf16_t*)d->resid; a.gate = d->gate; a.out = (bf16_t*)d->out; a.zero = zp;
  a.T_out = d->t_out; a.H_out = d->h_out; a.W_out = d->w_out;
  a.T_in = d->t_in; a.H_in = d->h_in; a.W_in = d->w_in;
  a.Cin = d->cin; a.Cout_pad = d->cout_pad; a.Cout_st = d->cout_store;
  a.kt = d->kt; a.kh = d->kh; a.kw = d->kw; a.stride = d->stride; a.pad_h = d->pad_h; a.pad_w = d->pad_w;
  a.up = d->up; a.tmode = d->tmode; a.act = d->act; a.ldo = d->ldo; a.ldr = d->ldr; a.gate_split = d->gate_split;
  // tile shape: 8x16 pixels for images, 1x128 for token-major (H == 1) tensors
  int twl = 7;
  if (d->h_out > 1) {
    twl = 4;
    if (d->w_out <= 8) twl = 3;
    if (d->w_out <= 4) twl = 2;
  }
  a.tw_log2 = twl;
  const int TW = 1 << twl, TH = 128 >> twl;
  a.tiles_w = (d->w_out + TW - 1) / TW;
  a.tiles_h = (d->h_out + TH - 1) / TH;
  const int BN = (d->cout_pad % 128 == 0) ? 128 : ((d->cout_pad % 64 == 0) ? 64 : 32);
  a.tiles_n = d->cout_pad / BN;
  const long long grid = (long long)a.T_out * a.tiles_h * a.tiles_w * a.tiles_n;
  DOVE_CHECK_ARG(grid > 0 && grid < (1ll << 31), "conv_igemm: grid too large");
  hipStream_t s = (hipStream_t)stream;
  const bool bk64 = (d->cin % 64 == 0);
  if (BN == 128) return bk64 ? launch_igemm<128, 64>(a, (unsigned)grid, s) : launch_igemm<128, 32>(a, (unsigned)grid, s);
  if (BN == 64) return bk64 ? launch_igemm<64, 64>(a, (unsigned)grid, s) : launch_igemm<64, 32>(a, (unsigned)grid, s);
  return bk64 ? launch_igemm<32, 64>(a, (unsigned)grid, s) : launch_igemm<32, 32>(a, (unsigned)grid, s);
}
