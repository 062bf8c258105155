// Shared pieces of the implicit-GEMM translation units (igemm.hip: the product kernels and the dispatch; igemm_legacy.hip: the generic
// v1 kernel that only test-sized upsample-fused convs reach; gemm4x_timing.hip: round 2's GEMM, TIMING build only).
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "../../include/dove_hip.h"

// tools/*_timing.py hand a device buffer to the NEXT conv call (per-phase s_memtime logs / ablation operands).  It used to be a field
// of dove_conv_desc; the product struct no longer carries it - the hook exists in libdove_hip_timing.so only.
#ifdef DOVE_TIMING_BUILD
extern void* g_timing_debug_buf;                     // defined in igemm.hip (with dove_timing_set_debug_buf)
#define DOVE_DBG_BUF g_timing_debug_buf
#else
#define DOVE_DBG_BUF ((void*)nullptr)
#endif

// Work-skipping ablation switches and s_memtime phase logs exist ONLY in a -DDOVE_TIMING_BUILD library (built by the
// tools/*_timing.py helpers into a separate file); in the product build DOVE_DBG() is the constant 0, the branches fold
// away, no timing instantiation is emitted and no environment variable can make a kernel skip work.
#ifdef DOVE_TIMING_BUILD
#define DOVE_DBG(a) ((a).debug)
#else
#define DOVE_DBG(a) 0
#endif

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
  int debug;  // -DDOVE_TIMING_BUILD only (tools/, never the product library): 1 skip A loads, 2 skip B loads, 4 skip MFMA
  float* gn_partial;   // conv3x3_halo4x only: fused GroupNorm(32) partial sums of the stored output, [rows][32][2]
  int cpg_log;         // log2(channels per group) = log2(Cout / 32)
  int out_f32;         // igemm_fast only: `out` is float [..][ldo] (no bf16 rounding): the tap-split conv_out's partial sums
  int nt_out;          // gemm8p only: nontemporal output stores (outputs larger than the Infinity Cache: see conv_dispatch)
  // dove_conv_desc.nb independent instances back to back along the frame axis (the tile-batched VAE): T_out / T_in above are the TOTALS
  // (nb x per-instance), seg_out / seg_in the per-instance frame counts; temporal taps, the conv cache and tmode are per instance
  int seg_out, seg_in;
  long long cache_bs;  // elements between two instances' cache frames
  // conv3x3_halo4x, kt == 3, no conv cache (the first frame-batch of a clip / tile / chunk): the causal taps before an instance's first
  // frame all read the REPLICATED frame 0, so output frame 0 is (W0 + W1 + W2) x0 and output frame 1 is (W0 + W1) x0 + W2 x1 - one and two
  // temporal groups instead of three.  w_first = [2][9][Cout_pad][Cin]: the temporal sums W0 + W1 and W0 + W1 + W2, formed in fp32 and
  // rounded to bf16 ONCE at pack time (dove_conv_desc.w_first); nullptr = three groups for every frame
  const bf16_t* w_first;
  // conv3x3_halo4x, kt == 3, dove_conv_desc.tdup: the instance's frames come in bit-identical pairs (Upsample3D's time doubling), so two of
  // every frame's three causal taps read the same bits - two temporal groups per frame: (W0 + W1) x[t-1] + W2 x[t] or W0 x[t-2] + (W1 + W2) x[t].
  // w_pair = [2][9][Cout_pad][Cin]: W0 + W1 and W1 + W2, fp32 sums rounded once at pack time.  See h4_split
  const bf16_t* w_pair;
  int tdup;
  // conv3x3_halo4x: the rectangle this launch walks - origin (ty0, tx0) in 16 x 32 tiles of the conv's tiles_h x tiles_w grid, nty x ntx tiles
  // of the LAUNCH's geometry (a conv is one launch over the whole grid, or the grid minus its last tile column plus a 32 x 16-tile
  // launch for that column: see conv3x3_halo4x_kernel's kPart)
  int ty0, tx0, nty, ntx;
  int h_lim, w_lim;    // rows / columns of the output this launch owns end here (the image's H_out / W_out, or where a partial-tile launch takes over)
  int sub;             // conv3x3_halo4x<kSub>: sub-pixel form of the upsample-fused conv - `w` = dove_conv_desc.w_sub, tiling over the LOW-RES grid
};

// frame of the INPUT a temporal tap reads: output frame t (global index over all instances), tap dt of kt (causal: taps before an
// instance's first frame come from its conv cache, or replicate its frame 0), or the tmode map of the upsample convs (kt == 1)
__device__ __forceinline__ const bf16_t* igemm_src_frame(const IgemmArgs& a, int t, int dt, long long frame_elems) {
  const int b = a.seg_out == a.T_out ? 0 : t / a.seg_out;
  const int tl = t - b * a.seg_out;
  const long long f0 = (long long)b * a.seg_in;
  if (a.kt > 1) {
    const int fv = tl + dt - (a.kt - 1);
    if (fv >= 0) return a.x + (f0 + fv) * frame_elems;
    if (a.cache) return a.cache + (long long)b * a.cache_bs + (long long)(a.kt - 1 + fv) * frame_elems;
    return a.x + f0 * frame_elems;
  }
  const int tin = a.tmode == 0 ? tl : (a.tmode == 1 ? (tl >> 1) : (tl == 0 ? 0 : 1 + ((tl - 1) >> 1)));
  return a.x + (f0 + tin) * frame_elems;
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;

namespace gemm4x {
constexpr int BM = 256, BN = 256, BK = 32, ROWB = 64;
constexpr int A_ST = BM * ROWB, ST = A_ST + BN * ROWB;      // 16384 + 16384 per stage
constexpr int NST = 4;
constexpr int EPI = NST * ST;                                // epilogue staging: 4 waves x 32 rows x 256 B (XOR-swizzled)
constexpr int LDS_BYTES = EPI + 4 * 8192;                    // 163840 = all of a CU's LDS
}  // namespace gemm4x
struct G4Tile { int m0, n0; };
struct G4Const { int ntiles, G, tiles_n, nk4, K; long long M; };
struct G4State {                                             // the operand chunk (4 K-steps) being staged next
  int n_tile, n_k4;
  bool n_on;
  const bf16_t *a_base, *w_base;
  int a_nrec, w_nrec, soff;
};
__device__ __forceinline__ G4Tile g4_decode(const G4Const& k, int id) {
  const unsigned rest = xcd_remap((unsigned)(id < k.ntiles ? id : k.ntiles - 1), (unsigned)k.ntiles);
  // rasterisation: groups of GM row-tiles, row-tile fastest inside a group - the 32 CUs of an XCD (32 consecutive logical
  // tiles) then cover 8 x 4 tiles, i.e. 12 distinct operand panels per K step instead of 33, and every weight panel is
  // re-read from HBM / MALL once per 8 row-tiles instead of once per row-tile
  // (narrow outputs - N = 3072: 12 column tiles - keep the plain column-fastest order: their whole weight matrix is
  //  re-used by 2-3 row-tiles of the same XCD batch anyway and the plain order measured 3 % faster there)
  const unsigned GM = k.tiles_n > 16 ? 8u : 1u;
  const unsigned tiles_m = (unsigned)(k.ntiles / k.tiles_n);
  const unsigned per_group = GM * (unsigned)k.tiles_n;
  const unsigned group = rest / per_group, within = rest - group * per_group;
  const unsigned left = tiles_m - group * GM;
  const unsigned gm = left < GM ? left : GM;
  G4Tile q;
  q.m0 = __builtin_amdgcn_readfirstlane((int)(group * GM + within % gm) * gemm4x::BM);
  q.n0 = __builtin_amdgcn_readfirstlane((int)(within / gm) * gemm4x::BN);
  return q;
}
__device__ __forceinline__ void g4_open_tile(G4State& s, const IgemmArgs& a, const G4Const& k, int id) {
  s.n_tile = id;
  s.n_on = id < k.ntiles;
  const G4Tile q = g4_decode(k, id);
  const long long left = k.M - q.m0;
  const int rows = left < gemm4x::BM ? (int)left : gemm4x::BM;
  s.a_base = a.x + (long long)q.m0 * k.K;
  s.w_base = a.w + (long long)q.n0 * k.K;
  s.a_nrec = s.n_on ? rows * k.K * 2 : 0;                     // rows past M: offset >= num_records -> zeros in LDS
  s.w_nrec = s.n_on ? gemm4x::BN * k.K * 2 : 0;
  s.n_k4 = 0;
  s.soff = 0;
}
__device__ __forceinline__ void g4_advance(G4State& s, const IgemmArgs& a, const G4Const& k) {
  if (++s.n_k4 == k.nk4) g4_open_tile(s, a, k, s.n_tile + k.G);
  s.soff = s.n_k4 * (4 * gemm4x::ROWB);
}

// generic v1 kernel (igemm_legacy.hip): BN in {32, 64, 128}, BK = 64 if Cin % 64 == 0 else 32
int launch_igemm_legacy(const IgemmArgs& a, unsigned grid, int BN, bool bk64, hipStream_t s);
#ifdef DOVE_TIMING_BUILD
// round 2's one-wave-per-SIMD GEMM (gemm4x_timing.hip), for the within-run A/Bs of tools/archive/gemm8p_ab.py / gemm4x_sched.py / gemm4x_timing.py:
// variant 0 plain, 1 GELU, 2 gate, 3 s_memtime log (a.zero = debug buffer), 4 round-2 DMA order
int launch_gemm4x_timing(const IgemmArgs& a, long long M, unsigned grid, int variant, hipStream_t s);
#endif
