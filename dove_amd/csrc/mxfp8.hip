// MXFP8 path for the DiT linears (BASELINE configs[4]: "fp8 MFMA attention/FFN path (CDNA4 fp8), PSNR-gated vs bf16").
//
// Replaces the bf16 nn.Linear arithmetic of CogVideoXBlock (attn1.to_q/k/v fused, attn1.to_out.0, ff.net.0.proj,
// ff.net.2; /root/reference/inference_script.py:483-489) by OCP microscaling FP8: elements e4m3fn, one shared E8M0
// scale (a power of two) per 32 consecutive K elements of a row, for the activation AND the weight operand.  On gfx950
// only the block-scaled v_mfma_scale_f32_32x32x64_f8f6f4 runs above the bf16 MFMA rate (plain fp8 MFMA = bf16 rate), and
// it applies both scales inside the matrix pipe - no dequantisation pass exists anywhere.
//
//   dove_mx_quant_bf16 : bf16 [R][K] -> e4m3 [R][K] + scales; per block s = 2^ceil(log2(amax / 448)) (nothing clips),
//                        q = rne_e4m3(x / s).  Used for the weights once at load and for activations per call.
//   dove_linear_mxfp8  : out[M][N] (bf16) = epilogue( (xq * xs) (wq * ws)^T ), fp32 accumulation, the same fused
//                        epilogue as the bf16 GEMM (bias, GELU(tanh), residual, AdaLN-Zero gate per row class).
//
// Scale layout (both operands): u32 [K/256][R][2]; word (c, r, h) holds, in byte u = 0..3, the E8M0 scale of row r's
// 32-element block  8c + 2u + h.  That is exactly what one lane of the MFMA needs for four consecutive K-steps of 64:
// lane (row = l & 31, h = l >> 5) supplies the scale of elements [64 s + 32 h, 64 s + 32 h + 32) of its row at step
// s = 4c + u and selects byte u with op_sel - one coalesced dword load per fragment row per 256 K.
//
// GEMM structure = gemm4x of igemm.hip (256 x 256 tile, ONE wave per SIMD with the 512-register file, 4-stage LDS ring
// filled by 16-byte LDS-DMA three steps ahead with counted vmcnt, persistent workgroups, XCD-aware rasterisation,
// LDS-transposed full-line epilogue) with 64-BYTE K rows holding 64 fp8 elements: one 32x32x64 MFMA (131 kFLOP) replaces
// two 32x32x16 bf16 MFMAs per (i, p) pair, so a K-step carries the same 8 LDS-DMA + 16 fragment reads but twice the
// arithmetic - the LDS-DMA issue bound of the bf16 kernel halves relative to the MFMA time.
// Fragments of step s+1 are read into the other register set while the 16 MFMAs of step s run.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "../../include/dove_hip.h"

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef int v8i __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------------------------------------------------------
// quantiser: one thread = 8 consecutive elements (16 B in, 8 B out), 4 threads = one 32-element block
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned e8m0_for_amax(float amax) {
  // smallest power of two s with amax / s <= 448 (e4m3fn max): s = 2^ceil(log2(amax / 448)); amax = 0 -> 2^-127
  if (!(amax > 0.f)) return 0u;
  const float r = amax * (1.0f / 448.0f);
  const unsigned u = __float_as_uint(r);
  int e = (int)((u >> 23) & 0xff);                       // biased exponent of r (floor(log2 r) + 127)
  if (u & 0x7fffffu) e += 1;                             // not an exact power of two: round the exponent up
  if (e < 0) e = 0;
  if (e > 254) e = 254;
  return (unsigned)e;
}

__global__ __launch_bounds__(256) void mx_quant_kernel(const bf16_t* __restrict__ x, long long R, int K, unsigned char* __restrict__ q,
                                                       unsigned char* __restrict__ s) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const int per_row = K >> 3;
  const long long row = t / per_row;
  if (row >= R) return;                                   // whole quads leave together (per_row % 4 == 0)
  const int c8 = (int)(t - row * per_row);                // 8-element group inside the row
  const uint4 v = *(const uint4*)(x + row * K + (long long)c8 * 8);
  float f[8];
  unpack8(v, f);
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(f[i]));
  amax = fmaxf(amax, __shfl_xor(amax, 1));
  amax = fmaxf(amax, __shfl_xor(amax, 2));
  const unsigned e = e8m0_for_amax(amax);
  const float inv = __uint_as_float((254u - e) << 23);    // 2^-(e-127); e in [0,254] -> exponent field in [0,254]
  unsigned lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0] * inv, f[1] * inv, lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2] * inv, f[3] * inv, lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4] * inv, f[5] * inv, hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6] * inv, f[7] * inv, hi, true);
  *(uint2*)(q + row * K + (long long)c8 * 8) = make_uint2(lo, hi);
  if ((c8 & 3) == 0) {
    const int blk = c8 >> 2;                              // 32-element block of the row
    const int c = blk >> 3, u = (blk >> 1) & 3, h = blk & 1;
    s[(((long long)c * R + row) * 2 + h) * 4 + u] = (unsigned char)e;
  }
}

extern "C" int dove_mx_quant_bf16(const void* x, long long rows, int K, void* q, void* scales, void* stream) {
  DOVE_CHECK_ARG(x && q && scales, "mx_quant: null pointer");
  DOVE_CHECK_ARG(rows > 0 && K > 0 && K % 256 == 0, "mx_quant: K (%d) must be a positive multiple of 256", K);
  const long long nthreads = rows * (K >> 3);
  DOVE_CHECK_ARG((nthreads + 255) / 256 < (1ll << 31), "mx_quant: tensor too large");
  hipLaunchKernelGGL(mx_quant_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                     rows, K, (unsigned char*)q, (unsigned char*)scales);
  DOVE_CHECK_LAUNCH("dove_mx_quant_bf16");
  return DOVE_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// GEMM
// ------------------------------------------------------------------------------------------------------------------
namespace mxg {
constexpr int BM = 256, BN = 256, ROWB = 64;                 // 64 fp8 elements = 64 bytes of every row per K-step
constexpr int A_ST = BM * ROWB, ST = A_ST + BN * ROWB;       // 16384 + 16384 per stage
constexpr int NST = 4;
constexpr int EPI = NST * ST;                                // epilogue staging: 4 waves x 32 rows x 256 B
constexpr int LDS_BYTES = EPI + 4 * 8192;                    // 163840
}  // namespace mxg

struct MxArgs {
  const unsigned char *x, *w;                                // e4m3 [M][K], [N][K]
  const unsigned *xs, *ws;                                   // scales [K/256][M][2], [K/256][N][2]
  const float* bias;
  const bf16_t* resid;
  const float* gate;
  bf16_t* out;
  long long M, ldo, ldr, gate_split;
  int N, K, tiles_n;
};
struct MxTile { int m0, n0; };
struct MxConst { int ntiles, G, tiles_n, nk4, K; long long M; };
struct MxState {
  int n_tile, n_k4;
  bool n_on;
  const unsigned char *a_base, *w_base;
  int a_nrec, w_nrec, soff, m0, n0;
};
__device__ __forceinline__ MxTile mx_decode(const MxConst& k, int id) {
  const unsigned rest = xcd_remap((unsigned)(id < k.ntiles ? id : k.ntiles - 1), (unsigned)k.ntiles);
  const unsigned GM = k.tiles_n > 16 ? 8u : 1u;              // same rasterisation as gemm4x (igemm.hip g4_decode)
  const unsigned tiles_m = (unsigned)(k.ntiles / k.tiles_n);
  const unsigned per_group = GM * (unsigned)k.tiles_n;
  const unsigned group = rest / per_group, within = rest - group * per_group;
  const unsigned left = tiles_m - group * GM;
  const unsigned gm = left < GM ? left : GM;
  MxTile q;
  q.m0 = __builtin_amdgcn_readfirstlane((int)(group * GM + within % gm) * mxg::BM);
  q.n0 = __builtin_amdgcn_readfirstlane((int)(within / gm) * mxg::BN);
  return q;
}
__device__ __forceinline__ void mx_open_tile(MxState& s, const MxArgs& a, const MxConst& k, int id) {
  s.n_tile = id;
  s.n_on = id < k.ntiles;
  const MxTile q = mx_decode(k, id);
  const long long left = k.M - q.m0;
  const int rows = left < mxg::BM ? (int)left : mxg::BM;
  s.a_base = a.x + (long long)q.m0 * k.K;
  s.w_base = a.w + (long long)q.n0 * k.K;
  s.a_nrec = s.n_on ? rows * k.K : 0;                        // rows past M: offset >= num_records -> zeros in LDS
  s.w_nrec = s.n_on ? mxg::BN * k.K : 0;
  s.n_k4 = 0;
  s.soff = 0;
  s.m0 = q.m0; s.n0 = q.n0;
}
__device__ __forceinline__ void mx_advance(MxState& s, const MxArgs& a, const MxConst& k) {
  if (++s.n_k4 == k.nk4) mx_open_tile(s, a, k, s.n_tile + k.G);
  s.soff = s.n_k4 * (4 * mxg::ROWB);
}

template <bool kAct, bool kGate>
__global__ __launch_bounds__(256, 1) void gemm_mxfp8_kernel(const MxArgs a) {
  using namespace mxg;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __builtin_assume(wave >= 0 && wave < 4);
  const int hi = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  const long long M = a.M;

  MxConst kc;
  kc.K = a.K;
  kc.M = M;
  kc.tiles_n = a.tiles_n;
  kc.ntiles = (int)((M + BM - 1) / BM) * a.tiles_n;
  kc.G = (int)gridDim.x;
  kc.nk4 = a.K / (4 * ROWB);
  const int ntiles = kc.ntiles, G = kc.G, nk4 = kc.nk4;

  // staging: wave w moves rows 64w .. 64w+63 of either tile, 16 rows x 4 chunks of 16 B per instruction; the source chunk
  // is XOR-swizzled (chunk ^ ((row >> 2) & 3), the same for every 16-row group) so the fragment reads are bank-conflict
  // free.  ONE lane offset; the 16-row group and the K position ride in the scalar offset
  const unsigned voff = (unsigned)((lane >> 2) * a.K + (((lane & 3) ^ ((lane >> 4) & 3)) << 4));
  const int row_step = 16 * a.K;                               // bytes between consecutive 16-row groups
  const int wave_row0 = wave * 64 * a.K;
  MxState st;
  const unsigned char *ca_base = a.x, *cw_base = a.w, *na_base = a.x, *nw_base = a.w;
  int ca_nrec = 0, cw_nrec = 0, c_soff = 0, na_nrec = 0, nw_nrec = 0, n_soff = 0;
  int n_m0 = 0, n_n0 = 0, n_k4 = 0;
  auto publish = [&](const MxState& q) {                      // cur <- nxt, nxt <- q
    ca_base = na_base; cw_base = nw_base; ca_nrec = na_nrec; cw_nrec = nw_nrec; c_soff = n_soff;
    na_base = q.a_base; nw_base = q.w_base; na_nrec = q.a_nrec; nw_nrec = q.w_nrec; n_soff = q.soff;
    n_m0 = q.m0; n_n0 = q.n0; n_k4 = q.n_k4;
  };
  // one of the 8 LDS-DMA instructions of a stage (j = 0..3: activation rows, 4..7: weight rows)
  auto stage1 = [&](auto slotc, auto jc, const unsigned char* ab, int anrec, const unsigned char* wb, int wnrec, int soff) {
    constexpr int slot = decltype(slotc)::value, j = decltype(jc)::value, jj = j & 3;
    const auto srd = __builtin_amdgcn_make_buffer_rsrc((void*)(j < 4 ? ab : wb), (short)0, j < 4 ? anrec : wnrec, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_ptr_t)(smem + slot * ST + (j < 4 ? 0 : A_ST) + wave * 4096 + jj * 1024), 16, voff,
                                             soff + wave_row0 + jj * row_step, 0, 0);
  };
  auto stage = [&](auto slotc, const unsigned char* ab, int anrec, const unsigned char* wb, int wnrec, int soff) {
    stage1(slotc, std::integral_constant<int, 0>{}, ab, anrec, wb, wnrec, soff); stage1(slotc, std::integral_constant<int, 1>{}, ab, anrec, wb, wnrec, soff);
    stage1(slotc, std::integral_constant<int, 2>{}, ab, anrec, wb, wnrec, soff); stage1(slotc, std::integral_constant<int, 3>{}, ab, anrec, wb, wnrec, soff);
    stage1(slotc, std::integral_constant<int, 4>{}, ab, anrec, wb, wnrec, soff); stage1(slotc, std::integral_constant<int, 5>{}, ab, anrec, wb, wnrec, soff);
    stage1(slotc, std::integral_constant<int, 6>{}, ab, anrec, wb, wnrec, soff); stage1(slotc, std::integral_constant<int, 7>{}, ab, anrec, wb, wnrec, soff);
  };

  // fragment addresses: token rows wm*128 + p*32 + l31, weight rows wn*128 + i*32 + l31.  Operand layout of the 8-bit
  // 32x32x64 MFMA (probed on the hardware, tools/archive/mxprobe.py): lane (row = l & 31, h = l >> 5) holds in registers 0-3 the
  // row's K bytes [16 h, 16 h + 16) and in registers 4-7 the bytes [32 + 16 h, 32 + 16 h + 16) - one 16-byte piece of EACH
  // 32-element scale block - while the scale of block b is taken from the lanes with h = b.  So a lane reads the 16-byte
  // chunks h and 2 + h of the 64-byte row, at swizzled positions that do not depend on p (p*32 rows leave (row>>2)&3
  // alone): per operand TWO lane registers (chunk 0 / 1) - row block p, ring slot and operand half are immediates.  The
  // masks make the sign bit provably zero, which is what lets the compiler fold the constants into the ds_read offset
  // field (it otherwise re-derives every address with a v_add, or needs one register per (slot, p, chunk) as gemm4x does)
  unsigned ab_[2][2], bb_[2][2];                              // [ring half: slots 0-1 / 2-3][chunk 0 / 1 of the lane's 32 bytes]
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int ra = wm * 128 + l31, rb = wn * 128 + l31;
    ab_[0][j] = (unsigned)(ra * ROWB + (((j * 2 + hi) ^ ((ra >> 2) & 3)) << 4)) & 0x7fffu;
    bb_[0][j] = (unsigned)(A_ST + rb * ROWB + (((j * 2 + hi) ^ ((rb >> 2) & 3)) << 4)) & 0x7fffu;
    ab_[1][j] = ab_[0][j] + 2u * ST;
    bb_[1][j] = bb_[0][j] + 2u * ST;
    asm volatile("" : "+v"(ab_[1][j]), "+v"(bb_[1][j]));      // keep them as registers (immediates must stay below 64 KB)
  }
  auto frag = [&](const unsigned (&base)[2][2], int slot, int p) -> v8i {
    const int imm = (slot & 1) * ST + p * 32 * ROWB;
    const u32x4 lo = *(const u32x4*)(smem + base[slot >> 1][0] + imm);
    const u32x4 hh = *(const u32x4*)(smem + base[slot >> 1][1] + imm);
    return v8i{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hh[0], (int)hh[1], (int)hh[2], (int)hh[3]};
  };
  auto load_frag = [&](const unsigned (&base)[2][2], auto slotc, v8i (&f)[4]) {
    constexpr int slot = decltype(slotc)::value;
#pragma unroll
    for (int p = 0; p < 4; ++p) f[p] = frag(base, slot, p);
  };

  // block scales of the chunk being multiplied (c*) and of the next one (n*): one dword per fragment row = 4 K-steps.
  // The descriptor covers exactly chunk k4's [rows][2] words, so rows past M / N read 0 (= 2^-127, times zero data)
  // The loads are inline asm on purpose: beside LDS-DMA the compiler waits vmcnt(0) for the first use of an ordinary
  // VGPR-destination load (it did, once per chunk: the three-steps-ahead operand stream drained every 4 K-steps).  Here the
  // 8 dword loads of the NEXT chunk are issued right before step 0, older than that step's 8 LDS-DMA instructions, so the
  // `vmcnt(8)` of step 1 retires them; their registers are first read a whole chunk later (guide 5.7 item 1, form iii).
  const unsigned lane_s = (unsigned)((l31 * 2 + hi) * 4);
  unsigned sxA[4], swA[4], sxB[4], swB[4];              // two scale sets: chunks alternate, no copies (K % 512 == 0)
  auto srd_words = [&](const void* ptr, int nrec) -> u32x4 {
    const unsigned long long v = (unsigned long long)ptr;
    return u32x4{(unsigned)__builtin_amdgcn_readfirstlane((int)v), (unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) & 0xffffu,
                 (unsigned)__builtin_amdgcn_readfirstlane(nrec), 0x00020000u};
  };
  auto load_scales = [&](unsigned (&sx)[4], unsigned (&sw)[4], int m0, int n0, int k4) {
    const u32x4 dx = srd_words(a.xs + (long long)k4 * M * 2, (int)(M * 8));
    const u32x4 dw = srd_words(a.ws + (long long)k4 * a.N * 2, a.N * 8);
    const int ox = __builtin_amdgcn_readfirstlane((m0 + wm * 128) * 8), ow = __builtin_amdgcn_readfirstlane((n0 + wn * 128) * 8);
    asm volatile(
        "s_nop 4\n\t"
        "buffer_load_dword %0, %8, %9, %11 offen\n\t"
        "buffer_load_dword %1, %8, %9, %11 offen offset:256\n\t"
        "buffer_load_dword %2, %8, %9, %11 offen offset:512\n\t"
        "buffer_load_dword %3, %8, %9, %11 offen offset:768\n\t"
        "buffer_load_dword %4, %8, %10, %12 offen\n\t"
        "buffer_load_dword %5, %8, %10, %12 offen offset:256\n\t"
        "buffer_load_dword %6, %8, %10, %12 offen offset:512\n\t"
        "buffer_load_dword %7, %8, %10, %12 offen offset:768"
        : "=&v"(sx[0]), "=&v"(sx[1]), "=&v"(sx[2]), "=&v"(sx[3]), "=&v"(sw[0]), "=&v"(sw[1]), "=&v"(sw[2]), "=&v"(sw[3])
        : "v"(lane_s), "s"(dx), "s"(dw), "s"(ox), "s"(ow)
        : "memory");
  };

  f32x16 acc[4][4];
  using I0 = std::integral_constant<int, 0>;

  // ---- prologue (once per workgroup): K-steps 0, 1, 2 of the first tile ----
  mx_open_tile(st, a, kc, (int)blockIdx.x);
  publish(st);                                                // nxt = chunk 0 of the first tile
  load_scales(sxA, swA, n_m0, n_n0, n_k4);
  stage(std::integral_constant<int, 0>{}, na_base, na_nrec, nw_base, nw_nrec, n_soff);
  stage(std::integral_constant<int, 1>{}, na_base, na_nrec, nw_base, nw_nrec, n_soff + ROWB);
  stage(std::integral_constant<int, 2>{}, na_base, na_nrec, nw_base, nw_nrec, n_soff + 2 * ROWB);
  mx_advance(st, a, kc);
  publish(st);                                                // cur = chunk 0, nxt = its successor
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // fragments: activations single-buffered (x[p] is re-filled for the next step as soon as its four MFMAs have issued),
  // weights double-buffered (every MFMA of a step reads them); 96 registers instead of 128 for two full sets
  v8i xf[4], we[4], wo[4];
  load_frag(ab_, I0{}, xf);
  load_frag(bb_, I0{}, we);

  // one K-step (64 deep) of chunk cur; u = step within the chunk = its ring slot = the scale byte
  auto step = [&](auto uc, const unsigned (&csx)[4], const unsigned (&csw)[4]) {
    constexpr int u = decltype(uc)::value;
    using NSlot = std::integral_constant<int, (u + 1) & 3>;
    using SSlot = std::integral_constant<int, (u + 3) & 3>;
    // all but the previous step's 8 LDS-DMA loads have landed (step 0 of a chunk: the next chunk's 8 scale loads were
    // issued after them and may stay in flight too), then the step barrier (LDS hand-off point)
    if (u == 0) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- from here to the end of the step: ONE basic block, in EXACTLY this order (a sched_barrier after every MFMA
    // group: with one wave per SIMD nothing else covers a misplaced instruction).  MFMA m = (activation row block p = m / 4,
    // weight row block i = m % 4).  Fillers: gaps 0..3 the next step's weight fragments (2 reads each), gap 4 / 8 / 12 the
    // re-fill of x[0] / x[1] / x[2] right behind their own four MFMAs (x[3]: behind the last one), gaps 8..15 one LDS-DMA
    // instruction each (the stage three steps ahead) ----
    v8i (&wc)[4] = (u & 1) ? wo : we;
    v8i (&wn_)[4] = (u & 1) ? we : wo;
    const unsigned char* s_ab = u == 0 ? ca_base : na_base;
    const unsigned char* s_wb = u == 0 ? cw_base : nw_base;
    const int s_an = u == 0 ? ca_nrec : na_nrec, s_wn = u == 0 ? cw_nrec : nw_nrec;
    const int s_so = u == 0 ? c_soff + 3 * ROWB : n_soff + (u - 1) * ROWB;      // step 3 of this chunk / steps 0..2 of the next
    auto grp = [&](auto mc) {
      constexpr int m = decltype(mc)::value, p = m >> 2, i = m & 3;
      acc[i][p] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wc[i], xf[p], acc[i][p], 0, 0, u, csw[i], u, csx[p]);
      if (m < 4) wn_[m] = frag(bb_, (u + 1) & 3, m);
      if (m >= 4 && i == 0) xf[p - 1] = frag(ab_, (u + 1) & 3, p - 1);
      if (m >= 8) stage1(SSlot{}, std::integral_constant<int, (m >= 8 ? m - 8 : 0)>{}, s_ab, s_an, s_wb, s_wn, s_so);
      __builtin_amdgcn_sched_barrier(0);
    };
    grp(std::integral_constant<int, 0>{}); grp(std::integral_constant<int, 1>{}); grp(std::integral_constant<int, 2>{});
    grp(std::integral_constant<int, 3>{}); grp(std::integral_constant<int, 4>{}); grp(std::integral_constant<int, 5>{});
    grp(std::integral_constant<int, 6>{}); grp(std::integral_constant<int, 7>{}); grp(std::integral_constant<int, 8>{});
    grp(std::integral_constant<int, 9>{}); grp(std::integral_constant<int, 10>{}); grp(std::integral_constant<int, 11>{});
    grp(std::integral_constant<int, 12>{}); grp(std::integral_constant<int, 13>{}); grp(std::integral_constant<int, 14>{});
    grp(std::integral_constant<int, 15>{});
    xf[3] = frag(ab_, (u + 1) & 3, 3);
    __builtin_amdgcn_sched_barrier(0);
    // pin the step's MFMAs here: they touch no memory, so neither the barrier nor the memory clobbers keep the compiler from
    // sinking all 64 of a chunk into the loop latch (it did: every fragment then lives until the end of the chunk and spills)
    asm volatile("" : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[0][3]), "+a"(acc[1][0]), "+a"(acc[1][1]),
                      "+a"(acc[1][2]), "+a"(acc[1][3]), "+a"(acc[2][0]), "+a"(acc[2][1]), "+a"(acc[2][2]), "+a"(acc[2][3]),
                      "+a"(acc[3][0]), "+a"(acc[3][1]), "+a"(acc[3][2]), "+a"(acc[3][3]));
  };

  // epilogue-side lane role: 8 lanes x 8 columns cover 64 columns (128 B) of one output row
  const int e_px = lane >> 3, e_ch = lane & 7;
  for (int tile = (int)blockIdx.x; tile < ntiles; tile += G) {
    const MxTile c = mx_decode(kc, tile);
    const int col0 = c.n0 + wn * 128;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][p][r] = 0.f;

    for (int kq = 0; kq < nk4; kq += 2) {                       // two chunks per trip: the scale sets swap roles
      load_scales(sxB, swB, n_m0, n_n0, n_k4);                  // scales of the NEXT chunk (see load_scales)
      step(std::integral_constant<int, 0>{}, sxA, swA);
      step(std::integral_constant<int, 1>{}, sxA, swA);
      step(std::integral_constant<int, 2>{}, sxA, swA);
      step(std::integral_constant<int, 3>{}, sxA, swA);
      __builtin_amdgcn_sched_barrier(0);
      mx_advance(st, a, kc);
      publish(st);
      __builtin_amdgcn_sched_barrier(0);
      load_scales(sxA, swA, n_m0, n_n0, n_k4);
      step(std::integral_constant<int, 0>{}, sxB, swB);
      step(std::integral_constant<int, 1>{}, sxB, swB);
      step(std::integral_constant<int, 2>{}, sxB, swB);
      step(std::integral_constant<int, 3>{}, sxB, swB);
      __builtin_amdgcn_sched_barrier(0);
      mx_advance(st, a, kc);
      publish(st);
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---- epilogue: the wave's 128 x 128 result, one 32-row block x 64 columns at a time through its own 8 KB LDS slice
    // (fp32, XOR-swizzled 256-B rows), then 16-B stores with 8 lanes covering a full 128-B line ----
    {
      f32x4 bias_r[2][2], gate_r[2][2][2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int cb = col0 + h * 64 + e_ch * 8;
        bias_r[h][0] = bias_r[h][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.bias) { bias_r[h][0] = *(const f32x4*)(a.bias + cb); bias_r[h][1] = *(const f32x4*)(a.bias + cb + 4); }
        if (kGate) {
#pragma unroll
          for (int cls = 0; cls < 2; ++cls) {
            gate_r[cls][h][0] = *(const f32x4*)(a.gate + (long long)cls * a.N + cb);
            gate_r[cls][h][1] = *(const f32x4*)(a.gate + (long long)cls * a.N + cb + 4);
          }
        }
      }
      char* const eslice = smem + EPI + wave * 8192;
      unsigned o_off[4], r_off[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int px = it * 8 + e_px;
        o_off[it] = (unsigned)((px * (int)a.ldo + e_ch * 8) * 2);
        r_off[it] = (unsigned)((px * (int)a.ldr + e_ch * 8) * 2);
      }
      auto emit = [&](auto has_resid) {
        constexpr bool kRes = decltype(has_resid)::value;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const long long row0 = (long long)c.m0 + wm * 128 + p * 32;
          const long long vl = M - row0;
          const int rows = vl >= 32 ? 32 : (vl > 0 ? (int)vl : 0);     // rows past M: offset >= num_records -> dropped
          const auto srd_o = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + row0 * a.ldo + col0), (short)0,
                                                               rows * (int)a.ldo * 2, 0x00020000);
          const auto srd_r = __builtin_amdgcn_make_buffer_rsrc((void*)(kRes ? a.resid + row0 * a.ldr + col0 : a.out), (short)0,
                                                               kRes ? rows * (int)a.ldr * 2 : 0, 0x00020000);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            u32x4 rr[4];
            if (kRes) {
#pragma unroll
              for (int it = 0; it < 4; ++it) rr[it] = __builtin_amdgcn_raw_buffer_load_b128(srd_r, (int)r_off[it], h * 128, 0);
            }
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
              for (int gq = 0; gq < 4; ++gq) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = acc[h * 2 + i2][p][gq * 4 + e];
                const int ch = i2 * 8 + 2 * gq + hi;                   // 16-B chunk of the 256-B row
                *(f32x4*)(eslice + l31 * 256 + ((ch ^ (l31 & 15)) << 4)) = o;
              }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // wave-private slice: no barrier needed
            f32x4 lo[4], hi4[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
              const int px = it * 8 + e_px;
              lo[it] = *(const f32x4*)(eslice + px * 256 + (((2 * e_ch) ^ (px & 15)) << 4));
              hi4[it] = *(const f32x4*)(eslice + px * 256 + (((2 * e_ch + 1) ^ (px & 15)) << 4));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int it = 0; it < 4; ++it) {
              f32x4 x0 = lo[it] + bias_r[h][0], x1 = hi4[it] + bias_r[h][1];
              if (kAct) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { x0[e] = gelu_tanh_f(x0[e]); x1[e] = gelu_tanh_f(x1[e]); }
              }
              if (kRes) {
                const u32x4 r = rr[it];
                const f32x4 r0 = {__uint_as_float(r[0] << 16), __uint_as_float(r[0] & 0xffff0000u), __uint_as_float(r[1] << 16),
                                  __uint_as_float(r[1] & 0xffff0000u)};
                const f32x4 r1 = {__uint_as_float(r[2] << 16), __uint_as_float(r[2] & 0xffff0000u), __uint_as_float(r[3] << 16),
                                  __uint_as_float(r[3] & 0xffff0000u)};
                if (kGate) {
                  const bool vid = row0 + it * 8 + e_px >= a.gate_split;   // row class: text rows first
                  const f32x4 g0 = vid ? gate_r[1][h][0] : gate_r[0][h][0];
                  const f32x4 g1 = vid ? gate_r[1][h][1] : gate_r[0][h][1];
                  x0 = r0 + g0 * x0;
                  x1 = r1 + g1 * x1;
                } else {
                  x0 += r0;
                  x1 += r1;
                }
              }
              const u32x4 v = {pack_bf2(x0[0], x0[1]), pack_bf2(x0[2], x0[3]), pack_bf2(x1[0], x1[1]), pack_bf2(x1[2], x1[3])};
              __builtin_amdgcn_raw_buffer_store_b128(v, srd_o, (int)o_off[it], h * 128, 0);
              // store-data hazard (see gemm4x): keep the data registers untouched for a few cycles after the 16-B store
              __builtin_amdgcn_sched_barrier(0);
              asm volatile("s_nop 3" ::: "memory");
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
      };
      if (a.resid) emit(std::true_type{});
      else emit(std::false_type{});
    }
    // the counted-vmcnt scheme of the K walk restarts from an empty queue (stores count in vmcnt on gfx9).  The BUILTIN form,
    // so the compiler's own scoreboard sees the epilogue's loads / stores retired: with an asm wait it guarded the first
    // scale-load asm of the next tile walk (which redefines registers the epilogue used) with a vmcnt(0) of its own - at the
    // loop header, i.e. once per two chunks, draining the three-steps-ahead operand stream
    __builtin_amdgcn_s_waitcnt(0x0F70);                        // vmcnt(0), expcnt / lgkmcnt untouched
    asm volatile("" ::: "memory");
  }
}

extern "C" int dove_linear_mxfp8(const void* xq, const void* xs, const void* wq, const void* ws, const float* bias, const void* resid,
                                 const float* gate, void* out, long long M, int N, int K, long long ldo, long long ldr,
                                 long long gate_split, int act, void* stream) {
  DOVE_CHECK_ARG(xq && xs && wq && ws && out, "linear_mxfp8: null pointer");
  DOVE_CHECK_ARG(M > 0 && N > 0 && N % 256 == 0, "linear_mxfp8: N (%d) must be a positive multiple of 256", N);
  DOVE_CHECK_ARG(K > 0 && K % 512 == 0, "linear_mxfp8: K (%d) must be a positive multiple of 512", K);
  DOVE_CHECK_ARG((long long)256 * K < (1ll << 31) && (long long)(K / 256) * M * 8 < (1ll << 31) && ldo < (1 << 20) && ldr < (1 << 20),
                 "linear_mxfp8: operand too large for 31-bit buffer offsets");
  DOVE_CHECK_ARG(ldo % 4 == 0 && ldo >= N && (!resid || (ldr % 4 == 0 && ldr >= N)), "linear_mxfp8: bad ldo / ldr");
  DOVE_CHECK_ARG(!gate || resid, "linear_mxfp8: gate needs resid");
  DOVE_CHECK_ARG(act == 0 || act == 1, "linear_mxfp8: act must be 0 or 1 (GELU tanh)");
  DOVE_CHECK_ARG(!(gate && act), "linear_mxfp8: gate with activation is not a path of the reference");
  MxArgs a;
  a.x = (const unsigned char*)xq; a.w = (const unsigned char*)wq; a.xs = (const unsigned*)xs; a.ws = (const unsigned*)ws;
  a.bias = bias; a.resid = (const bf16_t*)resid; a.gate = gate; a.out = (bf16_t*)out;
  a.M = M; a.ldo = ldo; a.ldr = ldr; a.gate_split = gate_split; a.N = N; a.K = K; a.tiles_n = N / 256;
  static PerDeviceOnce attr;
  if (auto once_ = attr.guard()) {
    (void)hipFuncSetAttribute((const void*)gemm_mxfp8_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, mxg::LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm_mxfp8_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, mxg::LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm_mxfp8_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, mxg::LDS_BYTES);
  }
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const long long nt = ((M + mxg::BM - 1) / mxg::BM) * a.tiles_n;
  DOVE_CHECK_ARG(nt < (1ll << 31), "linear_mxfp8: grid too large");
  const unsigned grid = nt > cus ? (unsigned)cus : (unsigned)nt;
  hipStream_t s = (hipStream_t)stream;
  if (gate) hipLaunchKernelGGL((gemm_mxfp8_kernel<false, true>), dim3(grid), dim3(256), mxg::LDS_BYTES, s, a);
  else if (act == 1) hipLaunchKernelGGL((gemm_mxfp8_kernel<true, false>), dim3(grid), dim3(256), mxg::LDS_BYTES, s, a);
  else hipLaunchKernelGGL((gemm_mxfp8_kernel<false, false>), dim3(grid), dim3(256), mxg::LDS_BYTES, s, a);
  DOVE_CHECK_LAUNCH("dove_linear_mxfp8");
  return DOVE_OK;
}
