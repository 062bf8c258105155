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
#include "igemm_args.h"

#ifdef DOVE_TIMING_BUILD
void* g_timing_debug_buf = nullptr;
extern "C" int dove_timing_set_debug_buf(void* p) { g_timing_debug_buf = p; return 0; }
#endif

// ------------------------------------------------------------------------------------------------
// Fast path (up == 0: every conv except the upsample-fused one, and every linear).
// Same tiling / swizzle / MFMA order as igemm_kernel, but the staging is rebuilt around
// `buffer_load_dwordx4 ... lds` through wave-uniform buffer descriptors (PMC on v1: 9.8 VALU + 7.6 SALU
// instructions per MFMA, all of it 64-bit im2col address math):
//   * per-thread row byte offsets inside a frame and a 9-bit tap-validity mask are computed ONCE;
//   * per K-step a load costs v_and/v_cmp/v_add/v_cndmask: invalid (padding / tail) lanes get an
//     out-of-range offset and the descriptor's bounds check returns zeros -- no zero page, no branches;
//   * tap / channel-chunk / frame advance is scalar state (SGPR adds), the K offset rides in soffset;
//   * ds_read addresses are per-lane constants + compile-time (buffer, operand) immediates (loop unrolled x2).
// ------------------------------------------------------------------------------------------------

template <int BN, int BK>
struct FastCfg {
  static constexpr int BM = 128;
  static constexpr int CPR = BK / 8;
  static constexpr int CPR_LOG = (BK == 64) ? 3 : 2;
  static constexpr int RPG = 256 / CPR;
  static constexpr int NA = BM / RPG;
  static constexpr int B_SLOTS = BN * CPR;
  static constexpr int NB = (B_SLOTS + 255) / 256;
  static constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
  static constexpr int WN = (BN >= 128) ? 2 : 1, WM = 4 / WN;
  static constexpr int PT = (BM / WM) / 32, CT = (BN / WN) / 32;
  static constexpr int KK = BK / 16;
};

template <int BN, int BK>
__global__ __launch_bounds__(256) void igemm_fast_kernel(const IgemmArgs a) {
  using Cf = FastCfg<BN, BK>;
  constexpr int NA = Cf::NA, NB = Cf::NB, PT = Cf::PT, CT = Cf::CT, KK = Cf::KK, STAGE = Cf::STAGE, A_BYTES = Cf::A_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __builtin_assume(wave >= 0 && wave < 4);
  const int wm = wave / Cf::WN, wn = wave % Cf::WN;

  unsigned rest = xcd_remap(blockIdx.x, gridDim.x);
  const int tn = rest % a.tiles_n; rest /= a.tiles_n;
  const int twi = rest % a.tiles_w; rest /= a.tiles_w;
  const int thi = rest % a.tiles_h;
  const int t = rest / a.tiles_h;
  const int n0 = tn * BN;
  const int TWm = (1 << a.tw_log2) - 1;
  const int oh0 = thi * (128 >> a.tw_log2), ow0 = twi << a.tw_log2;

  // ---- one-time per-thread staging geometry ----
  const int cs = tid & (Cf::CPR - 1);
  const int rsub = tid >> Cf::CPR_LOG;
  const int c = (BK == 64) ? (cs ^ ((rsub >> 1) & 7)) : (cs ^ ((rsub >> 2) & 3));
  int rowoff[NA];
  unsigned mask[NA];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int m = j * Cf::RPG + rsub;
    const int oh = oh0 + (m >> a.tw_log2), ow = ow0 + (m & TWm);
    const bool mv = (oh < a.H_out) && (ow < a.W_out);
    const int ih0 = oh * a.stride - a.pad_h, iw0 = ow * a.stride - a.pad_w;
    rowoff[j] = ((ih0 * a.W_in + iw0) * a.Cin + c * 8) * 2;
    unsigned mk = 0;
    for (int dh = 0; dh < a.kh; ++dh)
      for (int dw = 0; dw < a.kw; ++dw) {
        const bool ok = mv && ((unsigned)(ih0 + dh) < (unsigned)a.H_in) && ((unsigned)(iw0 + dw) < (unsigned)a.W_in);
        mk |= (ok ? 1u : 0u) << (dh * a.kw + dw);
      }
    mask[j] = mk;
  }
  int boff_g[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) boff_g[j] = ((j * Cf::RPG + rsub) * a.Cin + c * 8) * 2;

  const long long frame_elems = (long long)a.H_in * a.W_in * a.Cin;
  const unsigned frame_bytes = (unsigned)(frame_elems * 2);
  const unsigned wtap_bytes = (unsigned)((long long)BN * a.Cin * 2);     // bytes of this block's weight rows per tap
  const long long wtap_stride = (long long)a.Cout_pad * a.Cin;            // elements between taps
  const int kc_per_tap = a.Cin / BK;
  const int nk = a.kt * a.kh * a.kw * kc_per_tap;

  // ---- scalar K-walk state (the NEXT K-step to stage); everything below lives in SGPRs ----
  auto frame_ptr = [&](int dt) -> const bf16_t* { return igemm_src_frame(a, t, dt, frame_elems); };
  int s_kc = 0, s_dw = 0, s_dh = 0, s_dt = 0;
  const bf16_t* s_fp = frame_ptr(0);
  const bf16_t* s_wp = a.w + (long long)n0 * a.Cin;
  int s_tapdelta = 0;
  unsigned s_tapbit = 1u;

  auto stage = [&](auto bufc) {
    constexpr int BUF = decltype(bufc)::value;
    const auto srd_a = __builtin_amdgcn_make_buffer_rsrc((void*)s_fp, (short)0, (int)frame_bytes, 0x00020000);
    const auto srd_b = __builtin_amdgcn_make_buffer_rsrc((void*)s_wp, (short)0, (int)wtap_bytes, 0x00020000);
    const int soff = s_kc * (BK * 2);
    if (!(DOVE_DBG(a) & 1)) {
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const unsigned voff = (mask[j] & s_tapbit) ? (unsigned)(rowoff[j] + s_tapdelta) : 0x80000000u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_a, (lds_ptr_t)(smem + BUF * STAGE + (j * 256 + wave * 64) * 16), 16, voff, soff, 0, 0);
    }
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      if (j * 256 + wave * 64 < Cf::B_SLOTS && !(DOVE_DBG(a) & 2))
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_b, (lds_ptr_t)(smem + BUF * STAGE + A_BYTES + (j * 256 + wave * 64) * 16), 16,
                                                 (unsigned)boff_g[j], soff, 0, 0);
    }
    // advance to the next K-step (tap / frame changes are rare: every Cin/BK steps)
    if (++s_kc == kc_per_tap) {
      s_kc = 0;
      s_wp += wtap_stride;
      if (++s_dw == a.kw) {
        s_dw = 0;
        if (++s_dh == a.kh) {
          s_dh = 0;
          ++s_dt;
          if (s_dt < a.kt) s_fp = frame_ptr(s_dt);
        }
      }
      s_tapdelta = ((s_dh * a.W_in + s_dw) * a.Cin) * 2;
      s_tapbit = 1u << (s_dh * a.kw + s_dw);
    }
  };

  // ---- per-lane LDS fragment offsets (constant over the K loop) ----
  const int hi = lane >> 5, l31 = lane & 31;
  int aoff[PT][KK], boff[CT][KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) {
    const int chunk = kk * 2 + hi;
#pragma unroll
    for (int p = 0; p < PT; ++p) {
      const int row = wm * (Cf::BM / Cf::WM) + p * 32 + l31;
      const int sc = (BK == 64) ? (chunk ^ ((row >> 1) & 7)) : (chunk ^ ((row >> 2) & 3));
      aoff[p][kk] = row * (BK * 2) + sc * 16;
    }
#pragma unroll
    for (int i = 0; i < CT; ++i) {
      const int row = wn * (BN / Cf::WN) + i * 32 + l31;
      const int sc = (BK == 64) ? (chunk ^ ((row >> 1) & 7)) : (chunk ^ ((row >> 2) & 3));
      boff[i][kk] = row * (BK * 2) + sc * 16;
    }
  }

  f32x16 acc[CT][PT];
#pragma unroll
  for (int i = 0; i < CT; ++i)
#pragma unroll
    for (int p = 0; p < PT; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][p][r] = 0.f;

  auto compute = [&](auto bufc) {
    constexpr int BUF = decltype(bufc)::value;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      bf16x8 xf[PT], wf[CT];
#pragma unroll
      for (int p = 0; p < PT; ++p) xf[p] = *(const bf16x8*)(smem + BUF * STAGE + aoff[p][kk]);
#pragma unroll
      for (int i = 0; i < CT; ++i) wf[i] = *(const bf16x8*)(smem + BUF * STAGE + A_BYTES + boff[i][kk]);
      if (DOVE_DBG(a) & 4) {
#pragma unroll
        for (int i = 0; i < CT; ++i)
#pragma unroll
          for (int p = 0; p < PT; ++p) { asm volatile("" ::"v"(wf[i]), "v"(xf[p])); }
        continue;
      }
#pragma unroll
      for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int p = 0; p < PT; ++p)
          acc[i][p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], xf[p], acc[i][p], 0, 0, 0);
    }
  };

  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  stage(B0{});
  int it = 0;
  for (; it + 2 <= nk; it += 2) {          // K-steps in pairs: LDS buffer index is a compile-time constant
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    stage(B1{});
    compute(B0{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (it + 2 < nk) stage(B0{});
    compute(B1{});
  }
  if (nk & 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    compute(B0{});
  }

  // ---- epilogue (the generic one: igemm_legacy.hip's igemm_kernel has the same) ----
#pragma unroll
  for (int p = 0; p < PT; ++p) {
    const int m = wm * (Cf::BM / Cf::WM) + p * 32 + l31;
    const int oh = oh0 + (m >> a.tw_log2), ow = ow0 + (m & TWm);
    if (!((oh < a.H_out) && (ow < a.W_out))) continue;
    const long long pix = ((long long)t * a.H_out + oh) * a.W_out + ow;
    const float* gate = a.gate ? (a.gate + (pix < a.gate_split ? 0 : a.Cout_pad)) : nullptr;
#pragma unroll
    for (int i = 0; i < CT; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cb = n0 + wn * (BN / Cf::WN) + i * 32 + 8 * g + 4 * hi;
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
        if (a.out_f32) *(f32x4*)((float*)a.out + pix * a.ldo + cb) = f32x4{v[0], v[1], v[2], v[3]};
        else *(uint2*)(a.out + pix * a.ldo + cb) = o;
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------
// 3x3 (x kt) stride-1 convolutions with an LDS HALO tile.  History of the design (profiles/r01_*): ablation of the
// igemm_fast path showed global->LDS staging + ds_reads alone (no MFMA) costing 76 % of the launch - the per-CU
// vector-memory path (~64 B/clk) is the limiter at 64 FLOP per staged byte.  A halo kernel walks K as
//   for dt (temporal tap) / for kc (32-channel chunk):   stage the (TH+2) x (TW+2) input halo ONCE
//     for the 9 spatial taps:                              stage only the 128 x 32 weight tile
// so the activations are fetched once per 9 taps (204 FLOP per staged byte).  Fragments for tap (dh, dw) are the same
// halo rows shifted by dh*(TW+2) + dw.  The first (4-wave, 8 x 32 pixel) kernel of that family is gone; its shapes
// (H < 16) run on igemm_fast.
// ------------------------------------------------------------------------------------------------
// 8-wave ping-pong variant of the halo kernel.  PMC on the 4-wave kernel: MFMA busy 56 % of cycles, waves
// parked/stalled 70 % -- each wave interleaves staging, 12 ds_reads, ~30 VALU and 16 MFMAs per K-step and the
// two co-resident workgroups of a CU only overlap by chance.  Here ONE workgroup of 8 waves owns 16 x 32 pixels
// (two stacked 8 x 32 tiles sharing every weight tile -> 42 % fewer staged bytes) and is split in two groups
// of 4 waves that run the same two-phase step shifted by one barrier:
//     phase 1: issue the LDS-DMA loads for step s+2, ds_read the 12 fragments of step s into registers
//     phase 2: 16 MFMAs from registers only
// Every SIMD hosts one wave of each group, so while one is in its MFMA phase the other does memory work
// (cdna_hip_programming.md "8-phase" idea, role split by stagger instead of by wave specialisation).
// Hazards (interval k = time between barriers; group A: phase1(s)=2s, phase2(s)=2s+1; group B one later):
//   * weight tile of step s: ring slot s%4, loads issued in phase1(s-3) of both groups (LDS-DMA issue->landed is
//     ~1.1 us, longer than one step); each wave drains them with a COUNTED vmcnt before the barrier ending ITS
//     phase1(s-1) (<= interval 2s-1), leaving the loads of steps s-2 and s-1 in flight; first read interval 2s.
//     The slot is next overwritten in phase1(s+1) of group A = interval 2s+2, after group B's last read (2s+1).
//   * halo of group g+1: 5 rounds issued during steps 0..4 of group g into the other halo buffer, whose last
//     reader (group B, last step of group g-1) finished one barrier earlier.
// ------------------------------------------------------------------------------------------------
namespace halo8 {   // tile geometry / LDS image of the halo kernels (the name is historical: the 8-wave kernel that introduced it is gone)
constexpr int TH = 16, TW = 32, HWID = TW + 2, HHGT = TH + 2, HPIX = HWID * HHGT;   // 612 halo pixels
constexpr int BK = 32, ROWB = 64;
// Halo rows are PADDED to 80 bytes (5 x 16 B, last slot unused) instead of XOR-swizzled: slot index 5*row + chunk is
// distinct mod 16 for any 16 rows that are distinct mod 16, so ds_read_b128 stays conflict-free, and the fragment
// address becomes LINEAR in the row -> one base VGPR + immediate offsets for all 9 taps (zero VALU per step; timing
// showed the memory phase is slowed ~2x by its SIMD partner's MFMA stream, so its instruction count is what matters).
constexpr int APITCH = 80, ASLOTS = HPIX * 5;                                         // 3060 x 16 B
constexpr int A_ROUNDS = (ASLOTS + 511) / 512;                                        // 6 rounds of 512 x 16 B
constexpr int A_BYTES = A_ROUNDS * 512 * 16;                                          // 49152
constexpr int BN = 128, B_BYTES = BN * ROWB;                                          // 8192 = 512 x 16 B: one load per thread
constexpr int B_RING = 4;                                                             // weights run 3 steps ahead
constexpr int LDS_BYTES = 2 * A_BYTES + B_RING * B_BYTES;                             // 131072
}  // namespace halo8

// ------------------------------------------------------------------------------------------------
// conv3x3_halo4x: the same 16 x 32 pixel x 128 channel workgroup tile and LDS image as conv3x3_halo8, but ONE wave per
// SIMD owning the whole 512-entry register file (4 waves, 1 workgroup per CU).  The phase log of halo8 showed that the
// memory phase of a wave is slowed ~2x whenever its SIMD partner streams MFMAs, and the partner's MFMA phase stretches
// from 512 to ~680 cycles: two waves per SIMD fight for issue slots.  Here each wave computes 128 pixels x 128 channels
// (16 accumulator tiles = 256 registers, 8 fragment reads per 16 MFMAs = 0.5 KB of LDS traffic per MFMA) and software-
// pipelines its own fragments through registers: while the 16 MFMAs of k-half 0 run, the fragments of k-half 1 are read,
// and while those run, k-half 0 of the NEXT step is read -- legal across the step barrier because the counted vmcnt drain
// keeps every staged tile two steps ahead of its first reader.  One barrier per 32 MFMAs, no MFMA ever waits on LDS.
// ------------------------------------------------------------------------------------------------
struct Halo4xCfg {
  static constexpr int BR = 6;                                  // weight ring slots; 9 taps x 2 group parities = 3 turns:
                                                                //   slot(tap, parity) = (tap + 3 parity) % 6 is compile-time
  static constexpr int BAHEAD = 3;                              // weights are staged 3 steps ahead
  static constexpr int HPS = 2;                                 // halo rounds staged per step (steps 0..5 of a group)
  static constexpr int DS0 = 4;                                 // first MFMA gap that carries a k-half-1 fragment read
  static constexpr int LDS_BYTES = 2 * halo8::A_BYTES + BR * halo8::B_BYTES;   // 147456
  static constexpr int NT = 9;                                  // steps (spatial taps) per group
  static constexpr int nh(int tap, int nr) { return (HPS * tap + HPS <= nr) ? HPS : ((HPS * tap < nr) ? nr - HPS * tap : 0); }
  static constexpr int round0(int tap) { return HPS * tap; }   // first halo round staged in step `tap`
  static constexpr int issued(int tap, int nr) { return 2 + nh(tap, nr); }
  static constexpr int inflight(int tap, int nr) {              // loads issued in the BAHEAD-2 steps before step `tap`
    int n = 0;
    for (int d = 1; d <= BAHEAD - 2; ++d) n += issued((tap + 9 - d) % 9, nr);
    return n;
  }
};
// SUB-PIXEL form of the upsample-fused conv (kSub): a nearest x2 upsample followed by a 3x3 conv is, per output phase (oy & 1, ox & 1), a
// 2x2 conv on the LOW-RES input with the 3x3 weights summed over the taps that hit the same low-res pixel (dove_conv_desc.w_sub: sums in
// fp32, rounded once at pack time): 4 / 9 of the MACs.  The phase is tied to the cout tile (tiles_n = 4 x Cout / 128), so a workgroup tile is
// 16 x 32 LOW-RES pixels of ONE phase x 128 channels: the plain halo geometry and weight sharing of the non-upsampling kernel, groups of
// FOUR steps whose halo offsets (py + a, px + b) are per-tile VGPR bases.  The 12 halo rounds of the next group are staged in the first two
// steps (they must have landed at the barrier of the fourth, whose tail reads the next group's first fragments); a 4-slot weight ring.
struct Halo4xSubCfg {
  static constexpr int NT = 4, BR = 4, BAHEAD = 3, HPS = 6, DS0 = 4;
  static constexpr int nh(int tap, int nr) { return tap < 2 ? nr / 2 : 0; }
  static constexpr int round0(int tap) { return HPS * tap; }
  static constexpr int issued(int tap, int nr) { return 2 + nh(tap, nr); }
  static constexpr int inflight(int tap, int nr) { return issued((tap + NT - 1) % NT, nr); }
};

// Tile GEOMETRY of a halo4x launch (conv3x3_halo4x_kernel's kPart).  The register tile is always 4 waves x 8 pixel blocks of 16 pixels x 128 couts;
// what changes is how the 32 blocks are laid over the image:
//   kPart 0   16 rows x 32 columns   wave w: rows 4 w .. 4 w + 3, block idx -> (row idx >> 1, columns 16 (idx & 1) ..)      halo 18 x 34
//   kPart 1   32 rows x 16 columns   wave w: rows 8 w .. 8 w + 7, block idx -> (row idx, columns 0 .. 15)                    halo 34 x 18
// (An 8 x 64 geometry for a partial last tile ROW was built and measured too, commit history of round 6: under the rounds rule of halo4x_plan it
// never wins a product shape, and its GroupNorm partial sums cannot reproduce the 16 x 32 form's bit for bit - two waves share a slot - so it is gone.)
template <int kPart> struct H4Geo {
  static constexpr int TH = kPart == 1 ? 32 : 16, TW = kPart == 1 ? 16 : 32;
  static constexpr int HWID = TW + 2, HROWS = TH + 2, HPIX = HWID * HROWS;               // 612 halo pixels either way
  static constexpr int NR = (HPIX * 5 + 255) / 256;                                      // 12 halo rounds of 256 x 16 B
  static constexpr int A_BYTES = NR * 4096;                                              // 49152
  static constexpr int RPW = TH / 4;                                                     // tile rows per wave
  static constexpr int LDS_BYTES = 2 * A_BYTES + 6 * halo8::B_BYTES;                     // 147456
  static __device__ __forceinline__ constexpr int prow(int idx) { return kPart == 1 ? idx : idx >> 1; }
  static __device__ __forceinline__ constexpr int pcol(int idx) { return kPart == 1 ? 0 : (idx & 1); }
};
constexpr int H4_MAX_ROUNDS = 12;

// Staging-side state of the persistent halo4x walk.  Plain structs + force-inlined functions (not by-reference lambda
// closures nested three deep: those left the counters in scratch memory, where the compiler treats them as per-lane
// values and builds every buffer descriptor through a waterfall loop).
struct H4Tile { int n0, t, oh0, ow0, ph; };                 // ph: output phase 2 py + px of the sub-pixel form (0 otherwise)
struct H4Const {
  int ntiles, G, kcn, frame_bytes, wtap_bytes, tid;
  long long frame_elems, wtap_stride;
};
struct H4State {
  unsigned voffA[H4_MAX_ROUNDS];                              // lane offsets of the halo rounds (spatial tile of nxt)
  int n_tile, n_dt, n_kc, n_oh0, n_ow0;                       // group `nxt`: tile, frame tap, channel chunk
  bool n_on;                                                  // nxt's tile exists (else: zero-length descriptors)
  int t;                                                      // output frame of nxt's tile WITHIN its instance (dove_conv_desc.nb)
  long long f0;                                               // first input frame of that instance
  const bf16_t* cache_b;                                      // that instance's conv cache (nullptr: none)
  int ndt;                                                    // temporal groups of nxt's tile: a.kt, or 1 / 2 where taps read equal frames
  int split;                                                  //   how the three causal taps fall into groups (h4_split)
  const bf16_t* wf_base;                                      //   IgemmArgs.w_first (W0 + W1 | W0 + W1 + W2), cout tile applied
  const bf16_t* wp_base;                                      //   IgemmArgs.w_pair (W0 + W1 | W1 + W2), cout tile applied
  const bf16_t* wt_base;                                      // weights of nxt's cout tile
  const bf16_t* h_base;                                       // nxt's halo source: frame + channel chunk
  const bf16_t* wg_nxt;                                       // nxt's weights: tap 0 of (frame tap, chunk)
  int h_nrec, nrec_b_cur, nrec_b_nxt;
};
// a 64-bit select is lowered to v_cndmask and runtime integer division runs on the VALU: pin such results back to SGPRs
__device__ __forceinline__ const bf16_t* h4_pin64(const bf16_t* p) {
  const unsigned long long v = (unsigned long long)p;
  return (const bf16_t*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
                         (unsigned)__builtin_amdgcn_readfirstlane((int)v));
}
template <int kPart = 0>
__device__ __forceinline__ H4Tile h4_decode(const IgemmArgs& a, const H4Const& k, int id) {
  unsigned rest = xcd_remap((unsigned)(id < k.ntiles ? id : k.ntiles - 1), (unsigned)k.ntiles);
  H4Tile q;
  const int nidx = (int)(rest % a.tiles_n); rest /= a.tiles_n;
  const int ctn = a.sub ? a.tiles_n >> 2 : a.tiles_n;          // cout tiles (sub-pixel form: per phase)
  q.ph = __builtin_amdgcn_readfirstlane(a.sub ? nidx / ctn : 0);
  q.n0 = __builtin_amdgcn_readfirstlane((nidx - q.ph * ctn) * halo8::BN);
  q.t = __builtin_amdgcn_readfirstlane((int)(rest % a.T_out)); rest /= a.T_out;
  // the launch walks ntx x nty tiles of ITS geometry from the origin (tx0, ty0), which is given in 16 x 32 tiles
  q.ow0 = __builtin_amdgcn_readfirstlane(a.tx0 * halo8::TW + (int)(rest % a.ntx) * H4Geo<kPart>::TW);
  q.oh0 = __builtin_amdgcn_readfirstlane(a.ty0 * halo8::TH + (int)(rest / a.ntx) * H4Geo<kPart>::TH);
  return q;
}
// How the three causal taps (frames t - 2, t - 1, t of the instance; before its first frame: the conv cache, else frame 0 replicated) of the
// tile whose output frame (global index over the instances) is t fall into TEMPORAL GROUPS - taps that read bit-identical frames share one
// group on the pre-summed weights:
//   0  one group per tap (kt groups)                     the general case
//   1  {0 1 2}: (W0 + W1 + W2) x[t]                       w_first block 1
//   2  {0 1}{2}: (W0 + W1) x[t-1], then W2 x[t]           w_first / w_pair block 0, then tap block 2 of w
//   3  {0}{1 2}: W0 x[t-2], then (W1 + W2) x[t]           tap block 0 of w, then w_pair block 1
// Without a declaration (tdup == 0) only the cache-less first two frames of an instance split (IgemmArgs.w_first).  tdup == 1: the frames
// are pairs (0,1), (2,3), ... and the cache, if any, is a pair; tdup == 2: frame 0 single, then pairs (1,2), (3,4), ... (no cache).
__device__ __forceinline__ int h4_split(const IgemmArgs& a, int t) {
  if (a.kt != 3) return 0;
  const int tl = a.seg_out == a.T_out ? t : t % a.seg_out;
  const bool cached = a.cache != nullptr;
  if (a.tdup == 0) {
    if (a.w_first == nullptr || cached) return 0;
    return tl == 0 ? 1 : (tl == 1 ? 2 : 0);
  }
  if (a.tdup == 1) {
    if (!cached && tl < 2) return 1;                          // frames 0 and 1 are equal and so is the replicated front
    return (tl & 1) ? 3 : 2;
  }
  if (tl == 0) return 1;
  return (tl & 1) ? 2 : 3;
}
__device__ __forceinline__ int h4_split_ndt(const IgemmArgs& a, int split) { return split == 0 ? a.kt : (split == 1 ? 1 : 2); }
template <bool kUp, int kPart = 0>
__device__ __forceinline__ void h4_open_tile(H4State& s, const IgemmArgs& a, const H4Const& k, int id) {
  using namespace halo8;
  constexpr int UHW = halo8::TW / 2 + 2, UHH = halo8::TH / 2 + 2;
  s.n_tile = id;
  s.n_on = id < k.ntiles;
  const H4Tile q = h4_decode<kPart>(a, k, id);
  if (q.oh0 != s.n_oh0 || q.ow0 != s.n_ow0) {                // lane offsets are redone only on a new spatial tile
    s.n_oh0 = q.oh0; s.n_ow0 = q.ow0;
#pragma unroll
    for (int r = 0; r < (kUp ? 12 : H4Geo<kPart>::NR); ++r) {
      const int sl = r * 256 + k.tid;
      const int px = sl / 5, c = sl - px * 5;
      int ih, iw;
      bool inb;
      if (kUp) {
        const int hh = px / UHW, hw = px - hh * UHW;
        ih = (q.oh0 >> 1) - 1 + hh; iw = (q.ow0 >> 1) - 1 + hw;
        inb = px < UHW * UHH;
      } else {
        constexpr int HW = H4Geo<kPart>::HWID;                 // halo image of the launch's tile geometry
        const int hh = px / HW, hw = px - hh * HW;
        ih = q.oh0 - 1 + hh; iw = q.ow0 - 1 + hw;
        inb = px < H4Geo<kPart>::HPIX;
      }
      const bool ok = inb && (c < 4) && ((unsigned)ih < (unsigned)a.H_in) && ((unsigned)iw < (unsigned)a.W_in);
      s.voffA[r] = ok ? (unsigned)(((ih * a.W_in + iw) * a.Cin + c * 8) * 2) : 0x80000000u;
    }
  }
  const int b = a.seg_out == a.T_out ? 0 : __builtin_amdgcn_readfirstlane(q.t / a.seg_out);
  s.t = q.t - b * a.seg_out;
  s.f0 = (long long)b * a.seg_in;
  s.cache_b = a.cache ? h4_pin64(a.cache + (long long)b * a.cache_bs) : nullptr;
  s.wt_base = a.w + (long long)q.n0 * a.Cin + (long long)(q.ph * 4) * k.wtap_stride;   // sub-pixel form: [phase][2x2 tap][Cout_pad][Cin]
  s.split = __builtin_amdgcn_readfirstlane(h4_split(a, q.t));
  s.ndt = h4_split_ndt(a, s.split);
  s.wf_base = a.w_first ? a.w_first + (long long)q.n0 * a.Cin : nullptr;
  s.wp_base = a.w_pair ? a.w_pair + (long long)q.n0 * a.Cin : nullptr;
  s.n_dt = 0; s.n_kc = 0;
}
__device__ __forceinline__ void h4_set_nxt(H4State& s, const IgemmArgs& a, const H4Const& k) {   // descriptors of group nxt
  using namespace halo8;
  // source frame of (tile frame t, frame tap n_dt): causal taps before the first frame come from the conv cache (or
  // replicate frame 0); plain index arithmetic - no table of per-tap pointers, whose select would turn into an indexed
  // load from the struct and keep all of it in scratch memory
  const int t = s.t;
  const int tin = a.tmode == 0 ? t : (a.tmode == 1 ? (t >> 1) : (t == 0 ? 0 : 1 + ((t - 1) >> 1)));
  // group n_dt of the tile under its tap split (h4_split): `wsel` + tap block `dtw` = the group's weights, `fv` its source frame (an instance
  // frame index; negative: before the first frame).  Where taps share a group any of their (equal) frames may be read: the LAST one is
  const int sp = s.split, g1 = s.n_dt;
  const int dtw = sp == 0 ? g1 : (sp == 2 && g1 == 1 ? 2 : 0);
  const int fv = sp == 0 ? t + g1 - (a.kt - 1) : (sp == 1 ? t : (g1 == 1 ? t : (sp == 2 ? t - 1 : t - 2)));
  const bf16_t* w01 = a.tdup ? s.wp_base : s.wf_base;        // W0 + W1 lives in block 0 of both tables
  const bf16_t* wsel = sp == 1 ? s.wf_base + 9 * k.wtap_stride : (sp == 2 && g1 == 0 ? w01 : (sp == 3 && g1 == 1 ? s.wp_base + 9 * k.wtap_stride : s.wt_base));
  wsel = h4_pin64(wsel);
  const bool from_cache = a.kt > 1 && fv < 0 && s.cache_b != nullptr;
  const int fidx = a.kt > 1 ? (fv >= 0 ? fv : (from_cache ? a.kt - 1 + fv : 0)) : tin;
  const bf16_t* f = h4_pin64(from_cache ? s.cache_b + (long long)fidx * k.frame_elems : a.x + (s.f0 + fidx) * k.frame_elems);
  s.h_base = f + s.n_kc * BK;
  s.h_nrec = s.n_on ? k.frame_bytes - s.n_kc * ROWB : 0;      // off stream: zero-length descriptor -> harmless zero fill
  s.wg_nxt = wsel + (long long)(dtw * 9) * k.wtap_stride + s.n_kc * BK;
  s.nrec_b_nxt = s.n_on ? k.wtap_bytes - s.n_kc * ROWB : 0;
}
template <bool kUp, int kPart = 0>
__device__ __forceinline__ void h4_advance(H4State& s, const IgemmArgs& a, const H4Const& k) {   // cur <- nxt, nxt <- successor
  s.nrec_b_cur = s.nrec_b_nxt;
  if (++s.n_kc == k.kcn) {
    s.n_kc = 0;
    if (++s.n_dt == s.ndt) h4_open_tile<kUp, kPart>(s, a, k, s.n_tile + k.G);
  }
  h4_set_nxt(s, a, k);
}

// (Round 3 tried two other placements of a step's 2-4 LDS-DMA instructions - spread by sched_group_barrier: -4 %, the compiler also
// re-clusters the fragment reads; four sched_barrier-fenced quarters of {<= 1 DMA, 4 reads, 8 MFMAs}: +-0.3 % - profiles/r03_halo4x_dma.log.
// Unlike gemm4x's eight DMAs per step, two to four do not back up the CU's address path; the pinned order below stays.)
// kM16 (THE PRODUCT since the end of round 4; kM16 = false is the walk of rounds 1-4, kept in the TIMING build for tools/archive/halo_m16_ab.py): the same
// walk on v_mfma_f32_16x16x32_bf16 - 8 x 8 accumulator blocks of 16 couts x 16 pixels in the same 256 registers, one K-32 fragment per 16 rows
// (the same 16 ds_read_b128 per step), 64 MFMAs per step split by COUT half: the step's 8 activation fragments and the first 4 weight fragments
// are in registers when its barrier opens; the other 4 weight fragments are read under the first 32 MFMAs, the next step's 8 + 4 under the second
// 32 (two activation register sets, alternating per step).  Why: the chip is power-limited on real operands, and in this shape
// the matrix pipe alone sustains 2.0-2.1 PF on them against 1.88 PF (half the accumulator traffic per MAC; profiles/r04_mfma_shape_and_order.log).
// Results are BIT-IDENTICAL to the 32 x 32 x 16 walk (same K order inside the pipe: every form of the kernel and the full-size VAE,
// profiles/r04_halo_m16.log), 4.3-6.7 % faster at the headline shapes, -14.8 ms per clip.
// kPart (round 6): the launch's TILE GEOMETRY (H4Geo).  A 16 x 32 tile whose image ends within its first 16 columns (the last tile column when
// W % 32 is 1..16) spends half of its MFMAs on pixels that do not exist - 6.25 % of a 360-px-wide tile of the tiled VAE.  That column can be
// walked by a second launch in 32 x 16 tiles (kPart 1): the same 512-pixel register tile, LDS-DMA halo image (34 x 18 halo pixels), weight ring,
// step schedule and epilogue, only the block -> pixel map differs; a launch covers a rectangle of the image (IgemmArgs tx0 / ty0 / ntx / nty
// origin and tile counts, h_lim / w_lim its end).  Every output pixel accumulates in the same K order as in a 16 x 32 tile and every
// gn_partial slot sums the same pixels in the same order, so outputs AND statistics are bit-identical to the one-launch form.  (First form,
// commit 93c1943: the last column on HALF a register tile - 32 MFMAs per step against the same weight stream - cost 0.9 of a full tile: a
// step is then bound by the 64-B requests of the weight tiles.)
// kFill (TIMING build only, tools/gn_fusion_cost.py): the COST SIDE of a consumer-side GroupNorm + SiLU fusion measured inside the product walk.
// In steps 2..7 of every group each thread reads the two halo rounds of the next group that have just landed (its own 16-byte slots), runs the
// instruction mix of normalise + SiLU + pack on their 16 elements (unpack, fma, mul, exp, add, rcp, mul, cvt: 124 VALU, 2 of 8 transcendental,
// eight independent chains) and writes the ORIGINAL bytes back, so the conv's result is unchanged and the usual tests still hold - what changes
// is the issue stream: +5 instructions behind most MFMA pairs of six of the nine steps, 2 ds_read_b128 and 2 ds_write_b128 per step.
template <bool kUp, bool kTiming, bool kPipe = true, bool kSub = false, bool kM16 = false, int kPart = 0, bool kFill = false>
__global__ __launch_bounds__(256, 1) void conv3x3_halo4x_kernel(const IgemmArgs a) {
  using namespace halo8;
  using CFG = typename std::conditional<kSub, Halo4xSubCfg, Halo4xCfg>::type;
  using GEO = H4Geo<kPart>;
  static_assert(!(kUp && kSub), "the sub-pixel form runs on the plain halo geometry of the low-res grid");
  static_assert(kPart == 0 || (kM16 && !kUp && !kSub && !kTiming && kPipe), "tile geometry 1: the 16 x 16 x 32 walk of the plain conv only");
  static_assert(!kFill || (kM16 && !kUp && !kSub && !kTiming && kPart == 0), "kFill: the product walk of the plain conv only");
  // LDS image of the launch's tile geometry (these shadow the halo8 constants of the 16 x 32 tile)
  constexpr int HWID = GEO::HWID, A_BYTES = GEO::A_BYTES, RPW = GEO::RPW;
  constexpr int UHW = halo8::TW / 2 + 2, UHH = halo8::TH / 2 + 2;
  constexpr int NR = kUp ? (UHW * UHH * 5 + 255) / 256 : GEO::NR;   // halo rounds of 256 x 16 B: 4 / 12
  static_assert(NR * 4096 <= A_BYTES && NR <= H4_MAX_ROUNDS, "halo rounds overflow the halo buffer");
  static_assert(2 * A_BYTES + CFG::BR * B_BYTES <= 160 * 1024, "LDS image larger than a CU's 160 KB");
  constexpr int BR = CFG::BR, BAHEAD = CFG::BAHEAD, NT = CFG::NT;
  constexpr int B0 = 2 * A_BYTES;                               // LDS: halo buffer 0 | halo buffer 1 | weight ring
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __builtin_assume(wave >= 0 && wave < 4);
  const int hi = lane >> 5, l31 = lane & 31;
  const int q4 = lane >> 4, l15 = lane & 15;                  // kM16: fragment row / 16-byte K chunk of the 16 x 16 x 32 shape

  // PERSISTENT workgroups: block b walks tiles b, b + G, b + 2G, ... as ONE continuous K walk of (frame tap, channel
  // chunk) GROUPS of 9 spatial taps.  While group `cur` is multiplied, the halo of group `nxt` (the next group of this
  // tile or the first of the next tile) and the weights 3 steps ahead are in flight, so a tile boundary costs the
  // epilogue only - no prologue bubble, no workgroup launch, and the stores drain under the next tile's walk.
  // Everything a step needs from the staging side is per-group scalar state prepared at the group boundary; LDS slots
  // and buffers are compile-time functions of (tap, group parity), so a step carries ~10 scalar instructions.
  H4Const kc;
  kc.ntiles = a.T_out * a.nty * a.ntx * a.tiles_n;
  kc.G = (int)gridDim.x;
  kc.kcn = a.Cin / BK;
  kc.frame_elems = (long long)a.H_in * a.W_in * a.Cin;
  kc.frame_bytes = (int)(kc.frame_elems * 2);
  kc.wtap_bytes = (int)((long long)BN * a.Cin * 2);
  kc.wtap_stride = (long long)a.Cout_pad * a.Cin;
  kc.tid = tid;
  const int ntiles = kc.ntiles, G = kc.G;
  const int ngroups = a.kt * kc.kcn;
  const long long wtap_stride = kc.wtap_stride;

  unsigned voffB[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = j * 64 + (tid >> 2);
    const int c = (tid & 3) ^ ((row >> 2) & 3);
    voffB[j] = (unsigned)((row * a.Cin + c * 8) * 2);
  }

  // `st` is touched only by the h4_* functions; what the steps consume is copied into plain locals after every advance
  // (a struct captured by the step lambdas' closures is not promoted to registers)
  H4State st;
  st.n_oh0 = -1; st.n_ow0 = -1;
  unsigned voffA[H4_MAX_ROUNDS];
  const bf16_t *h_base = a.x, *wg_nxt = a.w;
  int h_nrec = 0, nrec_b_cur = 0, nrec_b_nxt = 0;
  auto publish = [&](const H4State& q) {
#pragma unroll
    for (int r = 0; r < NR; ++r) voffA[r] = q.voffA[r];
    h_base = q.h_base; wg_nxt = q.wg_nxt; h_nrec = q.h_nrec; nrec_b_cur = q.nrec_b_cur; nrec_b_nxt = q.nrec_b_nxt;
  };
  const bf16_t* b_wp = a.w;                                   // running pointer of the weight stream

  auto stage_halo_round = [&](auto rc, auto bufc) {
    constexpr int r = decltype(rc)::value, buf = decltype(bufc)::value;
    const auto srd = __builtin_amdgcn_make_buffer_rsrc((void*)h_base, (short)0, h_nrec, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_ptr_t)(smem + buf * A_BYTES + r * 4096 + wave * 1024), 16, voffA[r], 0, 0, 0);
  };
  auto stage_b = [&](auto slotc, int nrec) {                  // one weight tap -> ring slot; the stream pointer moves on
    constexpr int slot = decltype(slotc)::value;
    const auto srd = __builtin_amdgcn_make_buffer_rsrc((void*)b_wp, (short)0, nrec, 0x00020000);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_ptr_t)(smem + B0 + slot * B_BYTES + j * 4096 + wave * 1024), 16, voffB[j],
                                               0, 0, 0);
    b_wp += wtap_stride;
  };
  auto stage_b_half = [&](auto slotc, auto jc, int nrec) {     // kM16: one of stage_b's two instructions (the caller moves the stream pointer)
    constexpr int slot = decltype(slotc)::value, j = decltype(jc)::value;
    const auto srd = __builtin_amdgcn_make_buffer_rsrc((void*)b_wp, (short)0, nrec, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_ptr_t)(smem + B0 + slot * B_BYTES + j * 4096 + wave * 1024), 16, voffB[j], 0, 0, 0);
  };

  // weight fragment offsets: 4 cout tiles x 2 k-halves (slot 0), XOR-swizzled 64-B rows
  int boff[4][2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = i * 32 + l31;
      boff[i][kk] = B0 + row * ROWB + (((kk * 2 + hi) ^ ((row >> 2) & 3)) << 4);
    }
  // activation fragment bases (padded 80-B halo rows -> base + immediate for every tap and either buffer)
  const int abase0 = kM16 ? ((RPW * wave) * HWID + l15) * APITCH + q4 * 16 : ((4 * wave) * HWID + l31) * APITCH + hi * 16;
  int abaseU[3];
#pragma unroll
  for (int dw = 0; dw < 3; ++dw)
    abaseU[dw] = kM16 ? ((2 * wave) * UHW + 1 + ((l15 + dw - 1) >> 1)) * APITCH + q4 * 16 : ((2 * wave) * UHW + 1 + ((l31 + dw - 1) >> 1)) * APITCH + hi * 16;
  // kM16: weight rows 16 i + l15 share (row >> 2) & 3 = (l15 >> 2) & 3, so the staging-side XOR swizzle leaves ONE base + i * 16 rows
  const int bbase16 = B0 + l15 * ROWB + ((q4 ^ ((l15 >> 2) & 3)) << 4);

  // ---- prologue (once per workgroup): whole first halo + the first BAHEAD weight taps ----
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  h4_open_tile<kUp, kPart>(st, a, kc, (int)blockIdx.x);
  h4_set_nxt(st, a, kc);
  publish(st);
  {
    stage_halo_round(std::integral_constant<int, 0>{}, I0{}); stage_halo_round(std::integral_constant<int, 1>{}, I0{});
    stage_halo_round(std::integral_constant<int, 2>{}, I0{}); stage_halo_round(std::integral_constant<int, 3>{}, I0{});
    if (!kUp) {
      stage_halo_round(std::integral_constant<int, 4>{}, I0{}); stage_halo_round(std::integral_constant<int, 5>{}, I0{});
      stage_halo_round(std::integral_constant<int, 6>{}, I0{}); stage_halo_round(std::integral_constant<int, 7>{}, I0{});
      stage_halo_round(std::integral_constant<int, 8>{}, I0{}); stage_halo_round(std::integral_constant<int, 9>{}, I0{});
      stage_halo_round(std::integral_constant<int, 10>{}, I0{}); stage_halo_round(std::integral_constant<int, 11>{}, I0{});
    }
  }
  static_assert(kUp || NR == 12, "the prologue stages 12 halo rounds");
  b_wp = wg_nxt;
  stage_b(std::integral_constant<int, 0>{}, nrec_b_nxt);
  stage_b(std::integral_constant<int, 1>{}, nrec_b_nxt);
  stage_b(std::integral_constant<int, 2>{}, nrec_b_nxt);
  h4_advance<kUp, kPart>(st, a, kc);
  publish(st);                                                  // cur = group 0 of the first tile, nxt = its successor
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // fragment loaders: every address is a base VGPR + an immediate
  // (kSub: the four taps of a group sit at halo offset (py + a, px + b) - per-tile bases abaseT[tap]; abaseN0 = tap 0 of the NEXT group's tile)
  int abaseT[4] = {0, 0, 0, 0}, abaseN0 = 0;
  auto sub_bases = [&](int ph, int (&bt)[4]) {
    const int py = ph >> 1, px = ph & 1;
#pragma unroll
    for (int tp = 0; tp < 4; ++tp) bt[tp] = abase0 + ((py + (tp >> 1)) * HWID + px + (tp & 1)) * APITCH;
  };
  auto load_a = [&](auto tapc, auto kkc, auto bufc, bf16x8 (&xf)[4]) {
    constexpr int tap = decltype(tapc)::value, kk = decltype(kkc)::value, gb = decltype(bufc)::value * A_BYTES;
    constexpr int dh = tap / 3, dw = tap % 3;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      if (kSub) {
        xf[p] = *(const bf16x8*)(smem + abaseT[tap & 3] + gb + p * HWID * APITCH + kk * 32);
      } else if (kUp) {
        const int rowimm = ((p + dh + 1) >> 1) * UHW * APITCH;
        xf[p] = *(const bf16x8*)(smem + abaseU[dw] + gb + rowimm + kk * 32);
      } else {
        xf[p] = *(const bf16x8*)(smem + abase0 + gb + ((p + dh) * HWID + dw) * APITCH + kk * 32);
      }
    }
  };
  auto load_a_next0 = [&](auto bufc, bf16x8 (&xf)[4]) {           // kSub: k-half 0 of tap 0 of group nxt (possibly another tile, another phase)
    constexpr int gb = decltype(bufc)::value * A_BYTES;
#pragma unroll
    for (int p = 0; p < 4; ++p) xf[p] = *(const bf16x8*)(smem + abaseN0 + gb + p * HWID * APITCH);
  };
  auto load_b = [&](auto kkc, auto slotc, bf16x8 (&wf)[4]) {
    constexpr int kk = decltype(kkc)::value, slot = decltype(slotc)::value;
#pragma unroll
    for (int i = 0; i < 4; ++i) wf[i] = *(const bf16x8*)(smem + boff[i][kk] + slot * B_BYTES);
  };
  f32x16 acc[4][4];
  auto mma = [&](const bf16x8 (&wf)[4], const bf16x8 (&xf)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int p = 0; p < 4; ++p)
        acc[i][p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], xf[p], acc[i][p], 0, 0, 0);
  };

  // ---- kM16: the 16 x 16 x 32 walk.  Fragment idx = 2 p + j of a step: tile row p, columns 16 j .. 16 j + 15 (one K-32 fragment each) ----
  auto a16_addr = [&](auto tapc, auto bufc, int idx) -> int {
    constexpr int tap = decltype(tapc)::value, gb = decltype(bufc)::value * A_BYTES;
    constexpr int dh = tap / 3, dw = tap % 3;
    const int p = idx >> 1, j = idx & 1;
    if (kSub) return abaseT[tap & 3] + gb + (p * HWID + 16 * j) * APITCH;
    if (kUp) return abaseU[dw] + gb + (((p + dh + 1) >> 1) * UHW + 8 * j) * APITCH;
    return abase0 + gb + ((GEO::prow(idx) + dh) * HWID + dw + 16 * GEO::pcol(idx)) * APITCH;
  };
  auto a16_addr_next0 = [&](auto bufc, int idx) -> int {          // kSub: tap 0 of group nxt (possibly another tile, another phase)
    constexpr int gb = decltype(bufc)::value * A_BYTES;
    return abaseN0 + gb + ((idx >> 1) * HWID + 16 * (idx & 1)) * APITCH;
  };
  auto b16_addr = [&](auto slotc, int ib) -> int {                // cout block ib (16 rows) of a ring slot
    constexpr int slot = decltype(slotc)::value;
    return bbase16 + slot * B_BYTES + ib * 16 * ROWB;
  };
  // Accumulator block (cout block ib, pixel block pb) = quad 8 ib + pb; lane: pixel l15, couts 4 q4 .. 4 q4 + 3.  The MFMAs of this walk are
  // inline asm with the accumulator TIED and constrained to the AGPR file: as builtins on 64 separate quads the allocator rotates the quads
  // through VGPRs (v_accvgpr_read after most MFMAs, 300-580 B of scratch).  hipcc does not model the hazards of an asm MFMA (guide 5.7); the
  // walk needs none: A / B come from ds_read (lgkmcnt waits are register-based and are inserted), a quad is touched once per 64 MFMAs, and
  // between the zeroing / the epilogue's reads and the nearest MFMA lie a barrier and the group bookkeeping.
  f32x4 acc16[64];
  auto mfma16 = [&](int k, const bf16x8& w, const bf16x8& x) {
    asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc16[k]) : "v"(w), "v"(x));
  };

  // ---- kFill: see the kernel's header ----
  u32x4 fd[2];                                     // the two rounds' 16 bytes per lane (8 bf16 each)
  float ff[8], ft[8];
  unsigned fp4[4];
  float f_scale = 1.0f, f_shift = 0.0f;
  if (kFill) {                                     // opaque per-lane values (a real fusion holds the lane's channel scale / shift here)
    f_scale = 1.0f + (float)lane * 0.0009765625f; f_shift = (float)(lane & 7) * 0.03125f;
    asm volatile("" : "+v"(f_scale), "+v"(f_shift));
  }
  auto fill_op = [&](int q, int rnd) {             // op q (0..63) of round rnd: stage q >> 3 of element q & 7 (eight independent chains)
    const int e = q & 7, stg = q >> 3;
    const unsigned dw = rnd ? fd[1][e >> 1] : fd[0][e >> 1];
    switch (stg) {
      case 0: if (e & 1) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(ff[e]) : "v"(dw)); else asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(ff[e]) : "v"(dw)); break;
      case 1: asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(ff[e]) : "v"(f_scale), "v"(f_shift)); break;   // per-channel scale / shift live in registers
      case 2: asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ft[e]) : "s"(-1.4426950408889634f), "v"(ff[e])); break;
      case 3: asm volatile("v_exp_f32 %0, %0" : "+v"(ft[e])); break;
      case 4: asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(ft[e])); break;
      case 5: asm volatile("v_rcp_f32 %0, %0" : "+v"(ft[e])); break;
      case 6: asm volatile("v_mul_f32 %0, %0, %1" : "+v"(ff[e]) : "v"(ft[e])); break;
      default: if (!(e & 1)) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(fp4[e >> 1]) : "v"(ff[e]), "v"(ff[e + 1])); break;
    }
  };
  auto fill_slot = [&](int slot, int tap_, int nbuf) {   // slot 0..31 of step tap_; nbuf = the halo buffer of group nxt
    if (!kFill || tap_ < 2 || tap_ > 7) return;
    char* const base = smem + nbuf * A_BYTES + (2 * (tap_ - 2)) * 4096 + wave * 1024 + lane * 16;
    if (slot == 0) fd[0] = *(const u32x4*)base;
    if (slot == 1) fd[1] = *(const u32x4*)(base + 4096);
    if (slot >= 6 && slot < 32) {                  // 26 slots x 5 ops >= 2 rounds x 64 ops (the last two slots also carry the write-back)
      for (int j = 0; j < 5; ++j) {
        const int q = (slot - 6) * 5 + j;
        if (q < 128) fill_op(q & 63, q >> 6);
      }
    }
    if (slot == 30) { asm volatile("" :: "v"(fp4[0]), "v"(fp4[1]), "v"(fp4[2]), "v"(fp4[3])); *(u32x4*)base = fd[0]; }
    if (slot == 31) { asm volatile("" :: "v"(fp4[0]), "v"(fp4[1]), "v"(fp4[2]), "v"(fp4[3])); *(u32x4*)(base + 4096) = fd[1]; }
  };

  bf16x8 xa[4], wa[4], xb[4], wb[4];              // fragment sets: a = k-half 0, b = k-half 1
  bf16x8 xs[2][8], wl[4], wh[4];                  // kM16: activation sets (alternating per step), cout-low / cout-high weight fragments
  if (kSub) sub_bases(h4_decode(a, kc, (int)blockIdx.x).ph, abaseT);
  if (!kM16) {                                    // (kM16 reads its first fragments at the top of every tile: see the tile loop)
    load_a(I0{}, I0{}, I0{}, xa);
    load_b(I0{}, I0{}, wa);
  }

  // one K-step = one spatial tap of group cur (parity par); fragments of k-half 0 are already in xa/wa
  auto step = [&](auto tapc, auto parc) {
    constexpr int tap = decltype(tapc)::value, par = decltype(parc)::value;
    constexpr int NH = CFG::nh(tap, NR);                       // halo rounds staged in this step
    constexpr int PEND = CFG::inflight(tap, NR);               // (issue counts are tap-periodic: loads are unconditional)
    constexpr int stap = tap + BAHEAD;                         // the weight tap staged now (>= NT: of group nxt)
    constexpr int R0 = CFG::round0(tap);
    using Par = std::integral_constant<int, par>;
    using NPar = std::integral_constant<int, 1 - par>;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PEND) : "memory");   // drain all but the last BAHEAD-2 steps' loads,
    __builtin_amdgcn_s_barrier();                                  // then the step barrier (LDS hand-off point)
    __builtin_amdgcn_sched_barrier(0);
    // ---- from here to the end of the step: ONE basic block ----
    if (!kM16) {
      if (stap == NT) b_wp = wg_nxt;
      stage_b(std::integral_constant<int, (stap + NT * par) % BR>{}, stap < NT ? nrec_b_cur : nrec_b_nxt);
      if (NH >= 1) stage_halo_round(std::integral_constant<int, (NH >= 1 ? R0 : 0)>{}, NPar{});
      if (NH >= 2) stage_halo_round(std::integral_constant<int, (NH >= 2 ? R0 + 1 : 0)>{}, NPar{});
      if (NH >= 3) stage_halo_round(std::integral_constant<int, (NH >= 3 ? R0 + 2 : 0)>{}, NPar{});
      if (NH >= 4) stage_halo_round(std::integral_constant<int, (NH >= 4 ? R0 + 3 : 0)>{}, NPar{});
      if (NH >= 5) stage_halo_round(std::integral_constant<int, (NH >= 5 ? R0 + 4 : 0)>{}, NPar{});
      if (NH >= 6) stage_halo_round(std::integral_constant<int, (NH >= 6 ? R0 + 5 : 0)>{}, NPar{});
    }
    if (kM16) {
      // (the staging calls above are NOT made for kM16 - see the guard - they are issued from inside the MFMA stream below)
      constexpr int SP = (tap + NT * par) & 1;                     // this step's activation register set (steps alternate; trips are even)
      using SlotCur = std::integral_constant<int, (tap + NT * par) % BR>;
      using SlotNxt = std::integral_constant<int, (tap + 1 + NT * par) % BR>;
      // first half: cout blocks 0-3 (wl) x the 8 pixel blocks.  Behind every pair of MFMAs ONE other instruction, in a fixed order
      // (sched_barrier-fenced: the MFMA mask of sched_group_barrier does not see an asm MFMA): the 4 cout-high fragments this step's
      // second half needs, then the step's LDS-DMAs (2 weight halves + NH halo rounds)
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        mfma16(2 * g, wl[(2 * g) >> 3], xs[SP][(2 * g) & 7]);
        mfma16(2 * g + 1, wl[(2 * g + 1) >> 3], xs[SP][(2 * g + 1) & 7]);
        if (g < 4) wh[g] = *(const bf16x8*)(smem + b16_addr(SlotCur{}, 4 + g));
        if (g == 4) {
          if (stap == NT) b_wp = wg_nxt;
          stage_b_half(std::integral_constant<int, (stap + NT * par) % BR>{}, I0{}, stap < NT ? nrec_b_cur : nrec_b_nxt);
        }
        if (g == 5) {
          stage_b_half(std::integral_constant<int, (stap + NT * par) % BR>{}, I1{}, stap < NT ? nrec_b_cur : nrec_b_nxt);
          b_wp += wtap_stride;
        }
        if (g == 6 && NH >= 1) stage_halo_round(std::integral_constant<int, (NH >= 1 ? R0 : 0)>{}, NPar{});
        if (g == 7 && NH >= 2) stage_halo_round(std::integral_constant<int, (NH >= 2 ? R0 + 1 : 0)>{}, NPar{});
        if (g == 8 && NH >= 3) stage_halo_round(std::integral_constant<int, (NH >= 3 ? R0 + 2 : 0)>{}, NPar{});
        if (g == 9 && NH >= 4) stage_halo_round(std::integral_constant<int, (NH >= 4 ? R0 + 3 : 0)>{}, NPar{});
        if (g == 10 && NH >= 5) stage_halo_round(std::integral_constant<int, (NH >= 5 ? R0 + 4 : 0)>{}, NPar{});
        if (g == 11 && NH >= 6) stage_halo_round(std::integral_constant<int, (NH >= 6 ? R0 + 5 : 0)>{}, NPar{});
        fill_slot(g, tap, 1 - par);
        __builtin_amdgcn_sched_barrier(0);
      }
      // second half: cout blocks 4-7 (wh); behind the first 12 pairs the next step's 8 activation + 4 cout-low weight fragments
      // (last tap: tap 0 of group nxt - after a tile's last group that is the NEXT tile's first, which the tile loop reads again)
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        mfma16(32 + 2 * g, wh[(2 * g) >> 3], xs[SP][(2 * g) & 7]);
        mfma16(32 + 2 * g + 1, wh[(2 * g + 1) >> 3], xs[SP][(2 * g + 1) & 7]);
        if (g < 8) {
          int ad;
          if (tap < NT - 1) ad = a16_addr(std::integral_constant<int, (tap + 1) % NT>{}, Par{}, g);
          else if (kSub) ad = a16_addr_next0(NPar{}, g);
          else ad = a16_addr(I0{}, NPar{}, g);
          xs[SP ^ 1][g] = *(const bf16x8*)(smem + ad);
        } else if (g < 12) {
          wl[g - 8] = *(const bf16x8*)(smem + b16_addr(SlotNxt{}, g - 8));
        }
        fill_slot(16 + g, tap, 1 - par);
        __builtin_amdgcn_sched_barrier(0);
      }
      return;
    }
    load_a(tapc, I1{}, Par{}, xb);
    load_b(I1{}, std::integral_constant<int, (tap + NT * par) % BR>{}, wb);
    mma(wa, xa);
    // next step's k-half 0 (last tap: tap 0 of group nxt - after a tile's last group that is the NEXT tile's first)
    if (tap < NT - 1) load_a(std::integral_constant<int, (tap + 1) % NT>{}, I0{}, Par{}, xa);
    else if (kSub) load_a_next0(NPar{}, xa);
    else load_a(I0{}, I0{}, NPar{}, xa);
    load_b(I0{}, std::integral_constant<int, (tap + 1 + NT * par) % BR>{}, wa);
    mma(wb, xb);
    // pinned interleave: the staging work and one fragment read per MFMA gap
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                // 1 MFMA
      if (i < 2 + NH) {
        __builtin_amdgcn_sched_group_barrier(0x004, 4, 0);              // <= 4 SALU (descriptor, m0)
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);              // 1 VMEM read (LDS-DMA)
      }
      if (i >= CFG::DS0 && i < CFG::DS0 + 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 ds_read (k-half 1)
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (i < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // next step's k-half 0 fragments
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto group = [&](auto parc) {
    step(std::integral_constant<int, 0>{}, parc); step(std::integral_constant<int, 1>{}, parc);
    step(std::integral_constant<int, 2>{}, parc); step(std::integral_constant<int, 3>{}, parc);
    if (!kSub) {
      step(std::integral_constant<int, (kSub ? 0 : 4)>{}, parc); step(std::integral_constant<int, (kSub ? 0 : 5)>{}, parc);
      step(std::integral_constant<int, (kSub ? 0 : 6)>{}, parc); step(std::integral_constant<int, (kSub ? 0 : 7)>{}, parc);
      step(std::integral_constant<int, (kSub ? 0 : 8)>{}, parc);
    }
  };

  // epilogue-side lane role: 8 lanes x 8 channels cover 64 channels (128 B) of one pixel
  const int e_px = lane >> 3, e_ch = lane & 7;
  unsigned long long tm_walk = 0, tm_bar = 0, tm_body = 0, tm_drain = 0, tm_n = 0, tm0 = 0, tm1 = 0, tm2 = 0, tm3 = 0;
  const unsigned long long tm_start = kTiming ? __builtin_amdgcn_s_memtime() : 0;
  for (int tile = (int)blockIdx.x; tile < ntiles; tile += G) {
    if (kTiming) tm0 = __builtin_amdgcn_s_memtime();
    const H4Tile c = h4_decode<kPart>(a, kc, tile);
    f32x4 bias_r[2][2];                                       // this lane's 8 channels in each 64-channel half
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int cbh = c.n0 + h * 64 + e_ch * 8;
      bias_r[h][0] = bias_r[h][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (a.bias) { bias_r[h][0] = *(const f32x4*)(a.bias + cbh); bias_r[h][1] = *(const f32x4*)(a.bias + cbh + 4); }
    }
    if (kM16) {
#pragma unroll
      for (int k = 0; k < 64; ++k) acc16[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][p][r] = 0.f;
    }

    // a tile has an even number of groups (Cin % 64 == 0): two per trip, one of each halo-buffer parity - straight-line,
    // so there is no control-flow merge at which the register allocator would have to reconcile two step bodies
    const int ng_tile = h4_split_ndt(a, __builtin_amdgcn_readfirstlane(h4_split(a, c.t))) * kc.kcn;   // (2 or 1 temporal groups where taps read equal frames)
    int next_base0 = 0;
    if (kSub) {                                                 // halo offsets of this tile's phase; tap 0 of the next tile's
      sub_bases(c.ph, abaseT);
      int nb4[4];
      sub_bases(h4_decode(a, kc, tile + G).ph, nb4);
      next_base0 = nb4[0];
    }
    if (kM16) {
      // The tile's first fragments are read HERE, not carried over from the previous tile's last step like the 32 x 32 x 16 walk does: 12
      // fragments alive across the epilogue are 48 registers the epilogue does not have (one exposed LDS latency per tile of >= 72 steps)
#pragma unroll
      for (int idx = 0; idx < 8; ++idx) xs[0][idx] = *(const bf16x8*)(smem + a16_addr(I0{}, I0{}, idx));
#pragma unroll
      for (int ib = 0; ib < 4; ++ib) wl[ib] = *(const bf16x8*)(smem + b16_addr(I0{}, ib));
    }
    for (int g = 0; g < ng_tile; g += 2) {
      if (kSub) abaseN0 = abaseT[0];                            // group(I0)'s last step prefetches this tile's next group
      group(I0{});
      __builtin_amdgcn_sched_barrier(0);
      h4_advance<kUp, kPart>(st, a, kc);
      publish(st);
      __builtin_amdgcn_sched_barrier(0);
      if (kSub) abaseN0 = (g + 2 >= ng_tile) ? next_base0 : abaseT[0];   // ... the tile's last group the NEXT tile's first
      group(I1{});
      __builtin_amdgcn_sched_barrier(0);
      h4_advance<kUp, kPart>(st, a, kc);
      publish(st);
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---- epilogue: wave w owns tile rows 4w..4w+3 (128 pixels), all 128 channels.  The MFMA result layout (lane = pixel,
    // 4 consecutive channels per register quad) would give 8-B stores scattered over 32 rows per instruction; instead the
    // wave transposes through its own 12 KB slice of the halo buffer the last group just finished with (the other one
    // already holds the next tile's first halo), one tile row x 64 channels at a time in fp32 (bias and residual are
    // added before the single bf16 rounding), and stores 16 B per lane with 8 lanes covering a full 128-B line. ----
    if (kTiming) tm1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_barrier();                            // every wave is done reading that buffer
    if (kTiming) tm2 = __builtin_amdgcn_s_memtime();
    {
      constexpr int EROW = 272;                              // 64 fp32 per pixel + 16 pad
      char* const eslice = smem + A_BYTES + wave * 12288;      // (a tile's last group has parity 1)
      // lane byte offsets inside one output row segment (32 pixels from ow0, channels from n0): buffer addressing, so a
      // column past the image edge is an out-of-range offset (dropped by the hardware) and a row past it a zero-length
      // descriptor - the epilogue has no divergent control flow and no per-store 64-bit address arithmetic
      // A wave's 128 pixels leave in four groups p of 32 (pixel blocks 2 p and 2 p + 1), a group as 4 x 8 lanes-of-pixels `it x e_px`:
      //   16 x 32 tiles: group p = tile row 4 w + p, pixel 8 it + e_px of its 32 columns
      //   32 x 16 tiles: group p = tile rows 8 w + 2 p (it < 2) and + 1 (it >= 2), column 8 (it & 1) + e_px
      unsigned o_off[4], r_off[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int px = it * 8 + e_px;
        if (kPart == 1) {
          const int col = px & 15, row = px >> 4;
          const bool okw = c.ow0 + col < a.w_lim;
          o_off[it] = okw ? (unsigned)(((row * a.W_out + col) * (int)a.ldo + e_ch * 8) * 2) : 0x80000000u;
          r_off[it] = okw ? (unsigned)(((row * a.W_out + col) * (int)a.ldr + e_ch * 8) * 2) : 0x80000000u;
          continue;
        }
        // kSub: the tile is 16 x 32 LOW-RES pixels of one phase: output pixel (2 y + py, 2 x + px) - every other pixel of a 64-pixel row segment
        const bool okw = kSub ? c.ow0 + px < a.W_in : c.ow0 + px < a.w_lim;
        o_off[it] = okw ? (unsigned)(((kSub ? 2 * px : px) * (int)a.ldo + e_ch * 8) * 2) : 0x80000000u;
        r_off[it] = okw ? (unsigned)((px * (int)a.ldr + e_ch * 8) * 2) : 0x80000000u;
      }
      auto emit = [&](auto has_resid, auto has_gn) {
        constexpr bool kRes = decltype(has_resid)::value;
        constexpr bool kGn = decltype(has_gn)::value;
        // fused GroupNorm statistics: this lane's 8 channels per 64-channel half = two 4-channel quads; (sum, sum of
        // squares) of the bf16-ROUNDED stored values kept pairwise (even / odd channel) until the end of the tile
        typedef __attribute__((ext_vector_type(2))) float f32x2;
        f32x2 gs[2][2], gq[2][2];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int q2 = 0; q2 < 2; ++q2) { gs[h][q2] = f32x2{0.f, 0.f}; gq[h][q2] = f32x2{0.f, 0.f}; }
        auto wr = [&](int p, int h) {                          // accumulators of tile row p, channel half h -> the wave's slice (fp32)
          if (kM16) {                                            // four 16-cout blocks x two 16-pixel blocks, one register quad each
#pragma unroll
            for (int ib = 0; ib < 4; ++ib)
#pragma unroll
              for (int j = 0; j < 2; ++j) *(f32x4*)(eslice + (16 * j + l15) * EROW + (ib * 16 + 4 * q4) * 4) = acc16[(h * 4 + ib) * 8 + 2 * p + j];
            return;
          }
#pragma unroll
          for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
              f32x4 o;
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = acc[h * 2 + i2][p][gq * 4 + e];
              *(f32x4*)(eslice + l31 * EROW + (i2 * 32 + 8 * gq + 4 * hi) * 4) = o;
            }
        };
        // GroupNorm partial sums -> gn_partial.  A row of gn_partial belongs to (16 x 32 tile, wave slot 0..3) of the conv's tile grid whatever
        // the launch's geometry, and holds the sums over THAT slot's pixels (tile rows 4 s .. 4 s + 3) in the 16 x 32 form's order.  A wave of
        // a 32 x 16 tile holds rows 8 w .. 8 w + 7 = slots 2 (w & 1) and 2 (w & 1) + 1 of 16 x 32 tile (ty + (w >> 1), tx): its groups
        // p = 0, 1 are the first slot's rows, p = 2, 3 the second's, each lane meets a slot's pixels in the same order as the 16 x 32
        // form's lane does (the columns that form pads with zeros add exact zeros) - the statistics are bit-identical.  Slots of tiles this
        // launch does not own (past h_lim / the grid) are left alone.
        auto gn_flush = [&](int half) {                          // half: 0 / 1 = first / second 4-row slot of a 32 x 16 tile's wave
          const int cpg_log = a.cpg_log;                       // 2, 3 or 4 channels-per-group bits (Cout 128 / 256 / 512)
          const int gty = (c.oh0 >> 4) + (kPart == 1 ? wave >> 1 : 0), gtx = c.ow0 >> 5;
          const int slot = kPart == 1 ? 2 * (wave & 1) + half : wave;
          const long long tix = ((long long)c.t * a.tiles_h + gty) * a.tiles_w + gtx;
          const long long row = ((kSub ? tix * 4 + c.ph : tix) << 2) + slot;      // kSub: a row per phase (the four phases write the same channels)
          float* dst = a.gn_partial + row * 64;
          const bool own = kPart != 1 || (gty < a.tiles_h && gty * 16 < a.h_lim);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            float sv[2], qv[2];
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2) {
              sv[q2] = gs[h][q2][0] + gs[h][q2][1];
              qv[q2] = gq[h][q2][0] + gq[h][q2][1];
#pragma unroll
              for (int m = 8; m < 64; m <<= 1) { sv[q2] += __shfl_xor(sv[q2], m); qv[q2] += __shfl_xor(qv[q2], m); }
              gs[h][q2] = f32x2{0.f, 0.f}; gq[h][q2] = f32x2{0.f, 0.f};
            }
            const int ch0 = c.n0 + h * 64 + e_ch * 8;          // first of this lane's 8 channels
            if (cpg_log == 2) {                                // 4 channels per group: each quad is a group
              if (e_px == 0 && own) {
                dst[(ch0 >> 2) * 2] = sv[0]; dst[(ch0 >> 2) * 2 + 1] = qv[0];
                dst[((ch0 >> 2) + 1) * 2] = sv[1]; dst[((ch0 >> 2) + 1) * 2 + 1] = qv[1];
              }
            } else {
              float s8 = sv[0] + sv[1], q8 = qv[0] + qv[1];    // 8 channels per group: the lane's two quads
              if (cpg_log == 4) { s8 += __shfl_xor(s8, 1); q8 += __shfl_xor(q8, 1); }   // 16: two neighbouring lanes
              const bool writer = e_px == 0 && (cpg_log == 3 || (e_ch & 1) == 0);
              if (writer && own) { dst[(ch0 >> cpg_log) * 2] = s8; dst[(ch0 >> cpg_log) * 2 + 1] = q8; }
            }
          }
        };
        if (kPipe) wr(0, 0);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int oh = c.oh0 + RPW * wave + (kPart == 1 ? 2 * p : p);
          const long long pix0 = kSub ? ((long long)c.t * a.H_out + 2 * oh + (c.ph >> 1)) * a.W_out + 2 * c.ow0 + (c.ph & 1)
                                      : ((long long)c.t * a.H_out + oh) * a.W_out + c.ow0;
          const bool okh = kSub ? oh < a.H_in : oh < a.h_lim;
          // bytes a group's descriptor spans: one 32-pixel row segment (kSub: 64 pixels, every other one), 32 x 16 tiles: two rows of the image
          const int span_px = kSub ? 2 * halo8::TW : (kPart == 1 ? a.W_out + 16 : halo8::TW);
          const auto srd_o = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + pix0 * a.ldo + c.n0), (short)0,
                                                               okh ? (int)(span_px * a.ldo * 2) : 0, 0x00020000);
          const auto srd_r = __builtin_amdgcn_make_buffer_rsrc((void*)(kRes ? a.resid + pix0 * a.ldr + c.n0 : a.out), (short)0,
                                                               (kRes && okh) ? (int)((kPart == 1 ? a.W_out + 16 : halo8::TW) * a.ldr * 2) : 0, 0x00020000);
          // this group's lane offsets: a 32 x 16 tile drops the group's second row when it is past the launch's end
          unsigned oo[4], ro[4];
          const bool ok1 = oh + 1 < a.h_lim;
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            if (kPart == 1) { oo[it] = (it >= 2 && !ok1) ? 0x80000000u : o_off[it]; ro[it] = (it >= 2 && !ok1) ? 0x80000000u : r_off[it]; }
            else { oo[it] = o_off[it]; ro[it] = r_off[it]; }
          }
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            u32x4 rr[4];
            if (kRes) {                                      // residual first: its latency hides under the LDS pass
#pragma unroll
              for (int it = 0; it < 4; ++it) rr[it] = __builtin_amdgcn_raw_buffer_load_b128(srd_r, (int)ro[it], h * 128, 0);
            }
            if (!kPipe) wr(p, h);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private slice: no barrier needed
            f32x4 lo[4], hi4[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
              lo[it] = *(const f32x4*)(eslice + (it * 8 + e_px) * EROW + e_ch * 32);
              hi4[it] = *(const f32x4*)(eslice + (it * 8 + e_px) * EROW + e_ch * 32 + 16);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // the next block's accumulators go into the slice NOW: its reads above are complete, and the write latency then runs under
            // this block's bias / residual / statistics arithmetic and stores instead of in front of the next block's reads
            if (kPipe && (p < 3 || h < 1)) wr(h ? p + 1 : p, h ^ 1);
#pragma unroll
            for (int it = 0; it < 4; ++it) {
              f32x4 x0 = lo[it] + bias_r[h][0], x1 = hi4[it] + bias_r[h][1];
              if (kRes) {
                const u32x4 r = rr[it];
                x0 += f32x4{__uint_as_float(r[0] << 16), __uint_as_float(r[0] & 0xffff0000u), __uint_as_float(r[1] << 16),
                            __uint_as_float(r[1] & 0xffff0000u)};
                x1 += f32x4{__uint_as_float(r[2] << 16), __uint_as_float(r[2] & 0xffff0000u), __uint_as_float(r[3] << 16),
                            __uint_as_float(r[3] & 0xffff0000u)};
              }
              const u32x4 v = {pack_bf2(x0[0], x0[1]), pack_bf2(x0[2], x0[3]), pack_bf2(x1[0], x1[1]), pack_bf2(x1[2], x1[3])};
              if (kGn) {
                const bool valid = okh && oo[it] != 0x80000000u;        // pixels past the image edge are not part of the tensor
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const uint32_t w = valid ? v[e] : 0u;
                  const f32x2 f = {__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
                  gs[h][e >> 1] += f;
                  gq[h][e >> 1] += f * f;
                }
              }
              __builtin_amdgcn_raw_buffer_store_b128(v, srd_o, (int)oo[it], h * 128, 0);
              // store-data hazard (found on MI355X): the 16-B store reads its data VGPRs for the last lanes a few cycles after
              // issue; the next iteration's first VALU writes re-used them and its fp32 intermediates were stored instead
              __builtin_amdgcn_sched_barrier(0);
              asm volatile("s_nop 3" ::: "memory");
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          if (kGn && kPart == 1 && p == 1) gn_flush(0);          // rows 8 w .. 8 w + 3 are complete: the first of the wave's two slots
        }
        if (kGn) gn_flush(kPart == 1 ? 1 : 0);
      };
      if (a.gn_partial) {
        if (a.resid) emit(std::true_type{}, std::true_type{});
        else emit(std::false_type{}, std::true_type{});
      } else {
        if (a.resid) emit(std::true_type{}, std::false_type{});
        else emit(std::false_type{}, std::false_type{});
      }
    }
    // the counted-vmcnt scheme of the K walk restarts from an empty queue (stores count in vmcnt on gfx9; the loads of
    // the next tile's first steps were issued before them and have long landed)
    if (kTiming) tm3 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (kTiming) {
      const unsigned long long tm4 = __builtin_amdgcn_s_memtime();
      tm_walk += tm1 - tm0; tm_bar += tm2 - tm1; tm_body += tm3 - tm2; tm_drain += tm4 - tm3; ++tm_n;
    }
  }
  if (kTiming && a.gate && blockIdx.x == 100 && lane == 0) {       // TIMING build: `gate` is the host's debug buffer
    unsigned long long* o = (unsigned long long*)a.gate + wave * 8;
    o[0] = tm_walk; o[1] = tm_bar; o[2] = tm_body; o[3] = tm_drain; o[4] = tm_n; o[5] = (unsigned long long)ngroups * 9;
    o[6] = __builtin_amdgcn_s_memtime() - tm_start;
  }
}

// ------------------------------------------------------------------------------------------------
// Plain GEMM  out[M][N] = x[M][K] * w[N][K]^T  (DiT linears, 1x1x1 convs over a contiguous channels-last tensor): 256 x 256 workgroup
// tiles walked by PERSISTENT workgroups in an XCD-aware supertile order (g4_* below), operands staged by LDS-DMA, the LDS-transposed epilogue
// with buffer-addressed stores (bias, GELU(tanh), residual and AdaLN gate applied in fp32 on the read side).
//   gemm8p_kernel (the product): eight waves in ping-pong over two K-64 buffers of full 128-B rows - see its header.
//   gemm4x_kernel (round 2's kernel, TIMING build only, for the within-run A/B of tools/archive/gemm8p_ab.py): ONE wave per SIMD (512-register
//     budget), 4 waves = 2 x 2 wave tiles of 128 x 128, 4-stage K-32 ring staged 3 steps ahead with counted vmcnt, fragments
//     register-pipelined across the single per-step barrier, pinned MFMA / VMEM / DS interleave.
// (gemm4x's tile constants, the supertile walk g4_* and IgemmArgs live in igemm_args.h; gemm4x_kernel itself in gemm4x_timing.hip)

// ------------------------------------------------------------------------------------------------
// gemm8p: gemm4x's GEMM (same 256 x 256 tile, same persistent tile walk and supertile order, same arithmetic: results are bit-identical)
// with the two changes the measurements of round 3 asked for.
//  * FULL-LINE STAGING.  gemm4x's K-32 ring is filled by LDS-DMA instructions that fetch 16 rows x 64 B; a DMA-only kernel walking the same
//    tiles (tools/archive/stage_ab.py) stages at 11-14.5 TB/s that way - as long as the MFMA work itself takes - and 1.5-1.65x faster with 8 rows
//    x 128 B per instruction (64-B requests run into the L2 request rate: 11 requests per clock and XCD of 16; TA busy 78 %), while the
//    queue depth hardly matters (nothing in flight behind a 64 KB step: -9 %).  So the ring here is TWO K-64 buffers of 128-B rows
//    (XOR-swizzled like attention's K tile), each refilled in one go as soon as its last reader is through.
//  * PING-PONG.  Eight waves = two per SIMD; a wave owns 128 tokens x 64 channels (128 accumulator registers of its 256) and alternates
//        LOAD(q):  12 fragment reads of K-32 phase q (48 registers) [+ on even q its 8 LDS-DMAs: the K-64 step after this one]
//        MFMA(q):  16 MFMAs, nothing else in the stream
//    between workgroup barriers, the waves 4-7 (token rows 128-255) ONE barrier behind the waves 0-3: on every SIMD one wave computes
//    while its partner loads, and the pipe is handed over at each barrier with the last MFMA of one wave still executing.
//        slot 2q:   waves 0-3 LOAD(q)      waves 4-7 MFMA(q-1)
//        slot 2q+1: waves 0-3 MFMA(q)      waves 4-7 LOAD(q)
//    Buffer safety: K-64 step t = phases 2t, 2t+1 lives in buffer t & 1.  Its last reads are LOAD(2t+1): slots 4t+2 / 4t+3, closed by
//    barriers the readers pass after lgkmcnt(0).  Step t+2 is staged into it from LOAD(2t+2) = slots 4t+4 / 4t+5 on and first read in
//    slot 4t+8; every wave waits for its own DMAs (vmcnt(0): nothing younger is in flight) before the barrier that closes slot 4t+7 - the
//    end of MFMA(2t+3) for the waves 0-3, the end of LOAD(2t+3) for the waves 4-7.
namespace gemm8p {
constexpr int ROWB = 128;                                    // bytes of K per row and K-64 step
constexpr int OPB = 256 * ROWB;                              // one operand of one step: 32 KB
constexpr int XB = 0, WB = 2 * OPB;                          // x buffers at 0 / 32 KB (one address register + immediate reaches both), w at 64 / 96 KB
constexpr int EPI = 4 * OPB;                                 // epilogue staging: 8 waves x 32 rows x 128 B (XOR-swizzled)
constexpr int LDS_BYTES = EPI + 8 * 4096;                    // 163840
}  // namespace gemm8p
// kM16 (THE PRODUCT since the end of round 4; kM16 = false = round 3's phases, kept in the TIMING build for tools/archive/gemm_m16_ab.py, DOVE_GEMM_M16=0):
// the same phases on v_mfma_f32_16x16x32_bf16 - the MFMA shape the power-limited pipe sustains best on real operands and to which the dominant
// conv moved.  A wave's 128 tokens x 64 channels = 8 x 4 blocks of 16 x 16 (the same 128 accumulator registers), a phase =
// 12 K-32 fragment reads + 32 MFMAs.  Results are BIT-IDENTICAL to the 32 x 32 x 16 phases on every form the DiT uses (so the row tails on
// igemm_fast still sum like the main launch), 3-7 % faster at N = 18 226: qkv 1.29 -> 1.38 PF, ff1 1.26 -> 1.34 PF (profiles/r04_gemm_m16.log).
template <bool kAct, bool kGate, bool kTiming = false, bool kM16 = false>
__global__ __launch_bounds__(512, 2) void gemm8p_kernel(const IgemmArgs a, long long M) {
  using gemm4x::BM;
  using namespace gemm8p;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __builtin_assume(wave >= 0 && wave < 8);
  const int hi = lane >> 5, l31 = lane & 31;
  const int q4 = lane >> 4, l15 = lane & 15;                  // kM16: fragment row / 16-byte K chunk of the 16 x 16 x 32 shape
  const int grp = wave >> 2, wc = wave & 3;                   // token half = phase group, 64-channel slab

  G4Const kc;
  kc.K = a.Cin;
  kc.M = M;
  kc.tiles_n = a.tiles_n;
  kc.ntiles = (int)((M + BM - 1) / BM) * a.tiles_n;
  kc.G = (int)gridDim.x;
  kc.nk4 = a.Cin / 128;                                       // "chunk" of the g4_* walk = 128 of K = two K-64 steps
  const int ntiles = kc.ntiles, G = kc.G, nk4 = kc.nk4;

  // staging: wave w moves rows 32w .. 32w+31 of either operand, 8 rows x 8 chunks (one full 128-B line per row) per instruction; the source
  // chunk is XOR-swizzled with bits 1-3 of the row so that the fragment reads (32 rows, one chunk column) are bank-conflict free
  unsigned voff[4];
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const int row = (wave * 4 + jj) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    voff[jj] = (unsigned)((row * a.Cin + c * 8) * 2);
  }
  G4State st;
  const bf16_t *ca_base = a.x, *cw_base = a.w, *na_base = a.x, *nw_base = a.w;
  int ca_nrec = 0, cw_nrec = 0, c_soff = 0, na_nrec = 0, nw_nrec = 0, n_soff = 0;
  auto publish = [&](const G4State& q) {                      // cur <- nxt, nxt <- q
    ca_base = na_base; cw_base = nw_base; ca_nrec = na_nrec; cw_nrec = nw_nrec; c_soff = n_soff;
    na_base = q.a_base; nw_base = q.w_base; na_nrec = q.a_nrec; nw_nrec = q.w_nrec; n_soff = q.soff;
  };
  auto stage = [&](auto bufc, const bf16_t* ab, int anrec, const bf16_t* wb, int wnrec, int soff) {
    constexpr int buf = decltype(bufc)::value;
    const auto srd_a = __builtin_amdgcn_make_buffer_rsrc((void*)ab, (short)0, anrec, 0x00020000);
    const auto srd_w = __builtin_amdgcn_make_buffer_rsrc((void*)wb, (short)0, wnrec, 0x00020000);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_a, (lds_ptr_t)(smem + XB + buf * OPB + wave * 4096 + jj * 1024), 16, voff[jj], soff, 0, 0);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w, (lds_ptr_t)(smem + WB + buf * OPB + wave * 4096 + jj * 1024), 16, voff[jj], soff, 0, 0);
  };

  // fragment addresses: one per (K-32 half, K-16 slice) and operand; row blocks (+32 rows: same swizzle) and the buffer are immediates
  int aoff[2][2], boff[2][2];
  {
    const int ra = grp * 128 + l31, rb = wc * 64 + l31;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int ch = h * 4 + kk * 2 + hi;
        aoff[h][kk] = XB + ra * ROWB + ((ch ^ ((ra >> 1) & 7)) << 4);
        boff[h][kk] = WB + rb * ROWB + ((ch ^ ((rb >> 1) & 7)) << 4);
        asm volatile("" : "+v"(aoff[h][kk]), "+v"(boff[h][kk]));
      }
  }
  f32x16 acc[2][4];
  bf16x8 xf[4][2], wf[2][2];
  // kM16: one address per K-32 half and operand (row l15 of a 16-row block, chunk 4 h + q4 of the 128-B row; +16 rows keep the swizzle's
  // (row >> 1) & 7 term, so the blocks are immediates), token block pb = 16 rows, channel block ib = 16 rows of the wave's slab
  int aoff16[2], boff16[2];
  {
    const int ra = grp * 128 + l15, rb = wc * 64 + l15;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      aoff16[h] = XB + ra * ROWB + (((h * 4 + q4) ^ ((ra >> 1) & 7)) << 4);
      boff16[h] = WB + rb * ROWB + (((h * 4 + q4) ^ ((rb >> 1) & 7)) << 4);
      if (kM16) asm volatile("" : "+v"(aoff16[h]), "+v"(boff16[h]));
    }
  }
  f32x4 acc16[4][8];                                          // [channel block][token block]; lane: token l15, channels 4 q4 .. 4 q4 + 3
  bf16x8 x16[8], w16[4];

  // ---- prologue (once per workgroup): K-64 step 0 of the first tile ----
  g4_open_tile(st, a, kc, (int)blockIdx.x);
  publish(st);
  stage(std::integral_constant<int, 0>{}, na_base, na_nrec, nw_base, nw_nrec, n_soff);
  g4_advance(st, a, kc);
  publish(st);                                                // cur = chunk 0, nxt = its successor
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (grp) {                                                  // the second group starts one slot later
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }

  unsigned long long tm_lb = 0, tm_mb = 0;
  // phase u of a chunk: K-64 step u >> 1 (= its buffer), K-32 half u & 1
  auto step = [&](auto uc, bool hold) {
    constexpr int u = decltype(uc)::value;
    constexpr int buf = u >> 1, h = u & 1;
    // ---- LOAD ----
    if (kM16) {
#pragma unroll
      for (int pb = 0; pb < 8; ++pb) x16[pb] = *(const bf16x8*)(smem + aoff16[h] + buf * OPB + pb * (16 * ROWB));
#pragma unroll
      for (int ib = 0; ib < 4; ++ib) w16[ib] = *(const bf16x8*)(smem + boff16[h] + buf * OPB + ib * (16 * ROWB));
    } else {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int p = 0; p < 4; ++p) xf[p][kk] = *(const bf16x8*)(smem + aoff[h][kk] + buf * OPB + p * (32 * ROWB));
#pragma unroll
        for (int i = 0; i < 2; ++i) wf[i][kk] = *(const bf16x8*)(smem + boff[h][kk] + buf * OPB + i * (32 * ROWB));
      }
    }
    if (u == 0) stage(std::integral_constant<int, 1>{}, ca_base, ca_nrec, cw_base, cw_nrec, c_soff + ROWB);   // step 1 of this chunk
    if (u == 2) stage(std::integral_constant<int, 0>{}, na_base, na_nrec, nw_base, nw_nrec, n_soff);          // step 0 of the next one
    __builtin_amdgcn_sched_barrier(0);
    if (h == 1 && grp) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned long long tq = 0;
    if (kTiming) { tq = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
    __builtin_amdgcn_s_barrier();
    if (kTiming) { tm_lb += __builtin_amdgcn_s_memtime() - tq; }
    __builtin_amdgcn_sched_barrier(0);
    // ---- MFMA ----
    __builtin_amdgcn_s_setprio(1);                            // (with / without: 0.825 / 0.824 ms - kept for the hand-over at the barrier)
    if (kM16) {
#pragma unroll
      for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int pb = 0; pb < 8; ++pb) acc16[ib][pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w16[ib], x16[pb], acc16[ib][pb], 0, 0, 0);
    } else {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int p = 0; p < 4; ++p) acc[i][p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i][kk], xf[p][kk], acc[i][p], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    if (h == 1 && !grp) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (u == 3 && hold) return;                               // waves 4-7, last phase of a tile: epilogue first, then this barrier
    if (kTiming) { tq = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
    __builtin_amdgcn_s_barrier();
    if (kTiming) { tm_mb += __builtin_amdgcn_s_memtime() - tq; }
    __builtin_amdgcn_sched_barrier(0);
  };

  // epilogue-side lane role: 4 lanes x 8 columns cover the 32 columns (64 B) one accumulator block holds of an output row
  const int e_px = lane >> 2, e_ch = lane & 3;
  unsigned long long tm_walk = 0, tm_epi = 0, tm_n = 0, tm0 = 0, tm1 = 0;
  for (int tile = (int)blockIdx.x; tile < ntiles; tile += G) {
    if (kTiming) tm0 = __builtin_amdgcn_s_memtime();
    const G4Tile c = g4_decode(kc, tile);
    const int col0 = c.n0 + wc * 64;
    if (kM16) {
#pragma unroll
      for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int pb = 0; pb < 8; ++pb) acc16[ib][pb] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][p][r] = 0.f;
    }

    for (int kq = 0; kq < nk4; ++kq) {
      step(std::integral_constant<int, 0>{}, false);
      step(std::integral_constant<int, 1>{}, false);
      step(std::integral_constant<int, 2>{}, false);
      // Tile end.  In lockstep the two groups' epilogues would run one after the other (waves 0-3 during the others' last MFMA segment and
      // beyond, waves 4-7 during the first MFMA segment of the next tile: 2 E - 2 segments of idle matrix pipe per tile).  The waves 4-7
      // therefore run their epilogue BEFORE the barrier that closes their last MFMA segment: both epilogues overlap, E + 1 segment.
      step(std::integral_constant<int, 3>{}, grp && kq == nk4 - 1);
      __builtin_amdgcn_sched_barrier(0);
      g4_advance(st, a, kc);
      publish(st);
      __builtin_amdgcn_sched_barrier(0);
    }

    if (kTiming) tm1 = __builtin_amdgcn_s_memtime();
    // bias / gate rows of the lane's 16 channels: loaded HERE, not before the K walk - 48 registers the walk does not have (2 waves per SIMD);
    // they land under the first accumulator block's trip through LDS
    asm volatile("" ::: "memory");
    f32x4 bias_r[2][2], gate_r[2][2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int cb = col0 + h * 32 + e_ch * 8;
      bias_r[h][0] = bias_r[h][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (a.bias) { bias_r[h][0] = *(const f32x4*)(a.bias + cb); bias_r[h][1] = *(const f32x4*)(a.bias + cb + 4); }
      if (kGate) {
#pragma unroll
        for (int cls = 0; cls < 2; ++cls) {
          gate_r[cls][h][0] = *(const f32x4*)(a.gate + (long long)cls * a.Cout_pad + cb);
          gate_r[cls][h][1] = *(const f32x4*)(a.gate + (long long)cls * a.Cout_pad + cb + 4);
        }
      }
    }
    // ---- epilogue: the wave's 128 x 64 result, one 32-row x 32-column accumulator block at a time through its own 4 KB LDS slice
    // (fp32, XOR-swizzled 128-B rows), then 16-B stores with 4 lanes covering 64 contiguous bytes of a row ----
    {
      char* const eslice = smem + EPI + wave * 4096;
      unsigned o_off[2], r_off[2];
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int px = it * 16 + e_px;
        o_off[it] = (unsigned)((px * (int)a.ldo + e_ch * 8) * 2);
        r_off[it] = (unsigned)((px * (int)a.ldr + e_ch * 8) * 2);
      }
      auto emit = [&](auto has_resid) {
        constexpr bool kRes = decltype(has_resid)::value;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const long long row0 = (long long)c.m0 + grp * 128 + p * 32;
          const long long vl = M - row0;
          const int rows = vl >= 32 ? 32 : (vl > 0 ? (int)vl : 0);     // rows past M: offset >= num_records -> dropped
          const auto srd_o = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + row0 * a.ldo + col0), (short)0,
                                                               rows * (int)a.ldo * 2, 0x00020000);
          const auto srd_r = __builtin_amdgcn_make_buffer_rsrc((void*)(kRes ? a.resid + row0 * a.ldr + col0 : a.out), (short)0,
                                                               kRes ? rows * (int)a.ldr * 2 : 0, 0x00020000);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            u32x4 rr[2];
            if (kRes) {
#pragma unroll
              for (int it = 0; it < 2; ++it) rr[it] = __builtin_amdgcn_raw_buffer_load_b128(srd_r, (int)r_off[it], h * 64, 0);
            }
            if (kM16) {                                              // the block's 32 tokens x 32 channels = 2 x 2 quads of 16 x 16
#pragma unroll
              for (int dj = 0; dj < 2; ++dj)
#pragma unroll
                for (int di = 0; di < 2; ++di) {
                  const int row = dj * 16 + l15, ch = di * 4 + q4;
                  *(f32x4*)(eslice + row * 128 + ((ch ^ (row & 7)) << 4)) = acc16[2 * h + di][2 * p + dj];
                }
            } else {
#pragma unroll
              for (int gq = 0; gq < 4; ++gq) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = acc[h][p][gq * 4 + e];
                const int ch = 2 * gq + hi;                            // 16-B chunk of the 128-B row
                *(f32x4*)(eslice + l31 * 128 + ((ch ^ (l31 & 7)) << 4)) = o;
              }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // wave-private slice: no barrier needed
            f32x4 lo[2], hi4[2];
#pragma unroll
            for (int it = 0; it < 2; ++it) {
              const int px = it * 16 + e_px;
              lo[it] = *(const f32x4*)(eslice + px * 128 + (((2 * e_ch) ^ (px & 7)) << 4));
              hi4[it] = *(const f32x4*)(eslice + px * 128 + (((2 * e_ch + 1) ^ (px & 7)) << 4));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int it = 0; it < 2; ++it) {
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
                  const bool vid = row0 + it * 16 + e_px >= a.gate_split;   // row class: text rows first
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
              if (a.nt_out) __builtin_amdgcn_raw_buffer_store_b128(v, srd_o, (int)o_off[it], h * 64, 2);   // aux 2 = nt
              else __builtin_amdgcn_raw_buffer_store_b128(v, srd_o, (int)o_off[it], h * 64, 0);
              __builtin_amdgcn_sched_barrier(0);                       // store-data hazard: see gemm4x
              asm volatile("s_nop 3" ::: "memory");
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
      };
      if (a.resid) emit(std::true_type{});
      else emit(std::false_type{});
    }
    // the counted-vmcnt scheme of the K walk restarts from an empty queue (stores count in vmcnt on gfx9)
    __builtin_amdgcn_s_waitcnt(0x0F70);                       // vmcnt(0), as a builtin: the compiler's own wait tracking sees the queue empty
    asm volatile("" ::: "memory");
    if (kTiming) { const unsigned long long tm2 = __builtin_amdgcn_s_memtime(); tm_walk += tm1 - tm0; tm_epi += tm2 - tm1; ++tm_n; }
    if (grp) {                                                // the barrier held back above
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (kTiming && a.zero && blockIdx.x == 100 && lane == 0) {     // TIMING build: `zero` carries the host's debug buffer
    unsigned long long* o = (unsigned long long*)a.zero + wave * 8;
    o[0] = tm_walk; o[1] = tm_lb; o[2] = tm_mb; o[3] = tm_epi; o[4] = tm_n; o[5] = (unsigned long long)nk4 * 4;
  }
  if (!grp) {                                                 // balance the second group's extra first barrier
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  }
}

// ---- host side -------------------------------------------------------------------------------
static std::atomic<bf16_t*> g_zero_page[DOVE_MAX_DEVICES] = {};

static const bf16_t* zero_page() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= DOVE_MAX_DEVICES) return nullptr;
  bf16_t* z = g_zero_page[dev].load(std::memory_order_acquire);
  if (!z) {
    void* p = nullptr;
    if (hipMalloc(&p, 4096) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, 4096) != hipSuccess) return nullptr;
    bf16_t* expect = nullptr;
    if (g_zero_page[dev].compare_exchange_strong(expect, (bf16_t*)p, std::memory_order_acq_rel)) z = (bf16_t*)p;
    else { (void)hipFree(p); z = expect; }                      // another host thread was first
  }
  return z;
}

// ------------------------------------------------------------------------------------------------
// Dispatch.  ONE selection rule (used by the launch, by dove_conv_gn_partial_rows and by dove_conv_kernel_name); no
// environment switches.  Which shapes of the 33x720x1280 clip reach which kernel:
//   conv3x3_halo4x  3x3(x3) stride-1 convs with Cin % 64 == 0, Cout % 128 == 0, H, W >= 16 (every VAE resnet conv; without a conv cache and
//                   with dove_conv_desc.w_first the first two frames run 1 / 2 temporal groups) and the upsample-fused 3x3 convs of
//                   Upsample3D - in SUB-PIXEL form (<kSub>, dove_conv_desc.w_sub, low-res grid >= 16 x 32: 4 / 9 of the MACs), else with
//                   the upsample folded into the addressing (<kUp>)                             268 + 12 launches, 55 % of the step
//   gemm8p          plain GEMMs with M >= 4096, Cout % 256 == 0, Cin % 128 == 0, Cin >= 256 (DiT qkv / out / ff)    168 launches
//   smallk          pointwise convs with Cin_pad == 32: SpatialNorm conv_y||conv_b on the latent grid                    158 launches
//   igemm_fast      everything else without upsampling: stride-2 downsample convs, the (3,1,1) forms of encoder.conv_in / decoder.conv_out,
//                   decoder.conv_in (16 -> 512), the 1x1x1 shortcuts, encoder.conv_out, patch / text embedding, proj_out, the row
//                   tails of the gemm8p GEMMs, 3x3 convs of small clips (H or W < 16)
//   igemm (v1)      upsample-fused convs too small for the halo tile (test-sized clips), frames above the 31-bit buffer range     0
// (Round 1's 8-wave generations conv3x3_halo8 / gemm8 are gone: every production shape they still carried - 8 + ~10 launches
// per clip - runs on igemm_fast within noise of its old time.)
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------------------------
// smallk: out [M][ldo] = x [M][32] w^T [Cout][32] + bias for the 1x1x1 convs with (padded) Cin = 32 - CogVideoXSpatialNorm3D's
// conv_y || conv_b on the 16-channel latent, 158 launches per clip.  64 FLOP per output byte: HBM-write-bound, the MFMA tile
// pipelines (gemm8: 68 us per launch, 1.3 TB/s) only add latency.  A lane keeps 4 output channels x 32 K of weights as packed
// bf16 pairs (64 registers), a wave covers 256 consecutive channels of one row per pass: x is the same address for all
// lanes (staged per block in LDS, read as a broadcast), the products run on v_dot2c_f32_bf16, stores are 8 B per lane = 512
// contiguous bytes per wave.
// ------------------------------------------------------------------------------------------------------------------
namespace smallk { constexpr int ROWS = 32; }
__global__ __launch_bounds__(256) void smallk_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, const float* __restrict__ bias,
                                                     bf16_t* __restrict__ out, long long M, int cout_store, long long ldo) {
  typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
  __shared__ uint4 xs[4 * smallk::ROWS][4];                      // the block's 128 rows of x, one coalesced load
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long rb = (long long)blockIdx.x * 4 * smallk::ROWS;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = threadIdx.x + i * 256;
    const long long m = rb + (idx >> 2);
    xs[idx >> 2][idx & 3] = m < M ? *(const uint4*)(x + m * 32 + (idx & 3) * 8) : uint4{0u, 0u, 0u, 0u};
  }
  const int n0 = (blockIdx.y * 64 + lane) * 4;
  const bool live = n0 < cout_store;                             // Cout_store % 4 == 0: a lane's four channels are all in or all out
  uint32_t wr[4][16];
  float b[4] = {0.f, 0.f, 0.f, 0.f};
  if (live) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint4 v = *(const uint4*)(w + (long long)(n0 + j) * 32 + q * 8);
        wr[j][q * 4 + 0] = v.x; wr[j][q * 4 + 1] = v.y; wr[j][q * 4 + 2] = v.z; wr[j][q * 4 + 3] = v.w;
      }
    if (bias) { b[0] = bias[n0]; b[1] = bias[n0 + 1]; b[2] = bias[n0 + 2]; b[3] = bias[n0 + 3]; }
  }
  __syncthreads();
  if (!live) return;
  const long long r0 = rb + wave * smallk::ROWS;
#pragma unroll 4
  for (int r = 0; r < smallk::ROWS; ++r) {
    const long long m = r0 + r;
    if (m >= M) break;
    uint32_t xr[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint4 v = xs[wave * smallk::ROWS + r][q];           // same address in every lane: LDS broadcast
      xr[q * 4 + 0] = v.x; xr[q * 4 + 1] = v.y; xr[q * 4 + 2] = v.z; xr[q * 4 + 3] = v.w;
    }
    float acc[4] = {b[0], b[1], b[2], b[3]};
#pragma unroll
    for (int k = 0; k < 16; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[j] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2bf, wr[j][k]), __builtin_bit_cast(v2bf, xr[k]), acc[j], false);
    uint2 o;
    o.x = pack_bf2(acc[0], acc[1]);
    o.y = pack_bf2(acc[2], acc[3]);
    *(uint2*)(out + m * ldo + n0) = o;
  }
}

enum ConvKernel { K_IGEMM = 0, K_IGEMM_FAST, K_HALO4X, K_HALO4X_UP, K_GEMM8P, K_SMALLK, K_HALO4X_SUB };
static const char* const kKernelNames[] = {"igemm_kernel", "igemm_fast_kernel", "conv3x3_halo4x_kernel", "conv3x3_halo4x_kernel", "gemm8p_kernel",
                                           "smallk_kernel", "conv3x3_halo4x_kernel"};

static inline int desc_nb(const dove_conv_desc* d) { return d->nb > 1 ? d->nb : 1; }

static ConvKernel select_kernel(const dove_conv_desc* d) {
  const long long M = (long long)desc_nb(d) * d->t_out * d->h_out * d->w_out;
  const bool frame_fits = (long long)d->h_in * d->w_in * d->cin * 2 < (1ll << 31);
  const bool plain_gemm = d->kt == 1 && d->kh == 1 && d->kw == 1 && d->stride == 1 && d->up == 0 && d->tmode == 0 &&
                          d->t_in == d->t_out && d->h_in == d->h_out && d->w_in == d->w_out && d->cout_pad % 128 == 0 && M >= 4096 &&
                          (long long)512 * d->cin * 2 < (1ll << 31);
  const bool pointwise = d->kt == 1 && d->kh == 1 && d->kw == 1 && d->stride == 1 && d->up == 0 && d->tmode == 0 && d->t_in == d->t_out &&
                         d->h_in == d->h_out && d->w_in == d->w_out;
  if (pointwise && d->cin == 32 && d->act == 0 && !d->resid && !d->gate && d->cout_store >= 128) return K_SMALLK;
  if (plain_gemm) {
    if (d->cout_pad % 256 == 0 && d->cout_store % 256 == 0 && d->cin % 128 == 0 && d->cin >= 256 && (long long)256 * d->cin * 2 < (1ll << 31) &&
        d->ldo < (1 << 20) && d->ldr < (1 << 20) && (d->act == 0 || d->act == 1))
      return K_GEMM8P;
  }
  const bool conv3 = d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad_h == 1 && d->pad_w == 1 && d->act == 0 && !d->gate &&
                     d->cout_pad % 128 == 0 && frame_fits;
  if (conv3) {
    const bool same = d->up == 0 && d->tmode == 0 && d->h_out == d->h_in && d->w_out == d->w_in && d->w_out >= 16 && d->h_out >= 16;
    const bool ups = d->up == 1 && d->kt == 1 && d->h_out == 2 * d->h_in && d->w_out == 2 * d->w_in && d->h_out >= 16 && d->w_out >= 32;
    const bool h4 = d->cout_store % 128 == 0 && d->cin % 64 == 0 && d->ldo < (1 << 20) && (!d->resid || d->ldr < (1 << 20));
    if (same && h4) return K_HALO4X;
    // sub-pixel form (4 / 9 of the MACs) when the caller packed the phase-summed weights and the LOW-RES grid fills the 16 x 32 tile
    if (ups && h4 && d->w_sub && d->h_in >= 16 && d->w_in >= 32) return K_HALO4X_SUB;
    if (ups && h4) return K_HALO4X_UP;
  }
  const int BN = (d->cout_pad % 128 == 0) ? 128 : ((d->cout_pad % 64 == 0) ? 64 : 32);
  const bool fits = frame_fits && (long long)BN * d->cin * 2 < (1ll << 31);
  return (d->up == 0 && fits) ? K_IGEMM_FAST : K_IGEMM;
}

/* name of the kernel a call would dispatch to (reporting: bench.py's per-kernel roofline; tests pin the production shapes) */
static bool desc_size_ok(const dove_conv_desc* d, const char* who) {
  if (d && d->struct_size == sizeof(dove_conv_desc) && d->reserved == 0) return true;
  if (d) dove_set_error("%s: dove_conv_desc.struct_size is %u, this library (ABI %d) expects %zu - the caller's binding was written "
                        "against another include/dove_hip.h", who, d->struct_size, DOVE_ABI_VERSION, sizeof(dove_conv_desc));
  else dove_set_error("%s: null descriptor", who);
  return false;
}
extern "C" const char* dove_conv_kernel_name(const dove_conv_desc* d) { return desc_size_ok(d, "conv_kernel_name") ? kKernelNames[select_kernel(d)] : ""; }

template <int BN, int BK>
static int launch_igemm_fast(const IgemmArgs& a, unsigned grid, hipStream_t s) {
  constexpr int lds = 2 * (128 * BK * 2 + BN * BK * 2);
  static PerDeviceOnce attr_set;
  if (auto once_ = attr_set.guard()) (void)hipFuncSetAttribute((const void*)igemm_fast_kernel<BN, BK>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL((igemm_fast_kernel<BN, BK>), dim3(grid), dim3(256), lds, s, a);
  DOVE_CHECK_LAUNCH("dove_conv_igemm_bf16");
  return DOVE_OK;
}

static int cu_count() {
  static std::atomic<int> cus[DOVE_MAX_DEVICES] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  int n = (dev >= 0 && dev < DOVE_MAX_DEVICES) ? cus[dev].load(std::memory_order_relaxed) : 0;
  if (!n) {
    n = 256;
    (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    if (dev >= 0 && dev < DOVE_MAX_DEVICES) cus[dev].store(n, std::memory_order_relaxed);
  }
#ifdef DOVE_TIMING_BUILD
  {
    const char* e = getenv("DOVE_CU_LIMIT");                  // tools/archive/cumask_probe.py: persistent grids sized for a CU-masked stream
    if (e && atoi(e) > 0 && atoi(e) < n) return atoi(e);
  }
#endif
  return n;
}

extern "C" long long dove_conv_gn_partial_rows(const dove_conv_desc* d) {
  if (!desc_size_ok(d, "conv_gn_partial_rows")) return 0;
  const ConvKernel k = select_kernel(d);
  if (k != K_HALO4X && k != K_HALO4X_UP && k != K_HALO4X_SUB) return 0;
  if (d->cout_store != 128 && d->cout_store != 256 && d->cout_store != 512) return 0;   // 4 / 8 / 16 channels per group
  if (d->cout_store != d->cout_pad) return 0;
  if (k == K_HALO4X_SUB) {                                      // tiles of the LOW-RES grid x 4 phases x 4 waves
    const long long th = (d->h_in + halo8::TH - 1) / halo8::TH, tw = (d->w_in + halo8::TW - 1) / halo8::TW;
    return (long long)desc_nb(d) * d->t_out * th * tw * 16;
  }
  const long long th = (d->h_out + halo8::TH - 1) / halo8::TH, tw = (d->w_out + halo8::TW - 1) / halo8::TW;
  return (long long)desc_nb(d) * d->t_out * th * tw * 4;      // instance-major: instance b owns rows [b * rows / nb, (b + 1) * rows / nb)
}

// PARTIAL last tile column of a K_HALO4X conv (conv3x3_halo4x_kernel's kPart, H4Geo): when the image ends within the first half of its last
// tile column (W % 32 in 1..16), that column can leave the main launch and be walked in 32 x 16 tiles by a second launch:
//   main (16 x 32 tiles): columns [0, wm);   column (32 x 16 tiles): columns [wm, W)          outputs and gn_partial bit-identical
// It does so only when the two launches together take fewer ROUNDS of the persistent grid than the one launch: a launch costs whole rounds
// (a workgroup's tiles are a static share), so moving 4 % of the tiles into a launch of its own pays for a 68-round conv (the 240 x 360 tile
// class of the tiled VAE: 62 + 3 rounds, -4.3 %) and would cost a round for a 15-round one (profiles/r06_partial_tile_ab.log).
static bool halo4x_plan(const dove_conv_desc* d) {
  const int tiles_w = (d->w_out + halo8::TW - 1) / halo8::TW, tiles_h = (d->h_out + halo8::TH - 1) / halo8::TH;
  const int wrem = d->w_out % halo8::TW;
  if (wrem < 1 || wrem > halo8::TW / 2) return false;
  const int cus = cu_count();
  const long long per = (long long)desc_nb(d) * d->t_out * (d->cout_pad / 128);
  auto rounds = [&](long long tiles) { return (tiles * per + cus - 1) / cus; };
  return rounds((long long)tiles_h * (tiles_w - 1)) + rounds((d->h_out + 31) / 32) < rounds((long long)tiles_h * tiles_w);
}
/* partial-tile launches a conv call makes besides its main launch: 1 = the last tile column in 32 x 16 tiles (0 for every call that does not
 * dispatch to conv3x3_halo4x's plain form).  Reporting / tests. */
extern "C" int dove_conv_partial_launches(const dove_conv_desc* d) {
  if (!desc_size_ok(d, "conv_partial_launches") || select_kernel(d) != K_HALO4X) return 0;
  return halo4x_plan(d) ? 1 : 0;
}

static int conv_dispatch(const dove_conv_desc* d, void* stream, ConvKernel kern);
#ifdef DOVE_TIMING_BUILD
static bool halo_m16() {                                       // DOVE_HALO_M16=0 selects the predecessor walk; read per call (the A/B tool toggles it)
  const char* e = getenv("DOVE_HALO_M16");
  return !(e && atoi(e) == 0);
}
#endif

// GEMM tail: ntiles 256x256 tiles on G persistent workgroups take ceil(ntiles / G) rounds, and the last round of the DiT's
// N = 3072 GEMMs (864 tiles on 256 CUs) keeps 96 CUs busy for a whole tile time.  When the rows behind the last FULL round fit
// one resident wave of igemm_fast's 128x128 tiles (2 workgroups per CU), those rows go to igemm_fast instead: they then cost
// about 0.6 of a gemm8p round spread over every CU.  Rows are disjoint, both launches are ordered on the stream, the epilogue
// (bias / GELU / gated residual; gate_split shifted by the row offset) is the same code path as for any igemm_fast GEMM.
static bool gemm_tail_split(const dove_conv_desc* d, long long* rows_main) {
  const long long M = (long long)d->t_out * d->h_out * d->w_out;
  if (d->t_out != 1 || d->h_out != 1 || desc_nb(d) > 1) return false;               // token-major [1, 1, N] linears only, ONE instance (x, out
                                                                                      // and resid are split per row below: nb instances are not)
  const int cus = cu_count(), tiles_n = d->cout_pad / 256;
  const long long row_tiles = (M + gemm4x::BM - 1) / gemm4x::BM, ntiles = row_tiles * tiles_n;
  if (ntiles <= cus || ntiles % cus == 0) return false;
  const long long main_rt = (ntiles / cus) * cus / tiles_n;                           // row-tiles that fill whole rounds
  if (main_rt < 1 || main_rt >= row_tiles) return false;
  const long long tail_rows = M - main_rt * gemm4x::BM;
  const long long tail_blocks = ((tail_rows + 127) / 128) * (d->cout_pad / 128);
  if (tail_blocks > 2ll * cus) return false;                                          // more than one resident wave: no gain
  if ((long long)128 * d->cin * 2 >= (1ll << 31)) return false;
  *rows_main = main_rt * gemm4x::BM;
  return true;
}

extern "C" int dove_conv_igemm_bf16(const dove_conv_desc* d, void* stream) {
  if (!desc_size_ok(d, "conv_igemm")) return DOVE_EINVAL;
  DOVE_CHECK_ARG(d->x && d->w && d->out, "conv_igemm: null pointer");
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
  DOVE_CHECK_ARG(d->nb >= 0 && d->nb <= 4096 && (long long)desc_nb(d) * d->t_out < (1 << 20), "conv_igemm: bad instance count nb = %d", d->nb);
  DOVE_CHECK_ARG(d->cache_stride == 0 || (d->cache && d->cache_stride >= (long long)(d->kt - 1) * d->h_in * d->w_in * d->cin),
                 "conv_igemm: cache_stride (%lld) is smaller than one instance's cache", d->cache_stride);
  DOVE_CHECK_ARG(desc_nb(d) == 1 || !d->gate, "conv_igemm: nb > 1 is not combined with gate");
  DOVE_CHECK_ARG(d->reserved2 == 0 && d->tdup >= 0 && d->tdup <= 2, "conv_igemm: bad tdup %d", d->tdup);
  DOVE_CHECK_ARG(!d->tdup || (d->kt == 3 && d->w_pair), "conv_igemm: tdup declares frame pairs of a kt == 3 conv and needs w_pair");
  DOVE_CHECK_ARG(!d->tdup || d->cache || d->w_first, "conv_igemm: tdup without a conv cache needs w_first too (frames whose three taps read one frame)");
  DOVE_CHECK_ARG(d->tdup != 2 || !d->cache, "conv_igemm: tdup == 2 (first frame single) is the head of a clip: no conv cache");
  DOVE_CHECK_ARG(!d->gn_partial || dove_conv_gn_partial_rows(d) > 0,
                 "conv_igemm: gn_partial requested but this call does not dispatch to a kernel that fuses the statistics");
  const ConvKernel kern0 = select_kernel(d);
  DOVE_CHECK_ARG(!d->out_f32 || (kern0 == K_IGEMM_FAST && !d->resid && d->act == 0),
                 "conv_igemm: out_f32 is only implemented for plain convs that dispatch to igemm_fast_kernel");
  long long rows_main = 0;
  if (kern0 == K_GEMM8P && !DOVE_DBG_BUF && gemm_tail_split(d, &rows_main)) {
    dove_conv_desc m = *d, t = *d;
    m.w_in = m.w_out = (int)rows_main;
    const long long tail = (long long)d->w_out - rows_main;
    t.w_in = t.w_out = (int)tail;
    t.x = (const char*)d->x + rows_main * d->cin * 2;
    t.out = (char*)d->out + rows_main * d->ldo * 2;
    if (d->resid) t.resid = (const char*)d->resid + rows_main * d->ldr * 2;
    t.gate_split = d->gate_split > rows_main ? d->gate_split - rows_main : 0;
    const int rc = conv_dispatch(&m, stream, K_GEMM8P);
    return rc ? rc : conv_dispatch(&t, stream, K_IGEMM_FAST);
  }
  const int rc = conv_dispatch(d, stream, kern0);
#ifdef DOVE_TIMING_BUILD
  g_timing_debug_buf = nullptr;     // one call only
#endif
  return rc;
}

static int conv_dispatch(const dove_conv_desc* d, void* stream, ConvKernel kern) {
  const bf16_t* zp = zero_page();
  DOVE_CHECK_ARG(zp, "conv_igemm: could not allocate the zero page");
  IgemmArgs a;
  a.x = (const bf16_t*)d->x; a.cache = (const bf16_t*)d->cache; a.w = (const bf16_t*)d->w;
  a.bias = d->bias; a.resid = (const bf16_t*)d->resid; a.gate = d->gate; a.out = (bf16_t*)d->out; a.zero = zp;
  const int nb = desc_nb(d);
  a.T_out = nb * d->t_out; a.H_out = d->h_out; a.W_out = d->w_out;
  a.T_in = nb * d->t_in; a.H_in = d->h_in; a.W_in = d->w_in;
  a.seg_out = d->t_out; a.seg_in = d->t_in;
  a.cache_bs = d->cache_stride ? d->cache_stride : (long long)(d->kt - 1) * d->h_in * d->w_in * d->cin;
  a.Cin = d->cin; a.Cout_pad = d->cout_pad; a.Cout_st = d->cout_store;
  a.kt = d->kt; a.kh = d->kh; a.kw = d->kw; a.stride = d->stride; a.pad_h = d->pad_h; a.pad_w = d->pad_w;
  a.up = d->up; a.tmode = d->tmode; a.act = d->act; a.ldo = d->ldo; a.ldr = d->ldr; a.gate_split = d->gate_split;
  a.gn_partial = nullptr; a.cpg_log = 0;
  a.w_first = nullptr;
  a.w_pair = nullptr; a.tdup = 0;
  a.sub = 0;
  a.out_f32 = d->out_f32;
  a.nt_out = 0;
  a.debug = 0;
#ifdef DOVE_TIMING_BUILD
  {
    const char* e = getenv("DOVE_IGEMM_ABLATE");          // read per call: the A/B tools toggle it
    a.debug = e ? atoi(e) : 0;
  }
#endif
  hipStream_t s = (hipStream_t)stream;
  const long long M = (long long)nb * d->t_out * d->h_out * d->w_out;
  switch (kern) {
    case K_GEMM8P: {
      a.tiles_n = d->cout_pad / 256;
      // An output larger than the 256 MB Infinity Cache streams through it and evicts the operand panels the other workgroups are about to
      // re-read: written nontemporally, ff1 (18 226 x 12 288: 448 MB) runs 10-12 % faster back to back, qkv (336 MB) 1.5-2.4 %, the 112 MB
      // outputs of out / ff2 1.0-1.5 % slower.  Inside the operator the consumers then read from HBM what the cache might have kept: the
      // whole clip gains 0.1-0.25 % (within-run, three A/Bs; profiles/r03_gemm8p_nt.log) - kept because it never loses, not because it matters
      a.nt_out = (long long)M * d->ldo * 2 > (256ll << 20);
#ifdef DOVE_TIMING_BUILD
      if (a.debug & 8) a.nt_out = 1;                             // tools/archive/gemm8p_nt.py: force on / off
      if (a.debug & 16) a.nt_out = 0;
#endif
      const long long nt = ((M + gemm4x::BM - 1) / gemm4x::BM) * a.tiles_n;
      DOVE_CHECK_ARG(nt > 0 && nt < (1ll << 31), "conv_igemm: grid too large");
      DOVE_CHECK_ARG(!(d->gate && d->act), "conv_igemm: gate with activation is not a path of the reference");
      const int cus = cu_count();
      const unsigned grid4 = nt > cus ? (unsigned)cus : (unsigned)nt;
#ifdef DOVE_TIMING_BUILD
      {
        // the predecessor kernel, for within-run A/Bs: DOVE_GEMM8P=0 (read per call: tools/archive/gemm8p_ab.py toggles it), with DOVE_GEMM4X_SCHED=0
        // its round-2 DMA order (tools/archive/gemm4x_sched.py); a debug buffer selects the s_memtime instantiations (tools/gemm*_timing.py)
        const char* e8 = getenv("DOVE_GEMM8P");
        if (e8 && atoi(e8) == 0) {
          const char* es = getenv("DOVE_GEMM4X_SCHED");
          int variant = 0;
          if (DOVE_DBG_BUF && d->act == 0 && !d->gate) { a.zero = (const bf16_t*)DOVE_DBG_BUF; variant = 3; }
          else if (es && atoi(es) == 0 && !d->gate && d->act == 0) variant = 4;
          else if (d->gate) variant = 2;
          else if (d->act == 1) variant = 1;
          return launch_gemm4x_timing(a, M, grid4, variant, s);
        }
        {
          const char* em = getenv("DOVE_GEMM_M16");               // tools/archive/gemm_m16_ab.py: DOVE_GEMM_M16=0 = the 32 x 32 x 16 phases of round 3, read per call
          if (em && atoi(em) == 0 && !DOVE_DBG_BUF) {
            static PerDeviceOnce attr8m;
            if (auto once_ = attr8m.guard()) {
              (void)hipFuncSetAttribute((const void*)gemm8p_kernel<false, false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, gemm8p::LDS_BYTES);
              (void)hipFuncSetAttribute((const void*)gemm8p_kernel<true, false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, gemm8p::LDS_BYTES);
              (void)hipFuncSetAttribute((const void*)gemm8p_kernel<false, true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, gemm8p::LDS_BYTES);
            }
            if (d->gate) hipLaunchKernelGGL((gemm8p_kernel<false, true, false, false>), dim3(grid4), dim3(512), gemm8p::LDS_BYTES, s, a, M);
            else if (d->act == 1) hipLaunchKernelGGL((gemm8p_kernel<true, false, false, false>), dim3(grid4), dim3(512), gemm8p::LDS_BYTES, s, a, M);
            else hipLaunchKernelGGL((gemm8p_kernel<false, false, false, false>), dim3(grid4), dim3(512), gemm8p::LDS_BYTES, s, a, M);
            DOVE_CHECK_LAUNCH("dove_conv_igemm_bf16(gemm8p 32x32x16)");
            return DOVE_OK;
          }
        }
        if (DOVE_DBG_BUF) {
          static PerDeviceOnce attr8t;
          if (auto once_ = attr8t.guard()) {
            (void)hipFuncSetAttribute((const void*)gemm8p_kernel<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, gemm8p::LDS_BYTES);
            (void)hipFuncSetAttribute((const void*)gemm8p_kernel<true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, gemm8p::LDS_BYTES);
            (void)hipFuncSetAttribute((const void*)gemm8p_kernel<false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, gemm8p::LDS_BYTES);
          }
          a.zero = (const bf16_t*)DOVE_DBG_BUF;
          if (d->gate) hipLaunchKernelGGL((gemm8p_kernel<false, true, true>), dim3(grid4), dim3(512), gemm8p::LDS_BYTES, s, a, M);
          else if (d->act == 1) hipLaunchKernelGGL((gemm8p_kernel<true, false, true>), dim3(grid4), dim3(512), gemm8p::LDS_BYTES, s, a, M);
          else hipLaunchKernelGGL((gemm8p_kernel<false, false, true>), dim3(grid4), dim3(512), gemm8p::LDS_BYTES, s, a, M);
          DOVE_CHECK_LAUNCH("dove_conv_igemm_bf16(gemm8p timing)");
          return DOVE_OK;
        }
      }
#endif
      static PerDeviceOnce attr8;
      if (auto once_ = attr8.guard()) {
        (void)hipFuncSetAttribute((const void*)gemm8p_kernel<false, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, gemm8p::LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)gemm8p_kernel<true, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, gemm8p::LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)gemm8p_kernel<false, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, gemm8p::LDS_BYTES);
      }
      if (d->gate) hipLaunchKernelGGL((gemm8p_kernel<false, true, false, true>), dim3(grid4), dim3(512), gemm8p::LDS_BYTES, s, a, M);
      else if (d->act == 1) hipLaunchKernelGGL((gemm8p_kernel<true, false, false, true>), dim3(grid4), dim3(512), gemm8p::LDS_BYTES, s, a, M);
      else hipLaunchKernelGGL((gemm8p_kernel<false, false, false, true>), dim3(grid4), dim3(512), gemm8p::LDS_BYTES, s, a, M);
      DOVE_CHECK_LAUNCH("dove_conv_igemm_bf16(gemm8p)");
      return DOVE_OK;
    }
    case K_SMALLK: {
      dim3 grid((unsigned)((M + 4 * smallk::ROWS - 1) / (4 * smallk::ROWS)), (unsigned)((d->cout_store + 255) / 256));
      hipLaunchKernelGGL(smallk_kernel, grid, dim3(256), 0, s, a.x, a.w, a.bias, a.out, M, d->cout_store, d->ldo);
      DOVE_CHECK_LAUNCH("dove_conv_igemm_bf16(smallk)");
      return DOVE_OK;
    }
    case K_HALO4X_SUB: {
      a.gn_partial = d->gn_partial;
      a.cpg_log = d->cout_store == 128 ? 2 : (d->cout_store == 256 ? 3 : 4);
      a.sub = 1;
      a.w = (const bf16_t*)d->w_sub;                            // [4 phases][2x2 taps][cout_pad][cin]
      a.ty0 = a.tx0 = 0;
      a.tiles_w = (d->w_in + halo8::TW - 1) / halo8::TW;        // tiles of the LOW-RES grid, one per phase and cout tile
      a.tiles_h = (d->h_in + halo8::TH - 1) / halo8::TH;
      a.tiles_n = 4 * (d->cout_pad / 128);
      a.nty = a.tiles_h; a.ntx = a.tiles_w; a.h_lim = d->h_out; a.w_lim = d->w_out;
      const long long g4 = (long long)a.T_out * a.tiles_h * a.tiles_w * a.tiles_n;
      DOVE_CHECK_ARG(g4 > 0 && g4 < (1ll << 31), "conv_igemm: grid too large");
      static PerDeviceOnce attrs;
      if (auto once_ = attrs.guard()) (void)hipFuncSetAttribute((const void*)(conv3x3_halo4x_kernel<false, false, true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, Halo4xCfg::LDS_BYTES);
      const int cus = cu_count();
      const unsigned grid = g4 > cus ? (unsigned)cus : (unsigned)g4;
#ifdef DOVE_TIMING_BUILD
      if (!halo_m16()) {                                         // tools/archive/halo_m16_ab.py, DOVE_HALO_M16=0: the 32 x 32 x 16 walk of rounds 1-4
        static PerDeviceOnce attrm;
        if (auto once_ = attrm.guard()) (void)hipFuncSetAttribute((const void*)(conv3x3_halo4x_kernel<false, false, true, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, Halo4xCfg::LDS_BYTES);
        hipLaunchKernelGGL((conv3x3_halo4x_kernel<false, false, true, true, false>), dim3(grid), dim3(256), Halo4xCfg::LDS_BYTES, s, a);
      } else
#endif
      hipLaunchKernelGGL((conv3x3_halo4x_kernel<false, false, true, true, true>), dim3(grid), dim3(256), Halo4xCfg::LDS_BYTES, s, a);
      DOVE_CHECK_LAUNCH("dove_conv_igemm_bf16(halo4x sub-pixel)");
      return DOVE_OK;
    }
    case K_HALO4X:
    case K_HALO4X_UP: {
      a.gn_partial = d->gn_partial;
      a.w_first = (kern == K_HALO4X && d->kt == 3 && !d->cache) ? (const bf16_t*)d->w_first : nullptr;
      if (kern == K_HALO4X && d->kt == 3 && d->tdup) { a.tdup = d->tdup; a.w_pair = (const bf16_t*)d->w_pair; }   // (validated in dove_conv_igemm_bf16)
      a.cpg_log = d->cout_store == 128 ? 2 : (d->cout_store == 256 ? 3 : 4);
      a.tiles_w = (d->w_out + halo8::TW - 1) / halo8::TW;
      a.tiles_h = (d->h_out + halo8::TH - 1) / halo8::TH;
      a.tiles_n = d->cout_pad / 128;
      a.ty0 = a.tx0 = 0; a.nty = a.tiles_h; a.ntx = a.tiles_w; a.h_lim = d->h_out; a.w_lim = d->w_out;
      const long long g4 = (long long)a.T_out * a.tiles_h * a.tiles_w * a.tiles_n;
      DOVE_CHECK_ARG(g4 > 0 && g4 < (1ll << 31), "conv_igemm: grid too large");
      static PerDeviceOnce attr4;
      if (auto once_ = attr4.guard()) {
        (void)hipFuncSetAttribute((const void*)(conv3x3_halo4x_kernel<false, false, true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, Halo4xCfg::LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)(conv3x3_halo4x_kernel<false, false, true, false, true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, H4Geo<1>::LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)(conv3x3_halo4x_kernel<true, false, true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, Halo4xCfg::LDS_BYTES);
#ifdef DOVE_TIMING_BUILD
        (void)hipFuncSetAttribute((const void*)conv3x3_halo4x_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, Halo4xCfg::LDS_BYTES);
#endif
      }
      const int cus = cu_count();                              // persistent: one workgroup per CU walks its share of the tiles
      const unsigned grid = g4 > cus ? (unsigned)cus : (unsigned)g4;
#ifdef DOVE_TIMING_BUILD
      if (DOVE_DBG_BUF && kern == K_HALO4X) {                   // tools/archive/halo4x_timing.py
        a.gate = (const float*)DOVE_DBG_BUF;
        static PerDeviceOnce attrt;
        if (auto once_ = attrt.guard())
          (void)hipFuncSetAttribute((const void*)(conv3x3_halo4x_kernel<false, true, true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, Halo4xCfg::LDS_BYTES);
        if (halo_m16()) hipLaunchKernelGGL((conv3x3_halo4x_kernel<false, true, true, false, true>), dim3(grid), dim3(256), Halo4xCfg::LDS_BYTES, s, a);
        else hipLaunchKernelGGL((conv3x3_halo4x_kernel<false, true>), dim3(grid), dim3(256), Halo4xCfg::LDS_BYTES, s, a);
      } else
#endif
#ifdef DOVE_TIMING_BUILD
      if (!halo_m16()) {                                         // tools/archive/halo_m16_ab.py, DOVE_HALO_M16=0: the 32 x 32 x 16 walk of rounds 1-4
        static PerDeviceOnce attrm;
        if (auto once_ = attrm.guard()) {
          (void)hipFuncSetAttribute((const void*)(conv3x3_halo4x_kernel<false, false, true, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, Halo4xCfg::LDS_BYTES);
          (void)hipFuncSetAttribute((const void*)(conv3x3_halo4x_kernel<true, false, true, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, Halo4xCfg::LDS_BYTES);
        }
        if (kern == K_HALO4X_UP) hipLaunchKernelGGL((conv3x3_halo4x_kernel<true, false, true, false, false>), dim3(grid), dim3(256), Halo4xCfg::LDS_BYTES, s, a);
        else hipLaunchKernelGGL((conv3x3_halo4x_kernel<false, false, true, false, false>), dim3(grid), dim3(256), Halo4xCfg::LDS_BYTES, s, a);
      } else
#endif
#ifdef DOVE_TIMING_BUILD
      if ((a.debug & 64) && kern == K_HALO4X) {                  // tools/e2e_env_ab.py DOVE_IGEMM_ABLATE 64 0: epilogue without the early slice write
        static PerDeviceOnce attrp;
        if (auto once_ = attrp.guard()) (void)hipFuncSetAttribute((const void*)(conv3x3_halo4x_kernel<false, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, Halo4xCfg::LDS_BYTES);
        hipLaunchKernelGGL((conv3x3_halo4x_kernel<false, false, false>), dim3(grid), dim3(256), Halo4xCfg::LDS_BYTES, s, a);
      } else
#endif
      if (kern == K_HALO4X_UP) hipLaunchKernelGGL((conv3x3_halo4x_kernel<true, false, true, false, true>), dim3(grid), dim3(256), Halo4xCfg::LDS_BYTES, s, a);
      else {
        // a partial last tile column goes to a launch of its own in 32 x 16 tiles when that saves a round: see halo4x_plan
        const bool wpart = halo4x_plan(d);
        const int wm = wpart ? (a.tiles_w - 1) * halo8::TW : d->w_out, cols_main = wpart ? a.tiles_w - 1 : a.tiles_w;
        auto launch = [&](int part, int nty, int tx0, int ntx, int w_lim) {
          if (nty <= 0 || ntx <= 0) return;
          IgemmArgs b = a;
          b.ty0 = 0; b.nty = nty; b.tx0 = tx0; b.ntx = ntx; b.h_lim = d->h_out; b.w_lim = w_lim;
          const long long g = (long long)b.T_out * nty * ntx * b.tiles_n;
          const unsigned gr = g > cus ? (unsigned)cus : (unsigned)g;
          if (part == 0) hipLaunchKernelGGL((conv3x3_halo4x_kernel<false, false, true, false, true, 0>), dim3(gr), dim3(256), H4Geo<0>::LDS_BYTES, s, b);
          else hipLaunchKernelGGL((conv3x3_halo4x_kernel<false, false, true, false, true, 1>), dim3(gr), dim3(256), H4Geo<1>::LDS_BYTES, s, b);
        };
#ifdef DOVE_TIMING_BUILD
        if (const char* e = getenv("DOVE_HALO_FILL"); e && atoi(e) == 1) {     // tools/gn_fusion_cost.py: the product walk + the fusion's instruction mix
          static PerDeviceOnce attrf;
          if (auto once_ = attrf.guard())
            (void)hipFuncSetAttribute((const void*)(conv3x3_halo4x_kernel<false, false, true, false, true, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, H4Geo<0>::LDS_BYTES);
          IgemmArgs b = a;
          b.ty0 = 0; b.nty = a.tiles_h; b.tx0 = 0; b.ntx = a.tiles_w; b.h_lim = d->h_out; b.w_lim = d->w_out;
          hipLaunchKernelGGL((conv3x3_halo4x_kernel<false, false, true, false, true, 0, true>), dim3(grid), dim3(256), H4Geo<0>::LDS_BYTES, s, b);
          DOVE_CHECK_LAUNCH("dove_conv_igemm_bf16(halo4x, kFill)");
          return DOVE_OK;
        }
#endif
        launch(0, a.tiles_h, 0, cols_main, wm);
        if (wpart) launch(1, (d->h_out + 31) / 32, cols_main, 1, d->w_out);
      }
      DOVE_CHECK_LAUNCH("dove_conv_igemm_bf16(halo4x)");
      return DOVE_OK;
    }
    default: break;
  }
  // generic tiles: 8x16 pixels for images, 1x128 for token-major (H == 1) tensors
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
  const bool fast = kern == K_IGEMM_FAST;
  const bool bk64 = (d->cin % 64 == 0);
  if (!fast) return launch_igemm_legacy(a, (unsigned)grid, BN, bk64, s);        // igemm_legacy.hip
  if (BN == 128) return bk64 ? launch_igemm_fast<128, 64>(a, (unsigned)grid, s) : launch_igemm_fast<128, 32>(a, (unsigned)grid, s);
  if (BN == 64) return bk64 ? launch_igemm_fast<64, 64>(a, (unsigned)grid, s) : launch_igemm_fast<64, 32>(a, (unsigned)grid, s);
  return bk64 ? launch_igemm_fast<32, 64>(a, (unsigned)grid, s) : launch_igemm_fast<32, 32>(a, (unsigned)grid, s);
}
