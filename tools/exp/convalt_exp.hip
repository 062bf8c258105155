// EXPERIMENT (tools/archive/convalt.py): two structural alternatives for the dominant kernel (conv3x3_halo4x), measured on a model of its K walk:
// one wave per SIMD (256 threads, 1 workgroup per CU, 512 registers per lane), 256 accumulator registers, every operand fragment read
// from LDS with ds_read_b128, one barrier per 32-MFMA step, no global traffic (the LDS-DMA staging of the real kernel is not modelled: the
// model is an UPPER bound for both variants).
//
//  MODE 0  baseline: halo4x's register tile - 4 x 4 accumulator blocks of 32 x 32, 8 fragment reads per 16 MFMAs (1 read : 2 MFMA... per
//          step 16 reads : 32 MFMAs)
//  MODE 1   the same walk with GroupNorm-apply + SiLU done IN LDS on the staged halo slab: in 6 of the 9 steps of a
//          group every lane rewrites the two 16-byte slots it staged (ds_read_b128, a / b of its 8 channels from a table in LDS, 8 x
//          (x a + b), SiLU = mul, v_exp, add, v_rcp, mul, border mask, 4 v_cvt_pk, ds_write_b128), interleaved one slice per MFMA gap
//  MODE 2   Winograd F(2x2, 3x3): 16 transform-domain positions are 16 INDEPENDENT GEMMs, so the 256 accumulator
//          registers hold 16 blocks that share NO operand: every MFMA needs its own A and its own B fragment (2 reads : 1 MFMA)
//  MODE 3  Winograd F(2, 3) along W only (4 positions, 1.5 x fewer MACs): 4 positions x (2 x 2 blocks): 1 read : 1 MFMA
//  MODE 4  baseline tile with NO fragment reads in the loop (registers only): the matrix pipe's own rate in this harness
//  MODE 5  the baseline walk in the OTHER MFMA shape: v_mfma_f32_16x16x32_bf16, 8 x 8 accumulator blocks of 16 x 16 (the same 128 x 128 tile,
//          the same 256 accumulator registers, the same 16 fragment reads per step - one K-32 fragment of 16 rows each - and 64 MFMAs);
//          tools/archive/mfma_storm.py order: the pipe alone holds 2.0-2.1 PF in this shape on N(0,1) data against 1.8-1.9 PF for 32 x 32 x 16
//  MODE 6  mode 5 with the operands held in registers (no LDS reads)
//  MODE 7  mode 5 HAND-PIPELINED like the product kernel: fragments register-double-buffered across the step barrier (this step's x and the
//          first half of its w are already in registers when the barrier opens; the second half of w and the NEXT step's x / first w half
//          are read under this step's MFMAs), one ds_read pinned behind every 4 MFMAs with sched_group_barrier
//  MODE 8  mode 0 pipelined the same way (k-half 1 and the next step's k-half 0 under the MFMAs: the product kernel's own scheme), so that
//          7 vs 8 compares the two shapes at equal scheduling effort
//  MODE 10 the PRODUCT's 16x16x32 step as it is written in igemm.hip (asm MFMAs with the accumulator quad tied in the AGPR file, one other
//          instruction behind every MFMA pair, sched_barrier-fenced), without its LDS-DMAs
//  MODE 9  mode 10 + the product walk's LDS-DMA traffic: per step two 1-KB weight instructions per wave (8 KB per workgroup, ring slot three
//          steps ahead) and in six of nine steps two halo instructions per wave (48 KB per group into the other halo buffer), sourced from a
//          128 KB global buffer (L2-resident, like the weights), drained with a counted vmcnt before the step barrier - what the global ->
//          LDS path costs on top of the register-pipelined walk (10 vs 9), with no epilogue and no HBM stream
// Per mode the host reports ns per 32-MFMA step and the equivalent dense rate; speed-up of a variant = (MAC reduction) x (rate ratio).
#include "../../dove_amd/csrc/common.h"

constexpr int LDS_BYTES = 147456;          // as halo4x: forces one workgroup per CU

__device__ __forceinline__ bf16x8 lds128(const char* smem, int off) { return *(const bf16x8*)(smem + off); }

// one slice of the GN-apply rewrite: element e (0..7) of the lane's 16-byte slot
struct GnSlot { uint4 raw; f32x4 a0, a1, b0, b1; float f[8]; };

template <int MODE>
__global__ __launch_bounds__(256, 1) void convalt_kernel(const bf16_t* __restrict__ init, float* __restrict__ out, int groups) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // fill LDS with N(0,1)-like bf16 (the matrix pipes throttle with operand toggling: constants would flatter every variant)
  for (int i = tid; i < LDS_BYTES / 16; i += 256) ((uint4*)smem)[i] = ((const uint4*)init)[(blockIdx.x * 131 + i) % 8192];
  __syncthreads();
  f32x16 acc[16];
#pragma unroll
  for (int k = 0; k < 16; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  const int l31 = lane & 31, hi = lane >> 5;
  const int abase = ((4 * wave) * 34 + l31) * 80 + hi * 16;           // halo rows of 80 B, like the product kernel
  const int bbase = 98304 + l31 * 64 + hi * 16;                        // weight ring behind the two halo buffers
  const int slot0 = (wave * 64 + lane) * 16;                           // the lane's own staged slot of a round
  const int tab = 147456 - 4096 + (lane & 3) * 64;                     // a[8], b[8] of the lane's channel chunk
  float sink = 0.f;

  if (MODE == 7) {
    typedef __attribute__((ext_vector_type(4))) float f32x4_t;
    f32x4_t* acc4 = (f32x4_t*)acc;
    const int l15 = lane & 15, q4 = lane >> 4;
    const int abase5 = ((4 * wave) * 34 + l15) * 80 + q4 * 16, bbase5 = 98304 + l15 * 64 + q4 * 16;
    bf16x8 xs[2][8], wl[2][4], wh[4];
    auto xaddr = [&](int tap, int j) { return abase5 + (tap / 3) * 34 * 80 + (tap % 3) * 80 + (j >> 1) * 34 * 80 + (j & 1) * 16 * 80; };
    auto waddr = [&](int tap, int j) { return bbase5 + (tap % 6) * 8192 + j * 1024; };
#pragma unroll
    for (int j = 0; j < 8; ++j) xs[0][j] = lds128(smem, xaddr(0, j));
#pragma unroll
    for (int j = 0; j < 4; ++j) wl[0][j] = lds128(smem, waddr(0, j));
    for (int g = 0; g < groups; g += 2) {
#pragma unroll
      for (int st = 0; st < 18; ++st) {
        const int tap = st % 9, ntap = (st + 1) % 9, cur = st & 1, nxt = cur ^ 1;
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) wh[j] = lds128(smem, waddr(tap, 4 + j));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int p = 0; p < 8; ++p) acc4[i * 8 + p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[cur][i], xs[cur][p], acc4[i * 8 + p], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) xs[nxt][j] = lds128(smem, xaddr(ntap, j));
#pragma unroll
        for (int j = 0; j < 4; ++j) wl[nxt][j] = lds128(smem, waddr(ntap, j));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int p = 0; p < 8; ++p) acc4[(4 + i) * 8 + p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[i], xs[cur][p], acc4[(4 + i) * 8 + p], 0, 0, 0);
        // pinned interleave: 4 MFMAs, then one fragment read; wh (4 reads) must land before the second half - they go first
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  if (MODE == 9 || MODE == 10) {
    // the PRODUCT's step (conv3x3_halo4x_kernel<kM16>): asm MFMAs with the accumulator quad tied in the AGPR file, one other instruction behind
    // every MFMA pair in a fixed, sched_barrier-fenced order - cout-high fragments, then (MODE 9) the step's LDS-DMAs, then in the second
    // half the next step's 8 + 4 fragments
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef __attribute__((ext_vector_type(4))) float f32x4_t;
    const auto srd = __builtin_amdgcn_make_buffer_rsrc((void*)init, (short)0, 8192 * 16, 0x00020000);
    const unsigned dvoff = (unsigned)(lane * 16);
    const int l15 = lane & 15, q4 = lane >> 4;
    const int abase5 = ((4 * wave) * 34 + l15) * 80 + q4 * 16, bbase5 = 98304 + l15 * 64 + q4 * 16;
    f32x4_t q[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) q[k] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    bf16x8 xs[2][8], wl[4], wh[4];
    auto xaddr = [&](int tap, int j) { return abase5 + (tap / 3) * 34 * 80 + (tap % 3) * 80 + (j >> 1) * 34 * 80 + (j & 1) * 16 * 80; };
    auto waddr = [&](int tap, int j) { return bbase5 + (tap % 6) * 8192 + j * 1024; };
    auto mf = [&](int k, const bf16x8& w, const bf16x8& x) { asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(q[k]) : "v"(w), "v"(x)); };
#pragma unroll
    for (int j = 0; j < 8; ++j) xs[0][j] = lds128(smem, xaddr(0, j));
#pragma unroll
    for (int j = 0; j < 4; ++j) wl[j] = lds128(smem, waddr(0, j));
    for (int g = 0; g < groups; g += 2) {
#pragma unroll
      for (int st = 0; st < 18; ++st) {
        const int tap = st % 9, ntap = (st + 1) % 9, cur = st & 1, nxt = cur ^ 1;
        if (MODE == 9) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int gg = 0; gg < 16; ++gg) {
          mf(2 * gg, wl[(2 * gg) >> 3], xs[cur][(2 * gg) & 7]);
          mf(2 * gg + 1, wl[(2 * gg + 1) >> 3], xs[cur][(2 * gg + 1) & 7]);
          if (gg < 4) wh[gg] = lds128(smem, waddr(tap, 4 + gg));
          if (MODE == 9 && (gg == 4 || gg == 5))
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_ptr_t)(smem + 98304 + ((tap + 3) % 6) * 8192 + (gg - 4) * 4096 + wave * 1024), 16, dvoff,
                                                     ((st * 2 + gg) * 1024 + wave * 4096) & 0x1fc00, 0, 0);
          if (MODE == 9 && tap < 6 && (gg == 6 || gg == 7))
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_ptr_t)(smem + (((st / 9) & 1) ^ 1) * 49152 + (tap * 2 + gg - 6) * 4096 + wave * 1024), 16, dvoff,
                                                     ((st * 2 + gg) * 1024 + 65536 + wave * 4096) & 0x1fc00, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int gg = 0; gg < 16; ++gg) {
          mf(32 + 2 * gg, wh[(2 * gg) >> 3], xs[cur][(2 * gg) & 7]);
          mf(32 + 2 * gg + 1, wh[(2 * gg + 1) >> 3], xs[cur][(2 * gg + 1) & 7]);
          if (gg < 8) xs[nxt][gg] = lds128(smem, xaddr(ntap, gg));
          else if (gg < 12) wl[gg - 8] = lds128(smem, waddr(ntap, gg - 8));
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (MODE == 9) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_nop 7\ns_nop 7\ns_nop 7" ::: "memory");         // asm MFMA results -> VALU reads below
#pragma unroll
    for (int k = 0; k < 64; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[k >> 2][(k & 3) * 4 + e] = q[k][e];
  }
  if (MODE == 8) {
    const int abase8 = abase, bbase8 = bbase;
    bf16x8 xa[2][4], wa[2][4], xb[4], wb[4];
    auto xaddr = [&](int tap, int kk, int p) { return abase8 + (tap / 3) * 34 * 80 + (tap % 3) * 80 + p * 34 * 80 + kk * 32; };
    auto waddr = [&](int tap, int kk, int p) { return bbase8 + (tap % 6) * 8192 + p * 2048 + kk * 32; };
#pragma unroll
    for (int p = 0; p < 4; ++p) { xa[0][p] = lds128(smem, xaddr(0, 0, p)); wa[0][p] = lds128(smem, waddr(0, 0, p)); }
    for (int g = 0; g < groups; g += 2) {
#pragma unroll
      for (int st = 0; st < 18; ++st) {
        const int tap = st % 9, ntap = (st + 1) % 9, cur = st & 1, nxt = cur ^ 1;
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < 4; ++p) { xb[p] = lds128(smem, xaddr(tap, 1, p)); wb[p] = lds128(smem, waddr(tap, 1, p)); }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int p = 0; p < 4; ++p) acc[i * 4 + p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[cur][i], xa[cur][p], acc[i * 4 + p], 0, 0, 0);
#pragma unroll
        for (int p = 0; p < 4; ++p) { xa[nxt][p] = lds128(smem, xaddr(ntap, 0, p)); wa[nxt][p] = lds128(smem, waddr(ntap, 0, p)); }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int p = 0; p < 4; ++p) acc[i * 4 + p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[i], xb[p], acc[i * 4 + p], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  for (int g = 0; g < ((MODE >= 7 && MODE <= 10) ? 0 : groups); ++g) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      const int toff = (tap / 3) * 34 * 80 + (tap % 3) * 80;
      if (MODE == 0 || MODE == 1 || MODE == 4) {
        bf16x8 xf[2][4], wf[2][4];
        if (MODE != 4) {
#pragma unroll
          for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int p = 0; p < 4; ++p) {
              xf[kk][p] = lds128(smem, abase + toff + p * 34 * 80 + kk * 32);
              wf[kk][p] = lds128(smem, bbase + (tap % 6) * 8192 + p * 2048 + kk * 32);
            }
        } else {
#pragma unroll
          for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int p = 0; p < 4; ++p) { xf[kk][p] = lds128(smem, abase + p * 128 + kk * 32); wf[kk][p] = lds128(smem, bbase + p * 2048 + kk * 32); }
          if (g > 0 || tap > 0) {                                      // registers only after the first step
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
              for (int p = 0; p < 4; ++p) { asm volatile("" : "+v"(xf[kk][p]), "+v"(wf[kk][p])); }
          }
        }
        const bool rewrite = MODE == 1 && tap >= 2 && tap <= 7;        // rounds staged two steps earlier have landed
        GnSlot s0, s1;
        if (rewrite) {
          s0.raw = *(const uint4*)(smem + 49152 + (2 * (tap - 2)) * 4096 + slot0);
          s1.raw = *(const uint4*)(smem + 49152 + (2 * (tap - 2) + 1) * 4096 + slot0);
          s0.a0 = *(const f32x4*)(smem + tab); s0.a1 = *(const f32x4*)(smem + tab + 16); s0.b0 = *(const f32x4*)(smem + tab + 32); s0.b1 = *(const f32x4*)(smem + tab + 48);
          s1.a0 = s0.a0; s1.a1 = s0.a1; s1.b0 = s0.b0; s1.b1 = s0.b1;
          unpack8(s0.raw, s0.f); unpack8(s1.raw, s1.f);
        }
        const bool inside = ((lane * 7 + tap) & 31) != 0;              // border mask stand-in (per lane)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int p = 0; p < 4; ++p) {
              acc[i * 4 + p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kk][i], xf[kk][p], acc[i * 4 + p], 0, 0, 0);
              if (rewrite) {                                           // one slice of the rewrite per MFMA gap: 16 slices = 2 slots x 8 elements
                const int q = (kk * 16 + i * 4 + p);
                if (q < 16) {
                  GnSlot& s = (q < 8) ? s0 : s1;
                  const int e = q & 7;
                  const float a = e < 4 ? s.a0[e & 3] : s.a1[e & 3], b = e < 4 ? s.b0[e & 3] : s.b1[e & 3];
                  const float y = s.f[e] * a + b;
                  s.f[e] = inside ? silu_f(y) : 0.f;
                }
              }
            }
        if (rewrite) {
          *(uint4*)(smem + 49152 + (2 * (tap - 2)) * 4096 + slot0) = pack8(s0.f);
          *(uint4*)(smem + 49152 + (2 * (tap - 2) + 1) * 4096 + slot0) = pack8(s1.f);
        }
      } else if (MODE == 5 || MODE == 6) {
        typedef __attribute__((ext_vector_type(4))) float f32x4_t;
        f32x4_t* acc4 = (f32x4_t*)acc;                                  // 64 blocks of 16 x 16 in the same 256 registers
        const int l15 = lane & 15, q4 = lane >> 4;
        const int abase5 = ((4 * wave) * 34 + l15) * 80 + q4 * 16, bbase5 = 98304 + l15 * 64 + q4 * 16;
        bf16x8 xf[8], wf[8];
        if (MODE == 5) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            xf[j] = lds128(smem, abase5 + toff + (j >> 1) * 34 * 80 + (j & 1) * 16 * 80);
            wf[j] = lds128(smem, bbase5 + (tap % 6) * 8192 + j * 1024);
          }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) { xf[j] = lds128(smem, abase5 + j * 128); wf[j] = lds128(smem, bbase5 + j * 1024); }
          if (g > 0 || tap > 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { asm volatile("" : "+v"(xf[j]), "+v"(wf[j])); }
          }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int p = 0; p < 8; ++p) acc4[i * 8 + p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[p], acc4[i * 8 + p], 0, 0, 0);
      } else if (MODE == 7) {
        // handled outside the tap loop (needs two register sets alternating at compile time)
      } else if (MODE == 8) {
      } else if (MODE == 2) {
        // 16 positions, one 32 x 32 block each: A and B fragment per MFMA
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int pos = 0; pos < 16; ++pos) {
            const bf16x8 xv = lds128(smem, abase + (pos & 3) * 34 * 80 + (pos >> 2) * 160 + kk * 32 + toff % 2720);
            const bf16x8 wv = lds128(smem, bbase + (tap % 6) * 8192 + (pos & 3) * 2048 + (pos >> 2) * 512 + kk * 32);
            acc[pos] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv, xv, acc[pos], 0, 0, 0);
          }
      } else if (MODE == 3) {
        // 4 positions, 2 x 2 blocks each: 4 fragment reads per 4 MFMAs
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int pos = 0; pos < 4; ++pos) {
            bf16x8 xv[2], wv[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              xv[j] = lds128(smem, abase + pos * 34 * 80 + j * 160 + kk * 32 + toff % 2720);
              wv[j] = lds128(smem, bbase + (tap % 6) * 8192 + pos * 2048 + j * 512 + kk * 32);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int p = 0; p < 2; ++p) acc[pos * 4 + i * 2 + p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv[i], xv[p], acc[pos * 4 + i * 2 + p], 0, 0, 0);
          }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // (no drain inside the loop: fp32 accumulators of N(0,1) products stay far inside the range over 10^5 steps)
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += acc[k][r];
    out[((size_t)blockIdx.x * 16 + k) * 256 + tid] = t + sink;
  }
}

extern "C" void dove_set_error(const char*, ...) {}
template <int MODE>
static int launch(const void* init, void* out, int groups, int blocks, hipStream_t s) {
  static bool once = false;
  if (!once) { (void)hipFuncSetAttribute((const void*)convalt_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES); once = true; }
  hipLaunchKernelGGL((convalt_kernel<MODE>), dim3(blocks), dim3(256), LDS_BYTES, s, (const bf16_t*)init, (float*)out, groups);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int convalt(int mode, const void* init, void* out, int groups, int blocks, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  switch (mode) {
    case 0: return launch<0>(init, out, groups, blocks, s);
    case 1: return launch<1>(init, out, groups, blocks, s);
    case 2: return launch<2>(init, out, groups, blocks, s);
    case 3: return launch<3>(init, out, groups, blocks, s);
    case 4: return launch<4>(init, out, groups, blocks, s);
    case 5: return launch<5>(init, out, groups, blocks, s);
    case 6: return launch<6>(init, out, groups, blocks, s);
    case 7: return launch<7>(init, out, groups, blocks, s);
    case 8: return launch<8>(init, out, groups, blocks, s);
    case 9: return launch<9>(init, out, groups, blocks, s);
    case 10: return launch<10>(init, out, groups, blocks, s);
    default: return -1;
  }
}
