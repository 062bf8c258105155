// Flash-attention forward, NO-SHIFT path of dove_attention_fwd_bf16: every head the caller hands a finite score bound for (norm2) runs here
// first, WITHOUT a softmax shift; a head one of whose row sums leaves the window in which that is exact to rounding (attn_pipe::kRowSumMin /
// kRowSumMax below: impossible for a bound <= 80, i.e. every head of the DiT with LayerNorm gains up to ~2.5 x unit; not observed on real
// scores far above that, since the Cauchy-Schwarz bound is loose) marks itself NaN in norm2 and is recomputed by attn_fwd_kernel's running
// maximum in the same call.  Replaces F.scaled_dot_product_attention inside diffusers' CogVideoXAttnProcessor2_0,
// /root/reference/inference_script.py:483-489.  The one-wave-per-SIMD software-pipelined structure of the CDNA4 guide
// (cdna_hip_programming.md "4-wave, one-wave-per-SIMD, persistent structure"; MI355X_MICROARCH "single-issue instructions HIDDEN per
// v_mfma_f32_32x32x16_bf16 gap: <= 5"): on gfx950 the VALU work of one wave hides under the MFMAs of the SAME wave's stream and not under
// another wave's (tools/archive/coissue.py) - attn_fwd_kernel (two waves per SIMD, each QK^T -> softmax -> PV in turn) adds its 16 MFMAs and its
// ~200 ns of softmax per 32 x 64 wave-tile and reads 16 KB of LDS per 16 MFMAs.  Here:
//   * ONE wave per SIMD (256 threads, one workgroup per CU); a wave owns TWO query blocks of 32 (A, B): every K / V^T fragment read from LDS
//     feeds two MFMAs (half the LDS traffic per MFMA);
//   * software pipeline over the KV tiles inside the wave: a step issues the 16 QK^T MFMAs of tile j + 1 and the 16 PV MFMAs of tile j, and
//     behind every MFMA a fixed handful of the softmax instructions of tile j (8 chunks of {8 v_exp, 4 v_cvt_pk, 8 v_add} = one P^T fragment
//     each; a chunk spans 4 MFMA slots), pinned by sched_barrier;
//   * no shift: softmax is shift-invariant and fp32 / bf16 carry 8 exponent bits, so 2^s itself serves as long as the row sum stays in the
//     window below (round 5 took a static bound b <= 40 for that; round 6 checks the row sums themselves - one compare per row in the
//     epilogue - so the fast path no longer depends on the weights' LayerNorm gains).  The S chains start from the inline constant 0;
//   * register files chosen by hand (asm MFMAs, asm ds_reads): S in VGPRs (the exponentials read it), O, the Q fragments AND the K / V^T
//     fragments in AGPRs (only MFMAs touch them; LDS loads write AGPRs directly) - as builtins / plain loads the allocator moves S through
//     v_accvgpr_read (64 extra issues per step) or runs out of VGPRs;
//   * row sums as single v_add_f32 (asm: -O3 SLP-packs plain adds into v_pk_add_f32, an anti-lever beside MFMAs; v_dot2 on the packed P is
//     worse still) into 8 independent accumulators per block;
//   * K ring and V^T ring of four 8 KB slots each (64 KB: every LDS address is a lane constant + an immediate), four steps per loop trip so
//     that slots are compile-time; K tile j + 4 and V^T tile j + 2 are issued in step j (LDS-DMA), one barrier and one counted vmcnt per step;
//     fragments are read half a step ahead of their MFMAs, four counted lgkmcnt waits per step;
//   * the tile loop runs to a multiple of four steps: tiles at and past the ragged end are masked to -inf before their exponentials, so a
//     step on a tile that does not exist adds zeros - no tail variants, no copies of O at region boundaries.
// Operand layout: dove_qkv_post_bf16's (v_order 1), as attn_fwd_kernel.  Heads whose bound is NaN or infinite are LEFT UNTOUCHED, heads that
// mark themselves may be partly written: dove_attention_fwd_bf16 runs both kinds on attn_fwd_kernel's running maximum afterwards (whole heads).
// The last partial round of workgroups runs the one-block-per-wave form (NB = 1): 3.298 vs 3.341 ms at N = 18 226, 48 heads (profiles/r05_attn_tail.log).
// Measured (tools/attn2p_ab.py on the experiment twin tools/exp/attn2p_exp.hip, N = 18 226, 48 heads, within one process, profiles/r05_attn2p_*.log):
// 3.52 ms against attn_fwd_kernel's 3.77-3.96 ms by box (x 0.89-0.93; 1.16 PF), 2.56 vs 2.95 ms on all-zero operands; by parts (ns per step of
// 32 MFMAs on real operands): MFMAs alone 606, + fragment reads 698, + exponentials and packs 841, + row sums 916.
#include <atomic>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "../../include/dove_hip.h"

typedef __attribute__((address_space(3))) void* lds_ptr_t;

#define FENCE() __builtin_amdgcn_sched_barrier(0)
template <int V> using IC = std::integral_constant<int, V>;
template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(IC<I>{}); static_for<N, I + 1>(f); }
}

__device__ __forceinline__ bf16x8 frag4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  u32x4 v = {a, b, c, d};
  return __builtin_bit_cast(bf16x8, v);
}
// hipcc does not model an asm MFMA's hazards (guide 5.7); the pipeline keeps every consumer far behind its producer (S: >= 12 MFMA slots before
// the first exponential; P fragments: >= 2 slots before their PV MFMA; chains: 4 accumulators in rotation), prologue and epilogue pad by hand.
__device__ __forceinline__ void mfma_s_first(f32x16& d, const bf16x8& k, const bf16x8& q) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "a"(k), "a"(q));
}
__device__ __forceinline__ void mfma_s(f32x16& d, const bf16x8& k, const bf16x8& q) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "a"(k), "a"(q));
}
// (O as C++ values behind "+a" constraints: hipcc knows their liveness.  Naming fixed registers a[0:63] in the asm text with clobber lists - the
// "asm-owned" way - was tried and is WRONG here: between two statements that clobber a tuple the allocator is free to put a short-lived value
// (a V^T fragment read at slot 0 and dead by slot 18) into that tuple's registers.  The price of the C++ values: hipcc carries one of the four
// tuples through VGPRs around the main loop, 32 v_accvgpr moves per trip of four steps.)
__device__ __forceinline__ void mfma_o(f32x16& d, const bf16x8& v, const bf16x8& p) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(d) : "a"(v), "v"(p));
}
// LDS -> AGPR fragment read at lane address + immediate; completion is the caller's counted lgkmcnt
template <int OFF>
__device__ __forceinline__ void lds_frag(bf16x8& f, int addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(f) : "v"(addr), "n"(OFF));
}
__device__ __forceinline__ void add1(float& acc, float v) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(v)); }
__device__ __forceinline__ void exp2_inplace(float& v) { asm volatile("v_exp_f32 %0, %0" : "+v"(v)); }   // in place: no fresh destination whose reuse trips a trans hazard nop
__device__ __forceinline__ uint32_t cvtpk(float lo, float hi) {
  uint32_t r;
  asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

namespace attn_pipe {
constexpr int SLOT = 8192, VBASE = 4 * SLOT, LDS = 8 * SLOT;    // K slots 0-3 at 0 .. 24 KB, V^T slots 0-3 at 32 .. 56 KB
// Safe window of a row's UN-SHIFTED sum l = sum_j 2^s_ij, checked once per row in the epilogue.
//   l <= 2^100: no 2^s overflowed (that leaves inf or NaN in l) and O = sum_j p_j v_j <= l max |v| stays finite for |v| < 2^27;
//   l >= 2^-80: whatever flushed to zero below 2^-126 sums to < N 2^-126 <= 2^-102 (N < 2^24: the 31-bit buffer offsets), a 2^-22 share of
//   l, and every term that matters is a normal fp32 / bf16 number with its full relative precision.
// A score bound |s| <= b <= 80 implies N 2^-80 <= l <= N 2^80, inside the window for every N < 2^20: for such heads the check cannot fire.
constexpr float kRowSumMin = 0x1p-80f, kRowSumMax = 0x1p100f;
}

// NB = query blocks of 32 per wave.  2: the kernel described above (256 queries per workgroup).  1: the same pipeline with one block per
// wave (128 queries per workgroup, 16 MFMA slots per step, fragments not shared) - about 0.55 of the time per workgroup for half the
// queries: dove_attention_pipe_launch runs the LAST PARTIAL ROUND of workgroups this way (48 heads x 72 query blocks = 13.5 rounds of 256:
// the half round costs a whole one otherwise).  `item0`: index of this launch's first 256-query item in the head-major item list.
template <int NB>
__global__ __launch_bounds__(256, 1) void attn_pipe_kernel(const bf16_t* __restrict__ Qh, const bf16_t* __restrict__ Kh, const bf16_t* __restrict__ Vt,
                                                        bf16_t* __restrict__ O, long long N, long long Npad, long long ldo, int qblocks,
                                                        float* bound, int item0) {
  using namespace attn_pipe;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  // NB == 2: one item per workgroup, XCD-contiguous over this launch's items; NB == 1: two workgroups per item (query halves)
  const unsigned t = NB == 2 ? (unsigned)item0 + xcd_remap(blockIdx.x, gridDim.x) : (unsigned)item0 + (blockIdx.x >> 1);
  const int h = (int)(t / (unsigned)qblocks), qb = (int)(t - (unsigned)h * (unsigned)qblocks);
  if (!(bound[2 * h] * bound[2 * h + 1] < __builtin_inff())) return;   // NaN (also: marked by a workgroup below) or infinite: the running-maximum kernel owns this head
  const long long q0 = NB == 2 ? (long long)qb * 256 + wave * 64                      // block A: q0 .. q0 + 31, block B: q0 + 32 .. q0 + 63
                               : (long long)qb * 256 + (blockIdx.x & 1) * 128 + wave * 32;

  bf16x8 qf[NB][4];
#pragma unroll
  for (int x = 0; x < NB; ++x) {
    long long qrow = q0 + x * 32 + l31;
    if (qrow >= Npad) qrow = Npad - 1;                         // rows past the padded end are never stored
    const bf16_t* qp = Qh + ((long long)h * Npad + qrow) * 64 + hi * 8;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[x][kk] = *(const bf16x8*)(qp + kk * 16);
  }

  const int ntiles = (int)((N + 63) / 64);
  const int srow = tid >> 3;
  const int sc_ld = (tid & 7) ^ ((srow >> 1) & 7);
  const auto srd_k = __builtin_amdgcn_make_buffer_rsrc((void*)(Kh + (long long)h * Npad * 64), (short)0, (int)(Npad * 128), 0x00020000);
  const auto srd_v = __builtin_amdgcn_make_buffer_rsrc((void*)(Vt + (long long)h * 64 * Npad), (short)0, (int)(Npad * 128), 0x00020000);
  unsigned vk[2], vv[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    vk[j] = (unsigned)(((j * 32 + srow) * 64 + sc_ld * 8) * 2);
    vv[j] = (unsigned)((((long long)(j * 32 + srow)) * Npad + sc_ld * 8) * 2);
  }
  // one LDS-DMA instruction: half i of K tile `tile` into K slot S / of V^T tile `tile` into V slot S (a tile past the end: out of the
  // descriptor's range or stale columns - never used unmasked)
  auto dma_k = [&](auto slotc, int tile, int i) {
    constexpr int S = decltype(slotc)::value;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_k, (lds_ptr_t)(smem + S * SLOT + i * 4096 + wave * 1024), 16, vk[i], tile * (64 * 128), 0, 0);
  };
  auto dma_v = [&](auto slotc, int tile, int i) {
    constexpr int S = decltype(slotc)::value;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_v, (lds_ptr_t)(smem + VBASE + S * SLOT + i * 4096 + wave * 1024), 16, vv[i], tile * (64 * 2), 0, 0);
  };

  // lane part of the eight fragment addresses of a tile (row half b, 16-byte chunk pair c), XOR-swizzled like the staging side; the slot is an
  // immediate.  (dynamic LDS starts at the workgroup's LDS base: the address of smem itself)
  const int sbase = (int)(uintptr_t)(lds_ptr_t)smem;
  int koff[2][4];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int row = b * 32 + l31;
    const int sw = (row >> 1) & 7;
#pragma unroll
    for (int c = 0; c < 4; ++c) koff[b][c] = sbase + row * 128 + (((c * 2 + hi) ^ sw) << 4);
  }

  f32x16 o[NB][2];                                             // O^T [block][d half]                   (AGPRs)
  f32x16 sa[NB][2], sb[NB][2];                                  // S^T of two consecutive tiles: [block][key half]; roles swap every step (VGPRs)
  float ls[NB][8];                                             // row-sum partials [block][position in a chunk]: an add is a chunk behind the one it depends on
#pragma unroll
  for (int x = 0; x < NB; ++x) {
#pragma unroll
    for (int i = 0; i < 8; ++i) ls[x][i] = 0.f;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[x][d][r] = 0.f;
  }

  // keys at and past N: their scores become -inf BEFORE the exponentials (their K rows are zero: 2^0 = 1 would enter the row sums)
  auto mask_tail = [&](f32x16 (&s)[NB][2], int tile) {
    const long long kv0 = (long long)tile * 64;
#pragma unroll
    for (int x = 0; x < NB; ++x)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const long long kv = kv0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (kv >= N) s[x][kb][r] = -1e30f;
        }
  };

  bf16x8 pf[NB][4];                                            // P^T fragments of the tile being multiplied: [block][16-key slice c]   (VGPRs)
  bf16x8 kfr[4][2], vfr[4][2];                                 // K fragments [kk][key half], V^T fragments [slice c][d half]          (AGPRs)
  // softmax chunk (block x, slice c) of the tile held in `s`, cut in four quarters of FIVE instructions (one quarter per MFMA slot):
  // 8 exps, 4 packs, 8 adds
  uint32_t pk[4];
  auto chunk_q = [&](f32x16 (&s)[NB][2], auto x_, auto c_, auto q_) {
    constexpr int x = decltype(x_)::value, c = decltype(c_)::value, quarter = decltype(q_)::value;
    constexpr int kb = c >> 1, b = 8 * (c & 1);
    auto E = [&](int i) { float v = s[x][kb][b + i]; exp2_inplace(v); s[x][kb][b + i] = v; };
    auto C = [&](int i) { pk[i] = cvtpk(s[x][kb][b + 2 * i], s[x][kb][b + 2 * i + 1]); };
    auto A = [&](int i) { add1(ls[x][i], s[x][kb][b + i]); };
    // (packs and adds at least a quarter behind the exponentials they read: closer, hipcc pads every pack with an s_nop for the trans-use hazard)
    if constexpr (quarter == 0) { E(0); E(1); E(2); E(3); E(4); }
    if constexpr (quarter == 1) { E(5); E(6); E(7); A(0); A(1); }
    if constexpr (quarter == 2) { C(0); C(1); A(2); A(3); A(4); }
    if constexpr (quarter == 3) { C(2); C(3); A(5); A(6); A(7); pf[x][c] = frag4(pk[0], pk[1], pk[2], pk[3]); }
  };

  // ---- one pipeline step (P = j & 3, compile time): QK^T of tile j + 1 into `sn`, softmax + PV of tile j from `sc` ----
  // MFMA slot s (0 .. 31):  0-15  S chains, kk-major: (kk = s >> 2, kb = (s >> 1) & 1, block = s & 1)
  //                        16-31  O chains, slice-major: (c = (s - 16) >> 2, block = ((s - 16) >> 1) & 1, d half = s & 1)
  // VALU: chunk n = 2 c + block of tile j runs in slots 4 n - 4 .. 4 n - 1 (chunk 0 in the PREVIOUS step's last four slots, which therefore
  // carry chunk 0 of tile j + 1 here): PV slice c needs chunks 2 c and 2 c + 1 = done by slot 8 c + 3 < 16 + 4 c.
  // LDS reads, one every other slot, half a step ahead of their MFMAs: the 8 V^T fragments of tile j (V slot P) in slots 0, 2 .. 14
  // (fragment i: d half i & 1, slice i >> 1); the 8 K fragments of tile j + 2 (K slot (P + 2) & 3, visible since this step's barrier)
  // for the NEXT step's S half in slots 16, 18 .. 30 (fragment i: key half i & 1, kk = i >> 1).
  // Counted waits: reads return in order; four fragments ahead of each 8-slot block -> lgkmcnt(4) at slots 0, 8, 16, 24.
  // LDS-DMA: K tile j + 4 -> K slot P (its last reader, tile j, was read in step j - 2) in slots 3, 7; V^T tile j + 2 -> V slot (P + 2) & 3 (last
  // reader: tile j - 2 in step j - 2) in slots 11, 15.
  auto step = [&](int j, auto pc, f32x16 (&sc)[NB][2], f32x16 (&sn)[NB][2], auto maskc) {
    constexpr int P = decltype(pc)::value;
    constexpr bool kMask = decltype(maskc)::value;             // the tail trips: tile j + 1 may be ragged or past the end - mask it (no branch inside a step)
    if constexpr (NB == 2) asm volatile("" : "+a"(o[0][0]), "+a"(o[0][1]), "+a"(o[NB - 1][0]), "+a"(o[NB - 1][1]));   // O stays in the accumulator file across the step
    else asm volatile("" : "+a"(o[0][0]), "+a"(o[0][1]));                                                           // boundary (else hipcc parks a tuple in VGPRs around the loop)
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");           // all but the previous step's four loads: K tile j + 2 and V^T tile j have landed
    __builtin_amdgcn_s_barrier();
    FENCE();
    // NB == 1 (16 slots): S chains in slots 0-7 (kk = s >> 1, kb = s & 1), O chains in 8-15 (c = (s - 8) >> 1, d half = s & 1); chunk c of tile j
    // in slots 4 c - 4 .. 4 c - 1 (chunk 0 of tile j + 1 in slots 12-15): slice c is packed by slot 4 c - 1 < 8 + 2 c.  One fragment read per
    // slot: V^T fragment i (d = i & 1, c = i >> 1) in slot i for the MFMA of slot 8 + i, K fragment i of tile j + 2 in slot 8 + i for the NEXT
    // step's slot i; seven younger reads are in flight at every first use -> lgkmcnt(6) at the even slots covers the pair.
    constexpr int NS = 16 * NB, HALF = NS / 2;
    static_for<NS>([&](auto s_) {
      constexpr int s = decltype(s_)::value;
      if constexpr (NB == 2 && (s & 7) == 0) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
      if constexpr (NB == 1 && (s & 1) == 0) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
      // ---- the MFMA of this slot ----
      if constexpr (s < HALF) {
        constexpr int kk = NB == 2 ? s >> 2 : s >> 1, kb = NB == 2 ? (s >> 1) & 1 : s & 1, x = NB == 2 ? s & 1 : 0;
        if constexpr (kk == 0) mfma_s_first(sn[x][kb], kfr[0][kb], qf[x][0]);
        else mfma_s(sn[x][kb], kfr[kk][kb], qf[x][kk]);
      } else {
        constexpr int u = s - HALF;
        constexpr int c = NB == 2 ? u >> 2 : u >> 1, x = NB == 2 ? (u >> 1) & 1 : 0, d = s & 1;
        mfma_o(o[x][d], vfr[c][d], pf[x][c]);
      }
      // ---- the fragment read of this slot ----
      if constexpr (NB == 1 || (s & 1) == 0) {
        constexpr int i = NB == 2 ? (s & 15) >> 1 : s & 7;
        if constexpr (s < HALF) lds_frag<VBASE + P * SLOT>(vfr[i >> 1][i & 1], koff[i & 1][i >> 1]);
        else lds_frag<((P + 2) & 3) * SLOT>(kfr[i >> 1][i & 1], koff[i & 1][i >> 1]);
      }
      // ---- this step's LDS-DMAs (in the S half) ----
      constexpr int DS = NB == 2 ? 4 : 2;                       // slots 3, 7, 11, 15 / 1, 3, 5, 7
      if constexpr (s == DS - 1) dma_k(IC<P>{}, j + 4, 0);
      if constexpr (s == 2 * DS - 1) dma_k(IC<P>{}, j + 4, 1);
      if constexpr (s == 3 * DS - 1) dma_v(IC<((P + 2) & 3)>{}, j + 2, 0);
      if constexpr (s == 4 * DS - 1) dma_v(IC<((P + 2) & 3)>{}, j + 2, 1);
      // ---- keys at / past the end: S(j + 1) is complete (its chains ended with the S half), its first exponentials come in the last four slots ----
      if constexpr (s == (NB == 2 ? 21 : 10) && kMask) mask_tail(sn, j + 1);
      // ---- the softmax quarter of this slot ----
      if constexpr (s < NS - 4) {
        constexpr int n = (s >> 2) + 1;                         // chunk of tile j
        chunk_q(sc, IC<(NB == 2 ? (n & 1) : 0)>{}, IC<(NB == 2 ? (n >> 1) : n)>{}, IC<(s & 3)>{});
      } else {
        chunk_q(sn, IC<0>{}, IC<0>{}, IC<(s & 3)>{});
      }
      FENCE();
    });
  };

  // ---- prologue: K tiles 0 .. 3 and V^T tiles 0, 1 in flight; S(0), its chunk 0, the K fragments of tile 1 ----
  static_for<4>([&](auto tl) { dma_k(tl, decltype(tl)::value, 0); dma_k(tl, decltype(tl)::value, 1); });
  static_for<2>([&](auto tl) { dma_v(tl, decltype(tl)::value, 0); dma_v(tl, decltype(tl)::value, 1); });
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  FENCE();
  static_for<8>([&](auto i_) { constexpr int i = decltype(i_)::value; lds_frag<0>(kfr[i >> 1][i & 1], koff[i & 1][i >> 1]); });
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  FENCE();
  static_for<8 * NB>([&](auto s_) {
    constexpr int s = decltype(s_)::value;
    constexpr int kk = NB == 2 ? s >> 2 : s >> 1, kb = NB == 2 ? (s >> 1) & 1 : s & 1, x = NB == 2 ? s & 1 : 0;
    if constexpr (kk == 0) mfma_s_first(sa[x][kb], kfr[0][kb], qf[x][0]);
    else mfma_s(sa[x][kb], kfr[kk][kb], qf[x][kk]);
  });
  FENCE();
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");   // the last S MFMA -> its registers' next writer / reader: 18 wait states, by hand
  FENCE();
  static_for<8>([&](auto i_) { constexpr int i = decltype(i_)::value; lds_frag<SLOT>(kfr[i >> 1][i & 1], koff[i & 1][i >> 1]); });   // K tile 1: step 0's S half
  if (ntiles == 1) mask_tail(sa, 0);
  static_for<4>([&](auto q_) { chunk_q(sa, IC<0>{}, IC<0>{}, q_); });
  FENCE();
  // vmcnt bookkeeping: every step waits for "all but the last four" loads and issues four.  Nothing is in flight here, so step 0's wait is
  // trivially true (K tile 2 and V^T tile 0 have landed above), step 1's covers K tile 3 / V^T tile 1 (landed above too), and from step 2 on
  // the wait covers the loads issued two steps earlier = K tile j + 2, V^T tile j.
  // ---- the tile loop: four steps per trip (compile-time ring slots), to a multiple of four ----
  // (main trips: every tile a step touches is full and exists; then one or two tail trips whose steps mask S(j + 1) - a no-op on valid keys)
  const int last_full = ntiles - 1 - (((N & 63) != 0) ? 1 : 0);      // highest tile index that needs no mask
  int j = 0;
  for (; j + 4 <= last_full; j += 4) {
    step(j, IC<0>{}, sa, sb, std::false_type{});
    step(j + 1, IC<1>{}, sb, sa, std::false_type{});
    step(j + 2, IC<2>{}, sa, sb, std::false_type{});
    step(j + 3, IC<3>{}, sb, sa, std::false_type{});
  }
  FENCE();
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");   // region boundary: hipcc may copy O here, right behind the last PV MFMAs (guide 5.7)
  FENCE();
  for (; j < ntiles; j += 4) {
    step(j, IC<0>{}, sa, sb, std::true_type{});
    step(j + 1, IC<1>{}, sb, sa, std::true_type{});
    step(j + 2, IC<2>{}, sa, sb, std::true_type{});
    step(j + 3, IC<3>{}, sb, sa, std::true_type{});
  }
  FENCE();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");   // the last PV MFMAs -> v_accvgpr_read of O
  FENCE();
  static_for<NB>([&](auto x_) {
    constexpr int x = decltype(x_)::value;
    float l = ((ls[x][0] + ls[x][1]) + (ls[x][2] + ls[x][3])) + ((ls[x][4] + ls[x][5]) + (ls[x][6] + ls[x][7]));
    l += __shfl_xor(l, 32);
    const float inv = 1.0f / l;
    const long long q = q0 + x * 32 + l31;
    // the un-shifted row sum must have stayed where fp32 / bf16 lose nothing (header): outside, the head is handed to the running maximum
    if (q < N && !(l >= kRowSumMin && l <= kRowSumMax)) bound[2 * h] = __builtin_nanf("");
    if (q < N) {
      bf16_t* op = O + q * ldo + h * 64;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d = db * 32 + 8 * g + 4 * hi;
          uint2 w;
          w.x = pack_bf2(o[x][db][g * 4 + 0] * inv, o[x][db][g * 4 + 1] * inv);
          w.y = pack_bf2(o[x][db][g * 4 + 2] * inv, o[x][db][g * 4 + 3] * inv);
          *(uint2*)(op + d) = w;
        }
    }
  });
}

// Launch for the heads whose score bound (norm2: [heads][2] = max |q|^2, max |k|^2) is finite; norm2[2 h] becomes NaN for a head that has to be
// recomputed (dove_attention_fwd_bf16 runs those on attn_fwd_kernel).  Same operand contract as dove_attention_fwd_bf16.
// Items = (head, 256-query block) in head-major order.  Whole rounds of `cus` items run one per workgroup; a last partial round of at most
// half the CUs runs as twice as many one-block-per-wave workgroups (NB = 1), which take about 0.55 of a round.
static int pipe_cu_count() {
  static std::atomic<int> cus[DOVE_MAX_DEVICES] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  int n = (dev >= 0 && dev < DOVE_MAX_DEVICES) ? cus[dev].load(std::memory_order_relaxed) : 0;
  if (!n) {
    n = 256;
    (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    if (dev >= 0 && dev < DOVE_MAX_DEVICES) cus[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}
__attribute__((visibility("hidden"))) int dove_attention_pipe_launch(const void* Qh, const void* Kh, const void* Vt, void* O, long long N, long long Npad,
                                                                    int heads, long long ldo, float* norm2, void* stream) {
  static PerDeviceOnce attr_set;
  if (auto once_ = attr_set.guard()) {
    (void)hipFuncSetAttribute((const void*)attn_pipe_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, attn_pipe::LDS);
    (void)hipFuncSetAttribute((const void*)attn_pipe_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, attn_pipe::LDS);
  }
  const int qblocks = (int)((Npad + 255) / 256);
  const long long items = (long long)qblocks * heads;
  DOVE_CHECK_ARG(items < (1ll << 30), "attention_fwd: grid too large");
  const int cus = pipe_cu_count();
  long long rem = items % cus;
  if (rem * 2 > cus) rem = 0;                                    // a mostly full last round stays on the two-block kernel (a launch of at most half a
                                                                 // round runs entirely on the one-block form: twice the workgroups, each shorter)
  const long long main_items = items - rem;
  if (main_items > 0) {
    hipLaunchKernelGGL(attn_pipe_kernel<2>, dim3((unsigned)main_items), dim3(256), attn_pipe::LDS, (hipStream_t)stream, (const bf16_t*)Qh,
                       (const bf16_t*)Kh, (const bf16_t*)Vt, (bf16_t*)O, N, Npad, ldo, qblocks, norm2, 0);
    DOVE_CHECK_LAUNCH("dove_attention_fwd_bf16 (pipelined)");
  }
  if (rem > 0) {
    hipLaunchKernelGGL(attn_pipe_kernel<1>, dim3((unsigned)(2 * rem)), dim3(256), attn_pipe::LDS, (hipStream_t)stream, (const bf16_t*)Qh,
                       (const bf16_t*)Kh, (const bf16_t*)Vt, (bf16_t*)O, N, Npad, ldo, qblocks, norm2, (int)main_items);
    DOVE_CHECK_LAUNCH("dove_attention_fwd_bf16 (pipelined, last round)");
  }
  return DOVE_OK;
}
