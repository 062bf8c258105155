// Graph-level entry points (SURVEY.md 8(b) "proposed C-ABI beneath the facade"): a context that owns the packed weights and
// one workspace arena and runs whole stages of the one-step SR operator - so a host in any language drives the path with a
// handful of calls and no per-operator allocation:
//     dove_create -> dove_set_weight (diffusers state-dict names) x N -> dove_finalize_weights
//     dove_vae_encode / dove_dit_forward / dove_vae_decode, or dove_sr_clip = process_video
//     (/root/reference/inference_script.py:394-503: encode -> sample * scaling -> first-frame pad -> DiT -> get_velocity ->
//      decode -> (x*0.5+0.5).clamp(0,1)).
// This file is HOST orchestration only (plus two trivial conversion kernels): every FLOP goes through the operator entry
// points of this library, called in exactly the order dove_amd/vae.py and dove_amd/transformer.py call them, so the results
// are bit-identical to the Python facade (tests/test_graph_gpu.py) - the arithmetic is diffusers' (SURVEY.md App. A.1-A.6).
// Memory: activations come from ONE arena (first-fit free list, stream-ordered reuse: a block is released as soon as the
// last kernel reading it has been enqueued); causal-conv caches (diffusers' conv_cache: the last two input frames of every
// k_t = 3 convolution) are copied out of the batch's activations into per-conv buffers, so nothing outlives its frame-batch.
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "common.h"
#include "../../include/dove_hip.h"

namespace {

#define CHK(expr)            \
  do {                       \
    const int rc__ = (expr); \
    if (rc__ != 0) return rc__; \
  } while (0)
#define HIPCHK(expr)                                                              \
  do {                                                                            \
    const hipError_t e__ = (expr);                                                \
    if (e__ != hipSuccess) {                                                      \
      dove_set_error("%s failed: %s", #expr, hipGetErrorString(e__));             \
      return DOVE_ELAUNCH;                                                        \
    }                                                                             \
  } while (0)

inline long long ru(long long x, long long m) { return (x + m - 1) / m * m; }

// ---- conversion kernels (weight packing happens once, at dove_finalize_weights) -----------------------------------------------
__device__ inline float load_any(const void* p, int dt, long long i) {
  return dt == DOVE_F32 ? ((const float*)p)[i] : bf2f(((const bf16_t*)p)[i]);
}
// src [cout][cin][taps] (Conv3d / Conv2d / Linear weight, natural layout) -> dst rows [row0, row0+cout) of [taps][cout_pad][cin_pad] bf16
__global__ void pack_weight_kernel(const void* src, int dt, int cout, int cin, int taps, int cout_pad, int cin_pad, int row0, bf16_t* dst) {
  const long long n = (long long)taps * cout * cin_pad;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int ci = (int)(i % cin_pad);
    const long long r = i / cin_pad;
    const int co = (int)(r % cout), t = (int)(r / cout);
    const float v = ci < cin ? load_any(src, dt, ((long long)co * cin + ci) * taps + t) : 0.f;
    dst[((long long)t * cout_pad + row0 + co) * cin_pad + ci] = f2bf(v);
  }
}
// decoder.conv_out split by spatial tap (dove_conv_out_gather): src [C][cin][kt][3][3] -> dst [kt][32][cin_pad] bf16, row (dy*3+dx)*C + c
__global__ void pack_taps_kernel(const void* src, int dt, int C, int cin, int kt, int cin_pad, bf16_t* dst) {
  const long long n = (long long)kt * 9 * C * cin;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int ci = (int)(i % cin);
    long long r = i / cin;
    const int row = (int)(r % (9 * C)), t = (int)(r / (9 * C));
    const int c = row % C, tap = row / C;                       // tap = dy*3 + dx
    dst[((long long)t * 32 + row) * cin_pad + ci] = f2bf(load_any(src, dt, (((long long)c * cin + ci) * kt + t) * 9 + tap));
  }
}
// encoder.conv_in with the spatial taps in the input channels (dove_cl_im2col3x3_from_ncthw): src [cout][C][kt][3][3] ->
// dst [kt][cout_pad][cin_pad] bf16, column (dy*3+dx)*C + c
__global__ void pack_in_taps_kernel(const void* src, int dt, int cout, int C, int kt, int cout_pad, int cin_pad, bf16_t* dst) {
  const long long n = (long long)kt * cout * 9 * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int col = (int)(i % (9 * C));
    long long r = i / (9 * C);
    const int co = (int)(r % cout), t = (int)(r / cout);
    const int c = col % C, tap = col / C;
    dst[((long long)t * cout_pad + co) * cin_pad + col] = f2bf(load_any(src, dt, (((long long)co * C + c) * kt + t) * 9 + tap));
  }
}
// temporal sums of a packed kt = 3 weight [3][taps][n]: dst [2][taps][n] = w0 + w1, (w0 + w1) + w2 in fp32, rounded once (dove_amd/ops.py pack_conv)
__global__ void pack_first_kernel(const bf16_t* __restrict__ w, long long n, bf16_t* __restrict__ dst) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float s01 = bf2f(w[i]) + bf2f(w[n + i]);
    dst[i] = f2bf(s01);
    dst[n + i] = f2bf(s01 + bf2f(w[2 * n + i]));
  }
}
// pair sums of a packed kt = 3 weight [3][taps][n]: dst [2][taps][n] = w0 + w1, w1 + w2 in fp32, rounded once (dove_amd/ops.py pack_conv, pair=True)
__global__ void pack_pair_kernel(const bf16_t* __restrict__ w, long long n, bf16_t* __restrict__ dst) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    dst[i] = f2bf(bf2f(w[i]) + bf2f(w[n + i]));
    dst[n + i] = f2bf(bf2f(w[n + i]) + bf2f(w[2 * n + i]));
  }
}
// sub-pixel form of an upsample-fused 3x3 conv: w [3][3][n] packed bf16 -> dst [4 phases][2x2][n]; fp32 sums in (dh, dw) order, one rounding
__global__ void pack_sub_kernel(const bf16_t* __restrict__ w, long long n, bf16_t* __restrict__ dst) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    for (int py = 0; py < 2; ++py)
      for (int px = 0; px < 2; ++px) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int dh = 0; dh < 3; ++dh)
          for (int dw = 0; dw < 3; ++dw) {
            const int a = (py + dh + 1) / 2 - py, b = (px + dw + 1) / 2 - px;
            acc[2 * a + b] += bf2f(w[(long long)(dh * 3 + dw) * n + i]);
          }
        for (int tp = 0; tp < 4; ++tp) dst[(long long)((2 * py + px) * 4 + tp) * n + i] = f2bf(acc[tp]);
      }
  }
}
__global__ void to_f32_kernel(const void* src, int dt, long long n, float* dst) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) dst[i] = load_any(src, dt, i);
}
__global__ void to_bf16_kernel(const void* src, int dt, long long n, bf16_t* dst) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) dst[i] = f2bf(load_any(src, dt, i));
}

// rows of `width` bytes (multiple of 2) between pitched device buffers (hipMemcpy2DAsync rejects some pitches)
__global__ void copy2d_kernel(char* dst, long long dpitch, const char* src, long long spitch, long long width, int height) {
  const long long w2 = width >> 1, n = w2 * height;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / w2, c = i - r * w2;
    ((uint16_t*)(dst + r * dpitch))[c] = ((const uint16_t*)(src + r * spitch))[c];
  }
}
// GroupNorm pair exchange of a split frame-batch: msg = (sum, sumsq)[32] + element count; the pair's totals = a + b
__global__ void gn_msg_kernel(const double* __restrict__ sums, double count, double* __restrict__ msg) {
  const int i = threadIdx.x;
  if (i < 64) msg[i] = sums[i];
  else if (i == 64) msg[64] = count;
}
__global__ void gn_add_kernel(const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ out) {
  const int i = threadIdx.x;
  if (i < 65) out[i] = a[i] + b[i];
}
inline int copy2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, int height, hipStream_t s) {
  const long long n = (long long)(width >> 1) * height;
  const unsigned grid = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(copy2d_kernel, dim3(grid ? grid : 1), dim3(256), 0, s, (char*)dst, (long long)dpitch, (const char*)src, (long long)spitch,
                     (long long)width, height);
  return hipGetLastError() == hipSuccess ? 0 : DOVE_ELAUNCH;
}

struct Raw { const void* p; std::vector<long long> shape; int dt; long long numel() const { long long n = 1; for (auto d : shape) n *= d; return n; } };
struct Packed { bf16_t* w = nullptr; float* bias = nullptr; bf16_t* w_first = nullptr, *w_sub = nullptr;   // dove_conv_desc.w_first (kt == 3) / .w_sub (3x3, kt == 1)
                bf16_t* w_pair = nullptr;                                                                   // dove_conv_desc.w_pair (kt == 3, on request)
                int kt = 1, kh = 1, kw = 1, cin = 0, cin_pad = 0, cout = 0, cout_pad = 0;
                int cout_store() const { return (int)ru(cout, 4); } };
struct Tensor { bf16_t* p = nullptr; int T = 0, H = 0, W = 0, C = 0; long long elems() const { return (long long)T * H * W * C; } size_t bytes() const { return (size_t)elems() * 2; } };
struct Stats { float* stats = nullptr; };

// first-fit arena over one device allocation; offsets 256-byte aligned.
// STREAM TRACKING (round 6; off unless a VAE stage alternates its frame-batches between two streams): allocation and release are host-side
// and assume that whoever gets a block next runs BEHIND its last user.  On one stream that is stream order.  With two, `release` records an
// event on the stream that is current at the release (every use of the block is ordered before that point: an item's activations live and
// die on the item's stream, a conv cache produced on the other stream is released only after the conv that read it has been enqueued here),
// the free range keeps the latest event per stream, and `alloc` makes the current stream wait for the events of the OTHER stream on the range
// it carves from.  Conservative (a split range hands its events to both parts), never wrong.
struct Arena {
  char* base = nullptr; size_t cap = 0; bool owned = false; size_t used = 0, high = 0;   // high = peak of bytes in use
  std::map<size_t, size_t> free_;                       // offset -> size
  std::unordered_map<void*, size_t> live;
  struct Ev { long long e[2] = {-1, -1}; };             // per free range: pool index of the LAST release event of each stream (-1: none)
  std::map<size_t, Ev> free_ev;                         // offset -> events (only while `track`)
  bool track = false; int cur = 0; hipStream_t streams[2] = {nullptr, nullptr};
  std::vector<hipEvent_t> pool; size_t pool_next = 0;
  // While tracking, the space is split at `mid`: stream 0 allocates below it, stream 1 above (each falls back to the other half when its own is
  // full), and free ranges are never merged across it - otherwise the two streams keep carving from ranges the other one just released and every
  // allocation becomes a cross-stream wait (measured: the streams then run in lock-step and the second stream buys nothing).
  size_t mid = 0;
  void reset() { free_.clear(); live.clear(); free_ev.clear(); used = 0; pool_next = 0; track = false; cur = 0; mid = 0; if (cap) free_[0] = cap; }
  void begin_tracking(hipStream_t s0, hipStream_t s1) {
    track = true; cur = 0; streams[0] = s0; streams[1] = s1; free_ev.clear();
    mid = (size_t)ru((long long)(cap / 2), 256);
    auto it = free_.upper_bound(mid);
    if (it != free_.begin()) {
      --it;
      if (it->first < mid && it->first + it->second > mid) { const size_t end = it->first + it->second; it->second = mid - it->first; free_[mid] = end - mid; }
    }
  }
  void end_tracking() {
    track = false; cur = 0; free_ev.clear();
    if (mid) {
      auto hi = free_.find(mid);
      if (hi != free_.end() && hi != free_.begin()) { auto lo = std::prev(hi); if (lo->first + lo->second == mid) { lo->second += hi->second; free_.erase(hi); } }
      mid = 0;
    }
  }
  bool home(size_t off) const { return !track || ((off >= mid) == (cur == 1)); }
  long long next_event() {                              // events are handed out in record order: a larger index on one stream = a later point of it
    if (pool_next == pool.size()) { hipEvent_t e = nullptr; (void)hipEventCreateWithFlags(&e, hipEventDisableTiming); pool.push_back(e); }
    return (long long)pool_next++;
  }
  static void merge_ev(Ev& into, const Ev& other) { for (int i = 0; i < 2; ++i) if (other.e[i] > into.e[i]) into.e[i] = other.e[i]; }
  void on_alloc(size_t off) {                           // the range at `off` is about to be carved: order the current stream behind the other stream's releases
    if (!track) return;
    auto it = free_ev.find(off);
    if (it == free_ev.end()) return;
    if (it->second.e[cur ^ 1] >= 0) (void)hipStreamWaitEvent(streams[cur], pool[(size_t)it->second.e[cur ^ 1]], 0);
  }
  // transient activations are carved first-fit from the FRONT; long-lived blocks (conv caches, which persist for a whole stage)
  // from the BACK, so they do not fragment the space the big per-layer tensors cycle through
  void* alloc(size_t n, bool from_back = false) {
    n = (size_t)ru((long long)(n ? n : 1), 256);
    for (int pass = track ? 0 : 1; pass < 2; ++pass) {        // pass 0: the current stream's own half only
      if (from_back) {
        for (auto it = free_.rbegin(); it != free_.rend(); ++it) {
          if (it->second >= n && (pass == 1 || home(it->first))) {
            const size_t off = it->first, sz = it->second;
            on_alloc(off);
            free_.erase(std::next(it).base());
            if (sz > n) free_[off] = sz - n;                        // the front remainder keeps the range's events (same key)
            else if (track) free_ev.erase(off);
            void* p = base + off + (sz - n);
            live[p] = n;
            used += n;
            if (used > high) high = used;
            return p;
          }
        }
        continue;
      }
      for (auto it = free_.begin(); it != free_.end(); ++it) {
        if (it->second >= n && (pass == 1 || home(it->first))) {
          const size_t off = it->first, rest = it->second - n;
          on_alloc(off);
          free_.erase(it);
          if (track) {
            auto ie = free_ev.find(off);
            if (ie != free_ev.end()) { const Ev ev = ie->second; free_ev.erase(ie); if (rest) free_ev[off + n] = ev; }
          }
          if (rest) free_[off + n] = rest;
          void* p = base + off;
          live[p] = n;
          used += n;
          if (used > high) high = used;
          return p;
        }
      }
    }
    return nullptr;
  }
  void release(void* p, bool shared = false) {
    if (!p) return;
    auto it = live.find(p);
    if (it == live.end()) return;
    size_t off = (char*)p - base, n = it->second;
    live.erase(it);
    used -= n;
    Ev ev;
    if (track) { const long long e = next_event(); (void)hipEventRecord(pool[(size_t)e], streams[cur]); ev.e[cur] = e; }
    if (track && shared) {                                      // a block BOTH streams read (a conv cache): whoever gets it next runs behind both
      const long long e = next_event(); (void)hipEventRecord(pool[(size_t)e], streams[cur ^ 1]); ev.e[cur ^ 1] = e;
    }
    auto nx = free_.lower_bound(off);
    if (nx != free_.end() && off + n == nx->first && !(mid && nx->first == mid)) {
      if (track) { auto ie = free_ev.find(nx->first); if (ie != free_ev.end()) { Ev old = ie->second; free_ev.erase(ie); merge_ev(old, ev); ev = old; } }
      n += nx->second; nx = free_.erase(nx);
    }
    if (nx != free_.begin()) {
      auto pv = std::prev(nx);
      if (pv->first + pv->second == off && !(mid && off == mid)) {
        if (track) { auto ie = free_ev.find(pv->first); if (ie != free_ev.end()) { Ev old = ie->second; free_ev.erase(ie); merge_ev(old, ev); ev = old; } }
        off = pv->first; n += pv->second; free_.erase(pv);
      }
    }
    free_[off] = n;
    if (track) free_ev[off] = ev;
  }
};

struct PackedMx { uint8_t* q = nullptr; uint32_t* s = nullptr; float* bias = nullptr; int rows = 0, K = 0; };   // dove_mx_quant_bf16 layout
struct DitBlock {
  bf16_t *mod1_w, *mod2_w; float *mod1_b, *mod2_b;      // norm1.linear / norm2.linear (M = 1 GEMV operands)
  float *ln1_g, *ln1_b, *ln2_g, *ln2_b, *nq_g, *nq_b, *nk_g, *nk_b;
  Packed qkv, out, ff1, ff2;
  PackedMx qkv8, out8, ff18, ff28;                      // DOVE_OPT_DIT_LINEAR_MXFP8: the same four linears in MXFP8 (bf16 copies freed)
  float *m1 = nullptr, *g1 = nullptr, *m2 = nullptr, *g2 = nullptr;   // per-timestep: mod [2][2][D], gate [2][D]
};

}  // namespace

struct dove_ctx {
  dove_model_config cfg;
  int device = 0;
  std::unordered_map<std::string, Raw> raw;
  std::unordered_map<std::string, Packed> pc;           // VAE convs by diffusers module path (+ ".yb" for SpatialNorm conv_y || conv_b)
  std::unordered_map<std::string, std::pair<float*, float*>> aff;
  std::vector<void*> owned;                             // weight memory owned by the library
  bool finalized = false;
  // DiT
  Packed pe_proj, pe_text, proj_out;
  bf16_t *te1_w = nullptr, *te2_w = nullptr, *modout_w = nullptr; float *te1_b = nullptr, *te2_b = nullptr, *modout_b = nullptr;
  float *nf_g = nullptr, *nf_b = nullptr, *no_g = nullptr, *no_b = nullptr, *final_mod = nullptr;
  std::vector<DitBlock> blocks;
  int mod_t = -1;
  float *emb = nullptr, *e1 = nullptr, *temb = nullptr, *vtmp = nullptr;
  // rope cache
  std::vector<float> rope_host; float* rope_dev = nullptr; int rope_t = 0, rope_h = 0, rope_w = 0;
  // attention operand buffers (pad rows must stay zero)
  bf16_t *Qh = nullptr, *Kh = nullptr, *Vt = nullptr; long long attn_n = 0;
  // VAE
  std::unordered_map<std::string, Tensor> cache;        // conv_cache of the running clip (views or copies)
  std::unordered_map<std::string, void*> cache_owner;   // arena block a cache entry keeps alive (a retained conv input) or the copy itself
  std::unordered_map<std::string, long long> cache_stride;   // nb > 1: elements between two instances' cache frames (dove_conv_desc.cache_stride)
  std::unordered_map<std::string, bool> cache_pair;     // the cache entry is a known bit-identical frame pair (what dove_conv_desc.tdup == 1 declares of it)
  int nb = 1;                                           // > 1 while tiled() runs nb same-shaped tiles as one batch (Tensor.T = nb x frames)
  float* gn_ws = nullptr; int gn_ws_rows = 0;
  float* gn_ws2 = nullptr;                              // the second VAE stream's statistics scratch (allocated with the stream)
  // Two-stream VAE (DOVE_OPT_VAE_STREAMS, default 2; dove_amd/vae.py n_streams): the frame-batches of an un-tiled single-rank stage alternate
  // between the caller's stream and `vae_stream2`, ordered only by one event per causal conv (`cache_ev`: recorded behind the conv and its cache
  // copy, waited for by the next batch's conv of the same name) - one batch's HBM-bound GroupNorm kernels run beside the tail of the other's convs.
  // Same kernels on the same inputs: bit-identical.  The arena tracks the streams (struct Arena).
  int opt_vae_streams = 2;
  hipStream_t vae_stream2 = nullptr;
  bool vae_multi = false;
  std::unordered_map<std::string, long long> cache_ev;   // conv name -> arena event index of the batch that wrote the cache entry
  float* conv_out_bias = nullptr;                       // != NULL: decoder.conv_out runs tap-split ("decoder.conv_out.taps" + gather)
  Arena arena;
  std::string err;
  // multi-GPU (dove_comm_init*): this rank's frame-batches of a clip; halos travel rank -> rank + 1 in layer order
  int rank = 0, nranks = 1;
  dove_xfer_fn send_fn = nullptr, recv_fn = nullptr; void* xfer_user = nullptr;
  void* rccl_lib = nullptr; void* rccl_comm = nullptr;
  bool halo_recv = false, halo_send = false;            // set by the batch loop around the rank's first / last batch
  // Halo transport (round 6): the COMPUTE stream never runs a transfer.  Receives go to `recv_stream` (pre-posted at the start of the rank's
  // first work item once the (stage, shape) has been seen - the list of halos is recorded on the first pass, like dove_amd/dist.py HaloCache),
  // sends to `send_stream` behind an event recorded after the producing launch; the consuming conv waits on the receive's event.  The halo
  // callbacks may differ from send_fn / recv_fn: the RCCL binding gives each direction its own communicator (rccl_halo_send / _recv).
  hipStream_t recv_stream = nullptr, send_stream = nullptr;
  dove_xfer_fn halo_send_fn = nullptr, halo_recv_fn = nullptr;
  bool halo_can_prepost = true;                         // false: one communicator for both directions (RCCL without ncclCommSplit) - receives are posted where consumed
  std::vector<hipEvent_t> ev_pool; size_t ev_next = 0;  // events of the running stage (reused from stage to stage)
  struct HaloSlot { void* p; size_t bytes; hipEvent_t ev; };
  std::unordered_map<std::string, HaloSlot> halo_posted;
  std::map<std::string, std::vector<std::pair<std::string, size_t>>> halo_plans;   // (stage, shape, ranks) -> halos of the first work item, in conv order
  std::vector<std::pair<std::string, size_t>> halo_record;
  std::string halo_key;
  bool halo_sent = false;
  long long stat_halo_preposted = 0, stat_halo_blocking = 0, stat_halo_sent = 0;   // of the last VAE stage (dove_get_option DOVE_STAT_*)
  // a PIECE of a frame-batch split over a rank pair (more ranks than frame-batches: BASELINE configs[2], 8 ranks on 4 batches): GroupNorm
  // sums are combined with `piece_partner`, Upsample3D is told which piece starts an odd batch (dove_amd/dist.py plan_pieces)
  int piece_role = 0;                                   // 0: whole batch, 1: head (keeps the first frame single), 2: tail
  int piece_partner = -1; bool piece_lower = false;
  dove_group_fn group_begin = nullptr, group_end = nullptr;   // bracket of one exchange for rendezvous transports (RCCL: ncclGroupStart / End)
  // options (dove_set_option)
  bool opt_tiling = false, opt_linear_mx = false, opt_attn_mx = false, opt_weight_sums = true;
  int sample_h = 480, sample_w = 720;                   // vae/config.json sample_height / sample_width: tile geometry of enable_tiling()
  uint8_t *Q8 = nullptr, *K8 = nullptr, *V8 = nullptr, *Vs8 = nullptr; long long attn8_n = 0;   // MXFP8 attention operands
  int depth = 0;                                        // nesting of stage entry points (dove_sr_clip calls the others)
};

namespace {

int dev_alloc(dove_ctx* c, size_t bytes, void** out) {
  void* p = nullptr;
  HIPCHK(hipMalloc(&p, bytes ? bytes : 4));
  c->owned.push_back(p);
  *out = p;
  return 0;
}
int need(dove_ctx* c, const std::string& name, const Raw** out, int ndim_min = 1) {
  auto it = c->raw.find(name);
  if (it == c->raw.end()) { dove_set_error("weight '%s' was never set", name.c_str()); return DOVE_EINVAL; }
  if ((int)it->second.shape.size() < ndim_min) { dove_set_error("weight '%s' has too few dimensions", name.c_str()); return DOVE_EINVAL; }
  *out = &it->second;
  return 0;
}
int to_f32(dove_ctx* c, const std::string& name, float** out, long long expect = -1) {
  const Raw* r; CHK(need(c, name, &r));
  if (expect >= 0 && r->numel() != expect) { dove_set_error("weight '%s' has %lld elements, expected %lld", name.c_str(), r->numel(), expect); return DOVE_EINVAL; }
  void* p; CHK(dev_alloc(c, (size_t)r->numel() * 4, &p));
  hipLaunchKernelGGL(to_f32_kernel, dim3(256), dim3(256), 0, 0, r->p, r->dt, r->numel(), (float*)p);
  *out = (float*)p;
  return 0;
}
int to_bf16(dove_ctx* c, const std::string& name, bf16_t** out) {
  const Raw* r; CHK(need(c, name, &r));
  void* p; CHK(dev_alloc(c, (size_t)r->numel() * 2, &p));
  hipLaunchKernelGGL(to_bf16_kernel, dim3(1024), dim3(256), 0, 0, r->p, r->dt, r->numel(), (bf16_t*)p);
  *out = (bf16_t*)p;
  return 0;
}
// pack one or several weights (stacked along cout) + optional biases into the implicit-GEMM layout (dove_amd/ops.py pack_conv)
// `sub`: build the sub-pixel sums of a 3x3 Conv2d (read by upsample-fused launches only); `pair`: build the pair sums of a 3x3x3 conv
int pack(dove_ctx* c, const std::vector<std::string>& wnames, const std::vector<std::string>& bnames, Packed* out, bool sub = false, bool pair = false) {
  Packed q;
  int cout = 0;
  std::vector<const Raw*> ws;
  for (auto& n : wnames) {
    const Raw* r; CHK(need(c, n, &r, 2));
    ws.push_back(r);
    const auto& s = r->shape;
    const int nd = (int)s.size();
    const int kt = nd == 5 ? (int)s[2] : 1, kh = nd >= 4 ? (int)s[nd - 2] : 1, kw = nd >= 4 ? (int)s[nd - 1] : 1;
    if (cout == 0) { q.kt = kt; q.kh = kh; q.kw = kw; q.cin = (int)s[1]; }
    else if (q.kt != kt || q.kh != kh || q.kw != kw || q.cin != (int)s[1]) { dove_set_error("cannot stack '%s': shape mismatch", n.c_str()); return DOVE_EINVAL; }
    cout += (int)s[0];
  }
  q.cout = cout;
  q.cin_pad = (q.cin <= 32 || q.cin % 64) ? (int)ru(q.cin, 32) : q.cin;
  q.cout_pad = (int)ru(cout, 32);
  const int taps = q.kt * q.kh * q.kw;
  const size_t wbytes = (size_t)taps * q.cout_pad * q.cin_pad * 2;
  void* wp; CHK(dev_alloc(c, wbytes, &wp));
  HIPCHK(hipMemsetAsync(wp, 0, wbytes, 0));
  int row0 = 0;
  for (auto r : ws) {
    hipLaunchKernelGGL(pack_weight_kernel, dim3(1024), dim3(256), 0, 0, r->p, r->dt, (int)r->shape[0], q.cin, taps, q.cout_pad, q.cin_pad, row0, (bf16_t*)wp);
    row0 += (int)r->shape[0];
  }
  q.w = (bf16_t*)wp;
  if (q.kt == 3) {                                              // temporal sums for the cache-less first frames (dove_conv_desc.w_first)
    const long long n = (long long)q.kh * q.kw * q.cout_pad * q.cin_pad;
    void* wf; CHK(dev_alloc(c, (size_t)2 * n * 2, &wf));
    hipLaunchKernelGGL(pack_first_kernel, dim3(1024), dim3(256), 0, 0, (const bf16_t*)wp, n, (bf16_t*)wf);
    q.w_first = (bf16_t*)wf;
  }
  if (q.kt == 3 && pair) {                                       // pair sums for a conv behind a time-doubling upsampler (dove_conv_desc.w_pair)
    const long long n = (long long)q.kh * q.kw * q.cout_pad * q.cin_pad;
    void* wpp; CHK(dev_alloc(c, (size_t)2 * n * 2, &wpp));
    hipLaunchKernelGGL(pack_pair_kernel, dim3(1024), dim3(256), 0, 0, (const bf16_t*)wp, n, (bf16_t*)wpp);
    q.w_pair = (bf16_t*)wpp;
  }
  if (q.kt == 1 && q.kh == 3 && q.kw == 3 && sub) {              // sub-pixel form for the upsample-fused use of this conv (dove_conv_desc.w_sub)
    const long long n = (long long)q.cout_pad * q.cin_pad;
    void* wsb; CHK(dev_alloc(c, (size_t)16 * n * 2, &wsb));
    hipLaunchKernelGGL(pack_sub_kernel, dim3(1024), dim3(256), 0, 0, (const bf16_t*)wp, n, (bf16_t*)wsb);
    q.w_sub = (bf16_t*)wsb;
  }
  if (!bnames.empty()) {
    void* bp; CHK(dev_alloc(c, (size_t)q.cout_pad * 4, &bp));
    HIPCHK(hipMemsetAsync(bp, 0, (size_t)q.cout_pad * 4, 0));
    int off = 0;
    for (auto& n : bnames) {
      const Raw* r; CHK(need(c, n, &r));
      hipLaunchKernelGGL(to_f32_kernel, dim3(16), dim3(256), 0, 0, r->p, r->dt, r->numel(), (float*)bp + off);
      off += (int)r->numel();
    }
    q.bias = (float*)bp;
  }
  *out = q;
  return 0;
}
// nn.Linear weight [N][K] bf16 (a Packed of a 1x1x1 "conv") -> MXFP8 operand; the bf16 block is freed
int to_mx(dove_ctx* c, Packed* pc, PackedMx* out) {
  if (pc->cout_pad != pc->cout || pc->cin_pad != pc->cin || pc->cin % 256 || pc->cout % 256) {
    dove_set_error("MXFP8 linear: weight [%d][%d] must have N and K multiples of 256", pc->cout, pc->cin);
    return DOVE_EINVAL;
  }
  out->rows = pc->cout; out->K = pc->cin; out->bias = pc->bias;
  void* q; void* sc;
  CHK(dev_alloc(c, (size_t)out->rows * out->K, &q));
  CHK(dev_alloc(c, (size_t)(out->K / 256) * out->rows * 2 * 4, &sc));
  out->q = (uint8_t*)q; out->s = (uint32_t*)sc;
  CHK(dove_mx_quant_bf16(pc->w, out->rows, out->K, out->q, out->s, nullptr));
  HIPCHK(hipStreamSynchronize(nullptr));
  for (auto it = c->owned.begin(); it != c->owned.end(); ++it)
    if (*it == (void*)pc->w) { (void)hipFree(pc->w); c->owned.erase(it); pc->w = nullptr; break; }
  return 0;
}
bool ends_with(const std::string& s, const char* suf) { const size_t n = strlen(suf); return s.size() >= n && s.compare(s.size() - n, n, suf) == 0; }

// ---- operator wrappers (allocation from the arena + argument marshalling; mirrors dove_amd/ops.py) -----------------------------
inline float* gn_scratch(dove_ctx* c) { return (c->vae_multi && c->arena.cur == 1) ? c->gn_ws2 : c->gn_ws; }   // one statistics scratch per VAE stream
int alloc_t(dove_ctx* c, int T, int H, int W, int C, Tensor* t) {
  t->T = T; t->H = H; t->W = W; t->C = C;
  t->p = (bf16_t*)c->arena.alloc(t->bytes());
  if (!t->p) { dove_set_error("workspace exhausted: need %zu more bytes (capacity %zu); call dove_workspace_bytes / dove_set_workspace", t->bytes(), c->arena.cap); return DOVE_EINVAL; }
  return 0;
}
void free_t(dove_ctx* c, Tensor& t) { c->arena.release(t.p); t.p = nullptr; }

struct ConvOpt { const Tensor* cache = nullptr; int stride = 1, pad_h = -1, pad_w = -1, up = 0, tmode = 0, t_out = -1, act = 0;
                 const bf16_t* resid = nullptr; int ldr = 0; const float* gate = nullptr; long long gate_split = 0; bf16_t* out = nullptr; int ldo = -1;
                 float gn_eps = -1.f; float** gn_stats = nullptr;
                 bool out_f32 = false;
                 int nb = 1; long long cache_stride = 0; int tdup = 0; };   // nb > 1: dove_conv_desc.nb (x.T = nb x frames, t_out per instance)   // out_f32: the returned Tensor holds float [..][ldo] (its C counts bf16 units = 2 * ldo)
// x [T,H,W,cin_pad] -> out (allocated unless opt.out).  When opt.gn_eps >= 0 and the kernel fuses GroupNorm statistics, *opt.gn_stats
// receives [32][2] (mean, rstd) from the arena (caller releases); otherwise it is left NULL.
int conv(dove_ctx* c, const Tensor& x, const Packed& pc, const ConvOpt& o, Tensor* out, void* stream) {
  if (x.C != pc.cin_pad) { dove_set_error("conv: input has %d channels, packed weight expects %d", x.C, pc.cin_pad); return DOVE_EINVAL; }
  const int ph = o.pad_h < 0 ? (pc.kh - 1) / 2 : o.pad_h, pw = o.pad_w < 0 ? (pc.kw - 1) / 2 : o.pad_w;
  const int nb = o.nb > 1 ? o.nb : 1;
  const int t_in = x.T / nb;
  const int t_out = o.t_out < 0 ? t_in : o.t_out;             // per instance
  int ho, wo;
  if (o.stride == 1) { ho = x.H << o.up; wo = x.W << o.up; }
  else { ho = (x.H + 1 - pc.kh) / o.stride + 1; wo = (x.W + 1 - pc.kw) / o.stride + 1; }
  const int ldo = o.ldo < 0 ? pc.cout_store() : o.ldo;
  Tensor y;
  if (o.out) { y.p = o.out; y.T = nb * t_out; y.H = ho; y.W = wo; y.C = ldo; }
  else CHK(alloc_t(c, nb * t_out, ho, wo, o.out_f32 ? 2 * ldo : ldo, &y));
  dove_conv_desc d;
  memset(&d, 0, sizeof(d));
  d.struct_size = (unsigned)sizeof(d);
  d.x = x.p; d.cache = o.cache ? o.cache->p : nullptr; d.w = pc.w; d.bias = pc.bias; d.resid = o.resid; d.gate = o.gate; d.out = y.p;
  d.t_in = t_in; d.h_in = x.H; d.w_in = x.W; d.cin = x.C; d.t_out = t_out; d.h_out = ho; d.w_out = wo;
  d.nb = nb; d.cache_stride = o.cache ? o.cache_stride : 0;
  d.w_first = (o.cache || !c->opt_weight_sums) ? nullptr : pc.w_first;
  d.w_sub = (o.up == 1 && c->opt_weight_sums) ? pc.w_sub : nullptr;
  if (o.tdup && c->opt_weight_sums && pc.w_pair && pc.kt == 3 && !(o.tdup == 2 && o.cache)) { d.tdup = o.tdup; d.w_pair = pc.w_pair; }   // dove_amd/ops.py conv
  d.cout_pad = pc.cout_pad; d.cout_store = pc.cout_store();
  d.kt = pc.kt; d.kh = pc.kh; d.kw = pc.kw; d.stride = o.stride; d.pad_h = ph; d.pad_w = pw; d.up = o.up; d.tmode = o.tmode; d.act = o.act;
  d.ldo = ldo; d.ldr = o.ldr; d.gate_split = o.gate_split;
  d.out_f32 = o.out_f32 ? 1 : 0;
  float* partial = nullptr;
  long long rows = 0;
  if (o.gn_eps >= 0.f && o.gn_stats && ldo == pc.cout_store()) {
    rows = dove_conv_gn_partial_rows(&d);
    if (rows > 0) {
      partial = (float*)c->arena.alloc((size_t)rows * 64 * 4);
      if (!partial) { dove_set_error("workspace exhausted (GroupNorm partials)"); return DOVE_EINVAL; }
      d.gn_partial = partial;
    }
  }
  CHK(dove_conv_igemm_bf16(&d, stream));
  if (o.gn_stats) *o.gn_stats = nullptr;
  if (partial && c->piece_partner >= 0) {
    // a piece of a split frame-batch: the statistics are the PAIR's - hand the raw fp64 sums of this piece on (norm_silu adds the partner's)
    double* sums = (double*)c->arena.alloc(64 * sizeof(double));
    if (!sums) { dove_set_error("workspace exhausted (GroupNorm sums)"); return DOVE_EINVAL; }
    CHK(dove_groupnorm_sums_from_partials(partial, rows, gn_scratch(c), sums, stream));
    c->arena.release(partial);
    *o.gn_stats = (float*)sums;
  } else if (partial) {
    float* st = (float*)c->arena.alloc((size_t)nb * 64 * 4);
    const double count = (double)t_out * ho * wo * (pc.cout_store() / 32);
    CHK(dove_groupnorm_finalize_partials_nb(partial, rows / nb, nb, count, o.gn_eps, gn_scratch(c), (size_t)c->gn_ws_rows * 64 * 4, st, stream));
    c->arena.release(partial);
    *o.gn_stats = st;
  }
  *out = y;
  return 0;
}
int linear(dove_ctx* c, const bf16_t* x, long long N, const Packed& pc, ConvOpt o, bf16_t** out, void* stream) {
  Tensor xt; xt.p = (bf16_t*)x; xt.T = 1; xt.H = 1; xt.W = (int)N; xt.C = pc.cin_pad;
  Tensor y;
  CHK(conv(c, xt, pc, o, &y, stream));
  *out = y.p;
  return 0;
}

// ---- halo transport on the context's own streams (see dove_ctx) ----
int next_event(dove_ctx* c, hipEvent_t* out) {
  if (c->ev_next == c->ev_pool.size()) {
    hipEvent_t e;
    HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    c->ev_pool.push_back(e);
  }
  *out = c->ev_pool[c->ev_next++];
  return 0;
}
// `to` waits for everything enqueued on `from` so far
int stream_after(dove_ctx* c, hipStream_t to, hipStream_t from) {
  hipEvent_t e; CHK(next_event(c, &e));
  HIPCHK(hipEventRecord(e, from));
  HIPCHK(hipStreamWaitEvent(to, e, 0));
  return 0;
}
int ensure_comm_streams(dove_ctx* c) {
  if (!c->recv_stream) HIPCHK(hipStreamCreateWithFlags(&c->recv_stream, hipStreamNonBlocking));
  if (!c->send_stream) HIPCHK(hipStreamCreateWithFlags(&c->send_stream, hipStreamNonBlocking));
  return 0;
}
// Start of the rank's FIRST work item of a stage (it receives halos from rank - 1): with a recorded plan, every receive is posted now, in conv
// order, into buffers taken from the back of the arena; without one this pass records it.  An arena too small for the whole list falls back to
// posting each receive where it is consumed (the recording path), which is always correct.
int halo_begin(dove_ctx* c, const std::string& key, void* stream) {
  c->halo_key = key;
  c->halo_record.clear();
  c->halo_posted.clear();
  CHK(ensure_comm_streams(c));
  auto it = c->halo_plans.find(key);
  if (it == c->halo_plans.end() || !c->halo_can_prepost) return 0;
  // A custom transport matches messages in order per (source, destination) pair and has no channels: when this work item is the UPPER piece of
  // a split frame-batch, its partner's GroupNorm sums and its halos both come from rank - 1, interleaved in layer order - receives posted ahead
  // would pair with the wrong messages.  (RCCL: the halos have their own communicators; dove_amd/dist.py: the side group.)
  if (!c->rccl_comm && c->piece_partner == c->rank - 1) return 0;
  std::vector<void*> got;
  for (auto& nb : it->second) {
    void* p = c->arena.alloc(nb.second, true);
    if (!p) { for (void* q : got) c->arena.release(q); return 0; }
    got.push_back(p);
  }
  CHK(stream_after(c, c->recv_stream, (hipStream_t)stream));   // the buffers' previous users (this stream, earlier) are done before anything lands
  size_t i = 0;
  for (auto& nb : it->second) {
    dove_ctx::HaloSlot sl; sl.p = got[i++]; sl.bytes = nb.second;
    CHK(c->halo_recv_fn(c->xfer_user, c->rank - 1, sl.p, sl.bytes, (void*)c->recv_stream));
    CHK(next_event(c, &sl.ev));
    HIPCHK(hipEventRecord(sl.ev, c->recv_stream));
    c->halo_posted[nb.first] = sl;
  }
  return 0;
}
// the halo of conv `name` for the rank's first work item: *out owns an arena block
int halo_fetch(dove_ctx* c, const std::string& name, size_t bytes, void** out, void* stream) {
  auto it = c->halo_posted.find(name);
  if (it != c->halo_posted.end()) {
    if (it->second.bytes != bytes) { dove_set_error("halo of %s: pre-posted %zu bytes, the conv wants %zu", name.c_str(), it->second.bytes, bytes); return DOVE_EINVAL; }
    HIPCHK(hipStreamWaitEvent((hipStream_t)stream, it->second.ev, 0));
    *out = it->second.p;
    c->halo_posted.erase(it);
    ++c->stat_halo_preposted;
    return 0;
  }
  if (!c->halo_posted.empty()) { dove_set_error("halo of %s was not in the pre-posted plan (%zu others still posted)", name.c_str(), c->halo_posted.size()); return DOVE_EINVAL; }
  void* p = c->arena.alloc(bytes, true);
  if (!p) { dove_set_error("workspace exhausted (halo of %s: %zu bytes)", name.c_str(), bytes); return DOVE_EINVAL; }
  CHK(ensure_comm_streams(c));
  CHK(stream_after(c, c->recv_stream, (hipStream_t)stream));   // the block's previous users
  CHK(c->halo_recv_fn(c->xfer_user, c->rank - 1, p, bytes, (void*)c->recv_stream));
  CHK(stream_after(c, (hipStream_t)stream, c->recv_stream));
  c->halo_record.emplace_back(name, bytes);
  ++c->stat_halo_blocking;
  *out = p;
  return 0;
}
int halo_end_first_item(dove_ctx* c) {
  if (!c->halo_posted.empty()) { dove_set_error("%zu pre-posted halos were never consumed", c->halo_posted.size()); c->halo_posted.clear(); c->halo_plans.erase(c->halo_key); return DOVE_EINVAL; }
  if (!c->halo_record.empty() && !c->halo_plans.count(c->halo_key)) c->halo_plans[c->halo_key] = c->halo_record;
  c->halo_record.clear();
  return 0;
}
// a conv of the rank's LAST work item hands its cache (the last kt - 1 input frames) to rank + 1: behind an event, on the send stream
int halo_publish(dove_ctx* c, void* p, size_t bytes, void* stream) {
  CHK(ensure_comm_streams(c));
  CHK(stream_after(c, c->send_stream, (hipStream_t)stream));
  CHK(c->halo_send_fn(c->xfer_user, c->rank + 1, p, bytes, (void*)c->send_stream));
  c->halo_sent = true;
  ++c->stat_halo_sent;
  return 0;
}
// end of a stage: the memory the sends read is recycled by the next stage - the caller's stream continues behind the last of them
int halo_stage_end(dove_ctx* c, void* stream) {
  if (c->halo_sent) CHK(stream_after(c, (hipStream_t)stream, c->send_stream));
  c->halo_sent = false;
  return 0;
}

// ---- exchanges between ranks, built on the context's send / recv callbacks (dove_comm_init*: RCCL or any transport).  A transport
// whose sends rendezvous with the matching receive (RCCL) brackets every exchange with group_begin / group_end; a buffering transport
// (the tests' mailbox) needs no bracket: all sends of an exchange are issued before its receives. ----
int comm_begin(dove_ctx* c, void* stream) { return c->group_begin ? c->group_begin(c->xfer_user, stream) : 0; }
int comm_end(dove_ctx* c, void* stream) { return c->group_end ? c->group_end(c->xfer_user, stream) : 0; }
// symmetric swap with one partner: the lower rank sends first, the higher receives first (deadlock-free also without a bracket)
int pair_exchange(dove_ctx* c, int partner, bool lower, void* send_buf, void* recv_buf, size_t bytes, void* stream) {
  CHK(comm_begin(c, stream));
  if (lower) { CHK(c->send_fn(c->xfer_user, partner, send_buf, bytes, stream)); CHK(c->recv_fn(c->xfer_user, partner, recv_buf, bytes, stream)); }
  else { CHK(c->recv_fn(c->xfer_user, partner, recv_buf, bytes, stream)); CHK(c->send_fn(c->xfer_user, partner, send_buf, bytes, stream)); }
  return comm_end(c, stream);
}
// all-to-all of byte blocks: block j of `send` (offset soff[j], scnt[j] bytes) goes to rank j, block j of `recv` comes from rank j
int all_to_all(dove_ctx* c, const void* send, const size_t* soff, const size_t* scnt, void* recv, const size_t* roff, const size_t* rcnt, void* stream) {
  const int R = c->nranks, me = c->rank;
  CHK(comm_begin(c, stream));
  for (int k = 1; k < R; ++k) { const int j = (me + k) % R; if (scnt[j]) CHK(c->send_fn(c->xfer_user, j, (char*)send + soff[j], scnt[j], stream)); }
  for (int k = 1; k < R; ++k) { const int j = (me - k + R) % R; if (rcnt[j]) CHK(c->recv_fn(c->xfer_user, j, (char*)recv + roff[j], rcnt[j], stream)); }
  CHK(comm_end(c, stream));
  if (scnt[me]) HIPCHK(hipMemcpyAsync((char*)recv + roff[me], (const char*)send + soff[me], scnt[me], hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return 0;
}
// all-gather of variable-size blocks: every rank contributes `mine` (cnt[me] bytes) and ends with all blocks at off[j] of `out`
int all_gather_v(dove_ctx* c, const void* mine, void* out, const size_t* off, const size_t* cnt, void* stream) {
  const int R = c->nranks, me = c->rank;
  CHK(comm_begin(c, stream));
  for (int k = 1; k < R; ++k) { const int j = (me + k) % R; if (cnt[me]) CHK(c->send_fn(c->xfer_user, j, (void*)mine, cnt[me], stream)); }
  for (int k = 1; k < R; ++k) { const int j = (me - k + R) % R; if (cnt[j]) CHK(c->recv_fn(c->xfer_user, j, (char*)out + off[j], cnt[j], stream)); }
  CHK(comm_end(c, stream));
  if (cnt[me] && (const char*)mine != (char*)out + off[me]) HIPCHK(hipMemcpyAsync((char*)out + off[me], mine, cnt[me], hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return 0;
}

// ---- VAE (dove_amd/vae.py) ------------------------------------------------------------------------------------------------------
void frame_batches(int n, int batch, std::vector<std::pair<int, int>>* out) {
  const int nb = n / batch > 0 ? n / batch : 1, rem = n % batch;
  for (int i = 0; i < nb; ++i) {
    const int s = batch * i + (i == 0 ? 0 : rem), e = batch * (i + 1) + rem;
    out->push_back({s, e < n ? e : n});
  }
}
void spatial_norm_tmap(int tf, int tz, std::vector<int>* m) {
  m->resize(tf);
  if (tf > 1 && tf % 2 == 1) {
    for (int t = 0; t < tf; ++t) (*m)[t] = tz == 1 ? 0 : (t == 0 ? 0 : 1 + ((t - 1) * (tz - 1)) / (tf - 1));
  } else {
    for (int t = 0; t < tf; ++t) (*m)[t] = (int)(((long long)t * tz) / tf);
  }
}
// CogVideoXCausalConv3d with conv_cache (the last kt-1 INPUT frames of this conv, consumed by the next frame-batch).
// `x_owned`: x is an arena block the caller is done with.  Then, memory permitting, the cache simply KEEPS x alive and points at
// its last frames (what the Python facade's tensor views do; no copy) - the block is released when the next batch's conv has
// replaced the entry.  Otherwise (x is a view, the arena is tight, or x has fewer frames than the halo) the frames are copied into
// a small block at the back of the arena and x is released here.  Either way x must not be touched by the caller afterwards.
int cconv(dove_ctx* c, Tensor& x, bool x_owned, const std::string& name, ConvOpt o, Tensor* out, void* stream) {
  const Packed& pc = c->pc.at(name);
  o.nb = c->nb;
  if (pc.kt == 1) {
    CHK(conv(c, x, pc, o, out, stream));
    if (x_owned) free_t(c, x);
    return 0;
  }
  const int k = pc.kt - 1;
  if (c->nb > 1) {
    // nb same-shaped tiles in one launch (tiled()): x [nb][Ti][H][W][C]; the cache entry is nb x k frames, instance b's at p + b * stride
    // - a view of the retained previous input when x is an arena block with >= k frames per instance, else a dense copy
    const int nb = c->nb, Ti = x.T / nb;
    const long long frame = (long long)x.H * x.W * x.C;
    hipStream_t s = (hipStream_t)stream;
    auto it = c->cache.find(name);
    Tensor prev;
    const bool have = it != c->cache.end();
    if (have) { prev = it->second; o.cache = &prev; o.cache_stride = c->cache_stride[name]; }
    const int tdup_in = o.tdup;
    if (o.tdup == 1 && have && !c->cache_pair[name]) o.tdup = 0;   // the declaration covers the cache: only a doubled input of >= k frames leaves a pair
    CHK(conv(c, x, pc, o, out, stream));
    c->cache_pair[name] = tdup_in != 0 && Ti >= k;
    void* old_owner = have ? c->cache_owner[name] : nullptr;
    Tensor nc; nc.T = nb * k; nc.H = x.H; nc.W = x.W; nc.C = x.C;
    if (x_owned && Ti >= k && c->arena.cap - c->arena.used > 8 * x.bytes()) {
      nc.p = x.p + (long long)(Ti - k) * frame;
      c->cache[name] = nc; c->cache_owner[name] = x.p; c->cache_stride[name] = (long long)Ti * frame;
      c->arena.release(old_owner);
      x.p = nullptr;
      return 0;
    }
    nc.p = (bf16_t*)c->arena.alloc(nc.bytes(), true);
    if (!nc.p) { dove_set_error("workspace exhausted (conv cache of %s: %zu bytes)", name.c_str(), nc.bytes()); return DOVE_EINVAL; }
    if (Ti >= k) {
      CHK(copy2d(nc.p, (size_t)k * frame * 2, x.p + (long long)(Ti - k) * frame, (size_t)Ti * frame * 2, (size_t)k * frame * 2, nb, s));
    } else {                                                    // fewer frames than the halo: slide each instance's padded window
      const long long pstride = have ? c->cache_stride[name] : 0;
      for (int b = 0; b < nb; ++b) {
        bf16_t* dst = nc.p + (long long)b * k * frame;
        const bf16_t* xb = x.p + (long long)b * Ti * frame;
        for (int j = 0; j < k - Ti; ++j) {
          const bf16_t* src = have ? prev.p + (long long)b * pstride + (long long)(j + Ti) * frame : xb;
          HIPCHK(hipMemcpyAsync(dst + (long long)j * frame, src, (size_t)frame * 2, hipMemcpyDeviceToDevice, s));
        }
        HIPCHK(hipMemcpyAsync(dst + (long long)(k - Ti) * frame, xb, (size_t)Ti * frame * 2, hipMemcpyDeviceToDevice, s));
      }
    }
    c->cache[name] = nc; c->cache_owner[name] = nc.p; c->cache_stride[name] = (long long)k * frame;
    c->arena.release(old_owner);
    if (x_owned) free_t(c, x);
    return 0;
  }
  auto it = c->cache.find(name);
  if (c->halo_recv && it == c->cache.end()) {
    // first batch of a rank > 0: the conv_cache the previous batch would have left arrives from rank - 1 (same layer order there)
    Tensor h; h.T = k; h.H = x.H; h.W = x.W; h.C = x.C;
    void* hp = nullptr;
    CHK(halo_fetch(c, name, h.bytes(), &hp, stream));          // lands on the context's receive stream; this stream only waits on its event
    h.p = (bf16_t*)hp;
    c->cache[name] = h; c->cache_owner[name] = h.p;
    c->cache_pair[name] = true;                                // the previous rank's last frames: the same declaration holds there
    it = c->cache.find(name);
  }
  Tensor prev;
  const bool have = it != c->cache.end();
  if (have) prev = it->second;
  o.cache = have ? &prev : nullptr;
  if (c->vae_multi) {                                         // the entry was written on the OTHER stream: behind its conv and its cache copy
    auto ie = c->cache_ev.find(name);
    if (ie != c->cache_ev.end()) HIPCHK(hipStreamWaitEvent((hipStream_t)stream, c->arena.pool[(size_t)ie->second], 0));
  }
  auto mark = [&]() -> int {                                  // ... and this batch's entry is complete from here on
    if (!c->vae_multi) return 0;
    const long long e = c->arena.next_event();
    HIPCHK(hipEventRecord(c->arena.pool[(size_t)e], (hipStream_t)stream));
    c->cache_ev[name] = e;
    return 0;
  };
  const int tdup_in = o.tdup;
  if (o.tdup == 1 && have && !c->cache_pair[name]) o.tdup = 0;   //  a slid window of a short batch is not a pair: three taps, not two
  // retained view (below): the next batch's halo IS the tail of this conv's input, complete before the conv runs - the other stream may
  // start its conv of this name as soon as that input exists (dove_amd/vae.py records its event at the same point)
  const bool retain = x_owned && x.T >= k && c->arena.cap - c->arena.used > 8 * x.bytes();
  if (retain) CHK(mark());
  CHK(conv(c, x, pc, o, out, stream));
  c->cache_pair[name] = tdup_in != 0 && x.T >= k;
  const long long frame = (long long)x.H * x.W * x.C;
  hipStream_t s = (hipStream_t)stream;
  void* old_owner = have ? c->cache_owner[name] : nullptr;
  Tensor nc; nc.T = k; nc.H = x.H; nc.W = x.W; nc.C = x.C;
  if (retain) {
    nc.p = x.p + (long long)(x.T - k) * frame;               // retain: a view of x's last k frames, x stays alive
    c->cache[name] = nc; c->cache_owner[name] = x.p;
    c->arena.release(old_owner, true);                        // the conv that read the old entry is already enqueued (on this stream; the
    x.p = nullptr;                                            // entry's producer is the other one: Arena::release `shared`)
    if (c->halo_send) CHK(halo_publish(c, nc.p, nc.bytes(), stream));
    return 0;
  }
  nc.p = (bf16_t*)c->arena.alloc(nc.bytes(), true);
  if (!nc.p) { dove_set_error("workspace exhausted (conv cache of %s: %zu bytes)", name.c_str(), nc.bytes()); return DOVE_EINVAL; }
  if (x.T >= k) {
    HIPCHK(hipMemcpyAsync(nc.p, x.p + (long long)(x.T - k) * frame, (size_t)k * frame * 2, hipMemcpyDeviceToDevice, s));
  } else {                                                    // fewer frames than the halo: slide the padded window
    for (int j = 0; j < k - x.T; ++j) {
      const bf16_t* src = have ? prev.p + (long long)(j + x.T) * frame : x.p;   // old cache frames move up / frame 0 replicated
      HIPCHK(hipMemcpyAsync(nc.p + (long long)j * frame, src, (size_t)frame * 2, hipMemcpyDeviceToDevice, s));
    }
    HIPCHK(hipMemcpyAsync(nc.p + (long long)(k - x.T) * frame, x.p, (size_t)x.T * frame * 2, hipMemcpyDeviceToDevice, s));
  }
  c->cache[name] = nc; c->cache_owner[name] = nc.p;
  c->arena.release(old_owner, true);
  if (x_owned) free_t(c, x);
  if (c->halo_send) CHK(halo_publish(c, nc.p, nc.bytes(), stream));
  CHK(mark());
  return 0;
}
// frames one frame-batch of t input frames leaves behind the encoder's / decoder's temporal stages (diffusers' Downsample3D
// avg_pool1d with the first frame kept for odd t; Upsample3D: t -> 2t - 1 (t odd, > 1), 2t (t even), 1)
int n_temporal_stages(const dove_ctx* c) {
  int n = 0;
  for (int r = c->cfg.vae_temporal_compression; r > 1; r >>= 1) ++n;
  return n < c->cfg.vae_num_blocks - 1 ? n : c->cfg.vae_num_blocks - 1;
}
int enc_batch_frames(const dove_ctx* c, int t) {
  for (int i = 0; i < n_temporal_stages(c); ++i) t = t > 1 ? (t % 2 ? 1 + (t - 1) / 2 : t / 2) : 1;
  return t;
}
int dec_batch_frames(const dove_ctx* c, int t) {
  for (int i = 0; i < n_temporal_stages(c); ++i) t = t > 1 ? (t % 2 ? 2 * t - 1 : 2 * t) : 1;
  return t;
}
// Work list per rank of the halo-exact VAE (dove_amd/dist.py plan_pieces).  With at most as many ranks as frame-batches every rank gets a
// contiguous group of whole batches, earlier ranks take the extras.  With more ranks, batches are split in two PIECES on consecutive ranks (a
// rank pair): the conv halos flow rank -> rank + 1 exactly as between batches, GroupNorm statistics are combined across the pair, Upsample3D is
// told which piece starts an odd batch.  Split points keep every temporal 2:1 pooling pair inside one piece: pixel-frame batches (encoder) of
// 8k(+1) frames split at 4k(+1); latent batches (decoder) of 2 or 3 frames split before the last frame.
struct Piece { int s, e, role, partner; bool lower; };
int plan_pieces(const std::vector<std::pair<int, int>>& fb, int world, bool enc, std::vector<std::vector<Piece>>* out) {
  const int nb = (int)fb.size();
  out->assign(world, {});
  if (world <= nb) {
    const int base = nb / world, extra = nb % world;
    int b = 0, active = 0;
    for (int r = 0; r < world; ++r) {
      const int k = base + (r < extra ? 1 : 0);
      for (int i = 0; i < k; ++i, ++b) (*out)[r].push_back({fb[b].first, fb[b].second, 0, -1, false});
      if (k) ++active;
    }
    return active;
  }
  int r = 0;
  for (int i = 0; i < nb; ++i) {
    const int s = fb[i].first, e = fb[i].second, n = e - s;
    const int spare = (world - r) - (nb - i);                     // ranks we can still spend on splitting
    int head; bool can;
    if (enc) { head = (n % 2) + 4 * ((n - n % 2) / 8); can = n - (n % 2) >= 8 && (n - n % 2) % 8 == 0; }
    else { head = n - 1; can = n >= 2; }
    if (spare >= 1 && can) {
      const bool odd = n % 2 == 1;
      (*out)[r].push_back({s, s + head, odd ? 1 : 2, r + 1, true});
      (*out)[r + 1].push_back({s + head, e, 2, r, false});
      r += 2;
    } else {
      (*out)[r].push_back({s, e, 0, -1, false});
      r += 1;
    }
  }
  return r;
}
// frames a piece of t input frames leaves behind the temporal stages (whole batch: role 0; see Downsample3D / Upsample3D above)
int piece_out_frames(const dove_ctx* c, bool enc, int t, int role) {
  if (role == 0) return enc ? enc_batch_frames(c, t) : dec_batch_frames(c, t);
  for (int i = 0; i < n_temporal_stages(c); ++i) {
    if (enc) t = t > 1 ? (t % 2 ? 1 + (t - 1) / 2 : t / 2) : 1;  // the split keeps pooling pairs inside a piece: the piece's own parity decides
    else t = role == 1 ? 2 * t - 1 : 2 * t;
  }
  return t;
}
// this rank's pieces of a stage (n frames in), the output-frame offset of each, and the number of ranks that got work
struct RankPlan { std::vector<Piece> mine; std::vector<int> out_first; int active = 1; int total_out = 0; std::vector<int> rank_first, rank_count; };
void rank_plan(const dove_ctx* c, bool enc, int n, RankPlan* rp) {
  std::vector<std::pair<int, int>> fb;
  frame_batches(n, enc ? c->cfg.vae_enc_batch : c->cfg.vae_dec_batch, &fb);
  std::vector<std::vector<Piece>> plan;
  rp->active = plan_pieces(fb, c->nranks, enc, &plan);
  rp->rank_first.assign(c->nranks, 0); rp->rank_count.assign(c->nranks, 0);
  int f = 0;
  for (int r = 0; r < c->nranks; ++r) {
    rp->rank_first[r] = f;
    for (auto& pc : plan[r]) {
      const int k = piece_out_frames(c, enc, pc.e - pc.s, pc.role);
      if (r == c->rank) { rp->mine.push_back(pc); rp->out_first.push_back(f); }
      rp->rank_count[r] += k;
      f += k;
    }
  }
  rp->total_out = f;
}
void set_piece(dove_ctx* c, const Piece* pc) {
  c->piece_role = pc ? pc->role : 0; c->piece_partner = pc ? pc->partner : -1; c->piece_lower = pc ? pc->lower : false;
}
int norm_silu(dove_ctx* c, const Tensor& x, float* fused_stats, const std::string& name, const Tensor* zq, Tensor* out, void* stream) {
  const float eps = c->cfg.vae_norm_eps;
  float* stats = fused_stats;
  const int nb = c->nb, Ti = x.T / nb;
  if (c->piece_partner >= 0) {
    // this rank holds a PIECE of the frame-batch: whole-batch statistics = my (sum, sumsq, count) + the partner's, same fp64 finalize
    // (dove_amd/dist.py _run_pieces hook: 65 doubles each way, one round trip per norm)
    hipStream_t s = (hipStream_t)stream;
    double* sums = (double*)fused_stats;
    if (!sums) {
      sums = (double*)c->arena.alloc(64 * sizeof(double));
      DOVE_CHECK_ARG(sums, "workspace exhausted (GroupNorm sums)");
      CHK(dove_groupnorm_sums_bf16(x.p, x.elems() / x.C, (long long)x.H * x.W, x.C, gn_scratch(c), c->gn_ws_rows, sums, stream));
    }
    double* msg = (double*)c->arena.alloc(3 * 72 * sizeof(double));
    DOVE_CHECK_ARG(msg, "workspace exhausted (GroupNorm pair message)");
    double *mine = msg, *theirs = msg + 72, *tot = msg + 144;
    hipLaunchKernelGGL(gn_msg_kernel, dim3(1), dim3(128), 0, s, sums, (double)(x.elems() / 32), mine);
    CHK(pair_exchange(c, c->piece_partner, c->piece_lower, mine, theirs, 65 * sizeof(double), stream));
    hipLaunchKernelGGL(gn_add_kernel, dim3(1), dim3(128), 0, s, c->piece_lower ? mine : theirs, c->piece_lower ? theirs : mine, tot);
    c->arena.release(sums);
    stats = (float*)c->arena.alloc(64 * 4);
    CHK(dove_groupnorm_finalize_sums(tot, 0.0, eps, stats, stream));
    c->arena.release(msg);
  } else if (!stats) {
    stats = (float*)c->arena.alloc((size_t)nb * 64 * 4);
    CHK(dove_groupnorm_stats_nb_bf16(x.p, nb, x.elems() / x.C / nb, (long long)x.H * x.W, x.C, eps, gn_scratch(c), c->gn_ws_rows, stats, stream));
  }
  const auto& gb = c->aff.at(name);
  CHK(alloc_t(c, x.T, x.H, x.W, x.C, out));
  if (!zq) {
    CHK(dove_groupnorm_apply_nb_bf16(x.p, out->p, nb, Ti, x.H, x.W, x.C, stats, gb.first, gb.second, 1, nullptr, 0, 0, 0, 0, nullptr, stream));
  } else {
    Tensor yb;
    CHK(conv(c, *zq, c->pc.at(name + ".yb"), ConvOpt(), &yb, stream));
    const int ratio = x.H / zq->H;
    int sshift = 0;
    while ((1 << sshift) < ratio) ++sshift;
    std::vector<int> tmap;
    spatial_norm_tmap(Ti, zq->T / nb, &tmap);
    CHK(dove_groupnorm_apply_nb_bf16(x.p, out->p, nb, Ti, x.H, x.W, x.C, stats, gb.first, gb.second, 1, yb.p, zq->T / nb, yb.H, yb.W, sshift,
                                     tmap.data(), stream));
    free_t(c, yb);
  }
  c->arena.release(stats);
  return 0;
}
// x (+ its fused stats, consumed) -> block output (+ its fused stats).  x is released.
// `tdup`: x came out of a time-doubling Upsample3D (its frames are bit-identical pairs; 1 / 2 = the upsampler's tmode): the per-pixel norm1 keeps
// the pairs, conv1 is told (dove_conv_desc.tdup)
int resnet(dove_ctx* c, Tensor* x, float** xstats, const std::string& name, const Tensor* zq, void* stream, int tdup = 0) {
  const float eps = c->cfg.vae_norm_eps;
  Tensor h1, h, h2, y;
  CHK(norm_silu(c, *x, *xstats, name + ".norm1", zq, &h1, stream));
  float* hs = nullptr;
  ConvOpt o1; o1.gn_eps = eps; o1.gn_stats = &hs; o1.tdup = tdup;
  CHK(cconv(c, h1, true, name + ".conv1", o1, &h, stream));
  CHK(norm_silu(c, h, hs, name + ".norm2", zq, &h2, stream));
  free_t(c, h);
  Tensor sc = *x;
  bool own_sc = false;
  auto it = c->pc.find(name + ".conv_shortcut");
  if (it != c->pc.end()) { CHK(conv(c, *x, it->second, ConvOpt(), &sc, stream)); own_sc = true; }
  float* ys = nullptr;
  ConvOpt o2; o2.resid = sc.p; o2.ldr = sc.C; o2.gn_eps = eps; o2.gn_stats = &ys;
  CHK(cconv(c, h2, true, name + ".conv2", o2, &y, stream));
  if (own_sc) free_t(c, sc);
  free_t(c, *x);
  *x = y; *xstats = ys;
  return 0;
}
void clear_caches(dove_ctx* c);
// Every top-level stage call starts from an empty arena: a stage that returned early on an error (e.g. "workspace exhausted") used to
// leave its blocks live for good - the workspace shrank for every later call and a library-owned arena could never be regrown.
struct StageGuard {
  dove_ctx* c;
  explicit StageGuard(dove_ctx* ctx) : c(ctx) {
    if (c && c->depth++ == 0) {
      clear_caches(c); c->arena.reset(); c->halo_recv = c->halo_send = false; c->nb = 1; set_piece(c, nullptr);
      c->ev_next = 0; c->halo_posted.clear(); c->halo_record.clear(); c->halo_sent = false;
      c->vae_multi = false; c->cache_ev.clear();
    }
  }
  ~StageGuard() { if (c) --c->depth; }
};

int encoder(dove_ctx* c, const Tensor& x, Tensor* out, void* stream) {
  const auto& cf = c->cfg;
  Tensor h; float* hs = nullptr;
  Tensor xin = x;
  CHK(cconv(c, xin, false, c->pc.count("encoder.conv_in.taps") ? "encoder.conv_in.taps" : "encoder.conv_in", ConvOpt(), &h, stream));
  char nm[128];
  int n_tdown = 0;
  for (int r = cf.vae_temporal_compression; r > 1; r >>= 1) ++n_tdown;
  for (int i = 0; i < cf.vae_num_blocks; ++i) {
    for (int j = 0; j < cf.vae_layers_per_block; ++j) { snprintf(nm, sizeof nm, "encoder.down_blocks.%d.resnets.%d", i, j); CHK(resnet(c, &h, &hs, nm, nullptr, stream)); }
    if (i < cf.vae_num_blocks - 1) {
      snprintf(nm, sizeof nm, "encoder.down_blocks.%d.downsamplers.0", i);
      if (hs) { c->arena.release(hs); hs = nullptr; }
      Tensor p = h;
      const int Ti = h.T / c->nb;
      if (i < n_tdown && Ti > 1) {
        const int To = Ti % 2 ? 1 + (Ti - 1) / 2 : Ti / 2;
        CHK(alloc_t(c, c->nb * To, h.H, h.W, h.C, &p));
        CHK(dove_avgpool_time_nb_bf16(h.p, c->nb, Ti, (long long)h.H * h.W * h.C, p.p, stream));
        free_t(c, h);
      }
      ConvOpt od; od.stride = 2; od.pad_h = 0; od.pad_w = 0; od.nb = c->nb;
      Tensor d;
      CHK(conv(c, p, c->pc.at(nm), od, &d, stream));
      free_t(c, p);
      h = d;
    }
  }
  for (int j = 0; j < 2; ++j) { snprintf(nm, sizeof nm, "encoder.mid_block.resnets.%d", j); CHK(resnet(c, &h, &hs, nm, nullptr, stream)); }
  Tensor n;
  CHK(norm_silu(c, h, hs, "encoder.norm_out", nullptr, &n, stream));
  free_t(c, h);
  CHK(cconv(c, n, true, "encoder.conv_out", ConvOpt(), out, stream));
  return 0;
}
int decoder(dove_ctx* c, const Tensor& z, Tensor* out, void* stream) {
  const auto& cf = c->cfg;
  Tensor h; float* hs = nullptr;
  Tensor zin = z;
  CHK(cconv(c, zin, false, "decoder.conv_in", ConvOpt(), &h, stream));
  char nm[128];
  int n_tdown = 0;
  for (int r = cf.vae_temporal_compression; r > 1; r >>= 1) ++n_tdown;
  for (int j = 0; j < 2; ++j) { snprintf(nm, sizeof nm, "decoder.mid_block.resnets.%d", j); CHK(resnet(c, &h, &hs, nm, &z, stream)); }
  int tdup = 0;                                                 // the last upsampler's time map: the next block's first conv reads frame pairs
  for (int i = 0; i < cf.vae_num_blocks; ++i) {
    for (int j = 0; j < cf.vae_layers_per_block + 1; ++j) { snprintf(nm, sizeof nm, "decoder.up_blocks.%d.resnets.%d", i, j); CHK(resnet(c, &h, &hs, nm, &z, stream, j == 0 ? tdup : 0)); }
    if (i < cf.vae_num_blocks - 1) {
      snprintf(nm, sizeof nm, "decoder.up_blocks.%d.upsamplers.0", i);
      if (hs) { c->arena.release(hs); hs = nullptr; }
      ConvOpt ou; ou.up = 1; ou.pad_h = 1; ou.pad_w = 1; ou.gn_eps = cf.vae_norm_eps; ou.gn_stats = &hs; ou.nb = c->nb;
      const int Ti = h.T / c->nb;
      if (i < n_tdown && c->piece_role) {
        // a piece of a split frame-batch: the piece that starts an odd-length batch keeps its first frame single, every other piece doubles
        // all of its frames - also a single-frame piece, which is not a 1-frame batch (diffusers derives the rule from the BATCH's parity)
        ou.tmode = c->piece_role == 1 ? 2 : 1; ou.t_out = c->piece_role == 1 ? 2 * Ti - 1 : 2 * Ti;
      } else if (i < n_tdown && Ti > 1) { ou.tmode = Ti % 2 ? 2 : 1; ou.t_out = Ti % 2 ? 2 * Ti - 1 : 2 * Ti; }
      Tensor u;
      CHK(conv(c, h, c->pc.at(nm), ou, &u, stream));
      free_t(c, h);
      h = u;
      tdup = ou.tmode;
    }
  }
  Tensor n;
  CHK(norm_silu(c, h, hs, "decoder.norm_out", &z, &n, stream));
  free_t(c, h);
  if (c->conv_out_bias) {                                      // tap-split conv_out: fp32 partial planes for dove_conv_out_gather (_cl inside a tile)
    ConvOpt ot; ot.out_f32 = true;
    CHK(cconv(c, n, true, "decoder.conv_out.taps", ot, out, stream));
  } else {
    CHK(cconv(c, n, true, "decoder.conv_out", ConvOpt(), out, stream));
  }
  return 0;
}
void clear_caches(dove_ctx* c) {
  for (auto& kv : c->cache_owner) c->arena.release(kv.second, true);
  c->cache.clear();
  c->cache_owner.clear();
  c->cache_stride.clear();
  c->cache_pair.clear();
}

// ---- DiT (dove_amd/transformer.py) ------------------------------------------------------------------------------------------------
int modulation(dove_ctx* c, int t, const float* temb_in, void* stream) {
  if (c->mod_t == t && !temb_in) return 0;
  const auto& cf = c->cfg;
  const int D = cf.dit_heads * cf.dit_head_dim, te = cf.dit_time_embed_dim;
  hipStream_t s = (hipStream_t)stream;
  if (temb_in) {
    HIPCHK(hipMemcpyAsync(c->temb, temb_in, (size_t)D * 4, hipMemcpyDeviceToDevice, s));
  } else {
    // Timesteps(): [sin | cos] of t * exp(-ln(1e4) i / (half - shift)), flipped to [cos | sin], cast to the model dtype (bf16)
    std::vector<float> h(D);
    const int half = D / 2;
    for (int i = 0; i < half; ++i) {
      const float ang = (float)t * expf(-logf(10000.0f) * (float)i / ((float)half - cf.dit_freq_shift));
      const float sn = sinf(ang), cs = cosf(ang);
      if (cf.dit_flip_sin_to_cos) { h[i] = cs; h[half + i] = sn; } else { h[i] = sn; h[half + i] = cs; }
    }
    for (auto& v : h) { uint32_t u; memcpy(&u, &v, 4); u += 0x7fffu + ((u >> 16) & 1u); u &= 0xffff0000u; memcpy(&v, &u, 4); }
    HIPCHK(hipMemcpyAsync(c->temb, h.data(), (size_t)D * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));                          // h goes out of scope
  }
  CHK(dove_gemv_bf16(c->te1_w, c->te1_b, c->temb, D, te, 0, c->e1, stream));
  CHK(dove_gemv_bf16(c->te2_w, c->te2_b, c->e1, te, te, 1, c->emb, stream));
  auto regroup = [&](const float* v, float* m, float* g, int i_sh, int i_sc, int i_g, int e_sh, int e_sc, int e_g) -> int {
    // mod[class][shift|scale][D] (class 0 = text rows), gate[class][D]
    const int idx[4] = {e_sh, e_sc, i_sh, i_sc};
    for (int q = 0; q < 4; ++q) HIPCHK(hipMemcpyAsync(m + (size_t)q * D, v + (size_t)idx[q] * D, (size_t)D * 4, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(g, v + (size_t)e_g * D, (size_t)D * 4, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(g + D, v + (size_t)i_g * D, (size_t)D * 4, hipMemcpyDeviceToDevice, s));
    return 0;
  };
  for (auto& b : c->blocks) {
    CHK(dove_gemv_bf16(b.mod1_w, b.mod1_b, c->emb, te, 6 * D, 1, c->vtmp, stream));
    CHK(regroup(c->vtmp, b.m1, b.g1, 0, 1, 2, 3, 4, 5));
    CHK(dove_gemv_bf16(b.mod2_w, b.mod2_b, c->emb, te, 6 * D, 1, c->vtmp, stream));
    CHK(regroup(c->vtmp, b.m2, b.g2, 0, 1, 2, 3, 4, 5));
  }
  CHK(dove_gemv_bf16(c->modout_w, c->modout_b, c->emb, te, 2 * D, 1, c->vtmp, stream));      // (shift, scale)
  for (int cls = 0; cls < 2; ++cls) HIPCHK(hipMemcpyAsync(c->final_mod + (size_t)cls * 2 * D, c->vtmp, (size_t)2 * D * 4, hipMemcpyDeviceToDevice, s));
  c->mod_t = temb_in ? -1 : t;
  return 0;
}
// get_3d_rotary_pos_embed(grid_type="slice"): [t | h | w] split 16/24/24 of head_dim 64, theta 1e4, each frequency twice (dove_amd/rope.py)
int rope_tables(dove_ctx* c, int gt, int gh, int gw, const float** cosp, const float** sinp, void* stream) {
  const long long nv = (long long)gt * gh * gw;
  const int hd = c->cfg.dit_head_dim;
  if (!(c->rope_dev && c->rope_t == gt && c->rope_h == gh && c->rope_w == gw)) {
    const int dt = hd / 4, dh = hd / 8 * 3, dw = dh;
    std::vector<float> tab((size_t)2 * nv * hd);
    auto fill = [&](int dim, int n, std::vector<float>& cs, std::vector<float>& sn) {
      cs.resize((size_t)n * dim); sn.resize((size_t)n * dim);
      for (int i = 0; i < dim / 2; ++i) {
        const float f = 1.0f / powf(10000.0f, (float)(2 * i) / (float)dim);
        for (int p = 0; p < n; ++p) { const float a = (float)p * f; cs[(size_t)p * dim + 2 * i] = cs[(size_t)p * dim + 2 * i + 1] = cosf(a); sn[(size_t)p * dim + 2 * i] = sn[(size_t)p * dim + 2 * i + 1] = sinf(a); }
      }
    };
    std::vector<float> ct, st, ch, sh, cw, sw;
    fill(dt, gt, ct, st); fill(dh, gh, ch, sh); fill(dw, gw, cw, sw);
    for (int a = 0; a < gt; ++a) for (int b = 0; b < gh; ++b) for (int d = 0; d < gw; ++d) {
      float* co = &tab[(((size_t)a * gh + b) * gw + d) * hd];
      float* so = co + (size_t)nv * hd;
      memcpy(co, &ct[(size_t)a * dt], dt * 4); memcpy(co + dt, &ch[(size_t)b * dh], dh * 4); memcpy(co + dt + dh, &cw[(size_t)d * dw], dw * 4);
      memcpy(so, &st[(size_t)a * dt], dt * 4); memcpy(so + dt, &sh[(size_t)b * dh], dh * 4); memcpy(so + dt + dh, &sw[(size_t)d * dw], dw * 4);
    }
    if (c->rope_dev) (void)hipFree(c->rope_dev);
    HIPCHK(hipMalloc((void**)&c->rope_dev, tab.size() * 4));
    HIPCHK(hipMemcpy(c->rope_dev, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    c->rope_t = gt; c->rope_h = gh; c->rope_w = gw;
  }
  *cosp = c->rope_dev; *sinp = c->rope_dev + nv * hd;
  return 0;
}

}  // namespace

// ==================================================================================================================================
extern "C" int dove_create(int device, const dove_model_config* cfg, dove_ctx** out) {
  DOVE_CHECK_ARG(cfg && out, "dove_create: null pointer");
  DOVE_CHECK_ARG(cfg->dit_head_dim == 64, "dove_create: head_dim must be 64");
  DOVE_CHECK_ARG(cfg->vae_num_blocks >= 1 && cfg->vae_num_blocks <= 8 && cfg->dit_patch_t >= 1 && cfg->dit_patch >= 1, "dove_create: bad config");
  HIPCHK(hipSetDevice(device));
  dove_ctx* c = new dove_ctx();
  c->cfg = *cfg;
  c->device = device;
  *out = c;
  return DOVE_OK;
}
extern "C" void dove_comm_destroy(dove_ctx* c);
extern "C" void dove_destroy(dove_ctx* c) {
  if (!c) return;
  dove_comm_destroy(c);
  clear_caches(c);
  if (c->vae_stream2) { (void)hipStreamSynchronize(c->vae_stream2); (void)hipStreamDestroy(c->vae_stream2); }
  for (hipEvent_t e : c->arena.pool) (void)hipEventDestroy(e);
  for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
  for (void* p : c->owned) (void)hipFree(p);
  if (c->rope_dev) (void)hipFree(c->rope_dev);
  if (c->Qh) { (void)hipFree(c->Qh); (void)hipFree(c->Kh); (void)hipFree(c->Vt); }
  if (c->Q8) { (void)hipFree(c->Q8); (void)hipFree(c->K8); (void)hipFree(c->V8); (void)hipFree(c->Vs8); }
  if (c->arena.owned && c->arena.base) (void)hipFree(c->arena.base);
  delete c;
}
// ---- multi-GPU: halo exchange between the ranks of one clip (SURVEY.md 8(b) dove_comm_init, 8(e)) ---------------------------------
// The transport is two function pointers; dove_comm_init binds them to RCCL (library opened with dlopen so that libdove_hip.so itself does
// not depend on it), dove_comm_init_custom to anything else (tests: an in-process mailbox between contexts on one GPU).  The symmetric
// exchanges (GroupNorm pair sums, the DiT's all-to-alls, the moments gather) run on the caller's stream - their results are needed at once -
// as one ncclGroup each; the VAE's halos run on the context's OWN receive / send streams and, under RCCL, on two more communicators, so that
// the caller's stream never executes a transfer and a rank's pre-posted receives cannot hold its sends back (dove_ctx, struct Rccl).
namespace {
struct Rccl {
  void* lib; void* comm;
  // link[i]: communicator of the neighbour pairs (r - 1, r) with r % 2 == i - a rank receives its halos (from rank - 1) on link[rank % 2] and
  // sends (to rank + 1) on link[(rank + 1) % 2]: never both on one communicator, whose point-to-point operations RCCL runs in issue order
  // (pre-posted receives would hold every send back until the rank's LAST halo had arrived).  Made with ncclCommSplit; without that symbol
  // both entries are `comm` and nothing is pre-posted.
  void* link[2]; int rank;
  int (*Send)(const void*, size_t, int, int, void*, hipStream_t);
  int (*Recv)(void*, size_t, int, int, void*, hipStream_t);
  int (*CommDestroy)(void*);
  int (*GroupStart)();
  int (*GroupEnd)();
};
int rccl_group_begin(void* user, void*) {
  const int rc = ((Rccl*)user)->GroupStart();
  if (rc) { dove_set_error("ncclGroupStart failed (%d)", rc); return DOVE_ELAUNCH; }
  return 0;
}
int rccl_group_end(void* user, void*) {
  const int rc = ((Rccl*)user)->GroupEnd();
  if (rc) { dove_set_error("ncclGroupEnd failed (%d)", rc); return DOVE_ELAUNCH; }
  return 0;
}
struct UniqueId { char b[128]; };   // ncclUniqueId: a 128-byte struct, passed BY VALUE to ncclCommInitRank
int rccl_send(void* user, int peer, void* p, size_t bytes, void* stream) {
  Rccl* r = (Rccl*)user;
  const int rc = r->Send(p, bytes, /*ncclChar*/ 0, peer, r->comm, (hipStream_t)stream);
  if (rc) { dove_set_error("ncclSend to rank %d failed (%d)", peer, rc); return DOVE_ELAUNCH; }
  return 0;
}
int rccl_recv(void* user, int peer, void* p, size_t bytes, void* stream) {
  Rccl* r = (Rccl*)user;
  const int rc = r->Recv(p, bytes, /*ncclChar*/ 0, peer, r->comm, (hipStream_t)stream);
  if (rc) { dove_set_error("ncclRecv from rank %d failed (%d)", peer, rc); return DOVE_ELAUNCH; }
  return 0;
}
int rccl_halo_send(void* user, int peer, void* p, size_t bytes, void* stream) {
  Rccl* r = (Rccl*)user;
  const int rc = r->Send(p, bytes, /*ncclChar*/ 0, peer, r->link[(r->rank + 1) & 1], (hipStream_t)stream);
  if (rc) { dove_set_error("ncclSend (halo) to rank %d failed (%d)", peer, rc); return DOVE_ELAUNCH; }
  return 0;
}
int rccl_halo_recv(void* user, int peer, void* p, size_t bytes, void* stream) {
  Rccl* r = (Rccl*)user;
  const int rc = r->Recv(p, bytes, /*ncclChar*/ 0, peer, r->link[r->rank & 1], (hipStream_t)stream);
  if (rc) { dove_set_error("ncclRecv (halo) from rank %d failed (%d)", peer, rc); return DOVE_ELAUNCH; }
  return 0;
}
void* open_rccl() {
  void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  return h;
}
}  // namespace

extern "C" int dove_comm_unique_id(void* out128) {
  DOVE_CHECK_ARG(out128, "dove_comm_unique_id: null pointer");
  void* h = open_rccl();
  DOVE_CHECK_ARG(h, "dove_comm_unique_id: librccl.so not found (%s)", dlerror());
  auto get = (int (*)(UniqueId*))dlsym(h, "ncclGetUniqueId");
  DOVE_CHECK_ARG(get, "dove_comm_unique_id: ncclGetUniqueId not found");
  const int rc = get((UniqueId*)out128);
  DOVE_CHECK_ARG(rc == 0, "ncclGetUniqueId failed (%d)", rc);
  return DOVE_OK;
}
extern "C" void dove_comm_destroy(dove_ctx* c) {
  if (!c) return;
  if (c->rccl_comm) {
    Rccl* r = (Rccl*)c->rccl_comm;
    for (int i = 0; i < 2; ++i) if (r->link[i] && r->link[i] != r->comm) (void)r->CommDestroy(r->link[i]);
    if (r->comm) (void)r->CommDestroy(r->comm);
    delete r;
    c->rccl_comm = nullptr;
  }
  if (c->recv_stream) { (void)hipStreamSynchronize(c->recv_stream); (void)hipStreamDestroy(c->recv_stream); c->recv_stream = nullptr; }
  if (c->send_stream) { (void)hipStreamSynchronize(c->send_stream); (void)hipStreamDestroy(c->send_stream); c->send_stream = nullptr; }
  c->halo_plans.clear(); c->halo_posted.clear(); c->halo_record.clear(); c->halo_can_prepost = true;
  c->halo_send_fn = c->halo_recv_fn = nullptr;
  c->rank = 0; c->nranks = 1; c->send_fn = c->recv_fn = nullptr; c->xfer_user = nullptr; c->group_begin = c->group_end = nullptr;
}
extern "C" int dove_comm_set_group(dove_ctx* c, dove_group_fn begin, dove_group_fn end) {
  DOVE_CHECK_ARG(c && ((begin == nullptr) == (end == nullptr)), "dove_comm_set_group: give both callbacks or neither");
  c->group_begin = begin; c->group_end = end;
  return DOVE_OK;
}
extern "C" int dove_comm_init_custom(dove_ctx* c, int rank, int nranks, dove_xfer_fn send, dove_xfer_fn recv, void* user) {
  DOVE_CHECK_ARG(c && nranks >= 1 && rank >= 0 && rank < nranks, "dove_comm_init_custom: bad rank %d of %d", rank, nranks);
  DOVE_CHECK_ARG(nranks == 1 || (send && recv), "dove_comm_init_custom: send / recv callbacks are required");
  dove_comm_destroy(c);
  c->rank = rank; c->nranks = nranks; c->send_fn = send; c->recv_fn = recv; c->xfer_user = user;
  c->halo_send_fn = send; c->halo_recv_fn = recv;             // halos: the same callbacks, handed the context's send / receive stream
  return DOVE_OK;
}
extern "C" int dove_comm_init(dove_ctx* c, const void* nccl_unique_id, int rank, int nranks) {
  DOVE_CHECK_ARG(c && nccl_unique_id && nranks >= 1 && rank >= 0 && rank < nranks, "dove_comm_init: bad argument");
  void* h = open_rccl();
  DOVE_CHECK_ARG(h, "dove_comm_init: librccl.so not found (%s)", dlerror());
  Rccl* r = new Rccl();
  r->lib = h; r->comm = nullptr;
  auto init = (int (*)(void**, int, UniqueId, int))dlsym(h, "ncclCommInitRank");
  r->Send = (decltype(r->Send))dlsym(h, "ncclSend");
  r->Recv = (decltype(r->Recv))dlsym(h, "ncclRecv");
  r->CommDestroy = (decltype(r->CommDestroy))dlsym(h, "ncclCommDestroy");
  r->GroupStart = (decltype(r->GroupStart))dlsym(h, "ncclGroupStart");
  r->GroupEnd = (decltype(r->GroupEnd))dlsym(h, "ncclGroupEnd");
  if (!init || !r->Send || !r->Recv || !r->CommDestroy || !r->GroupStart || !r->GroupEnd) { delete r; dove_set_error("dove_comm_init: RCCL symbols not found"); return DOVE_EINVAL; }
  HIPCHK(hipSetDevice(c->device));
  UniqueId id;
  memcpy(&id, nccl_unique_id, sizeof id);
  const int rc = init(&r->comm, nranks, id, rank);
  if (rc) { delete r; dove_set_error("ncclCommInitRank failed (%d)", rc); return DOVE_ELAUNCH; }
  // two more communicators over the same ranks for the halo wavefront (struct Rccl): ncclCommSplit with one colour = a duplicate, collective
  r->rank = rank; r->link[0] = r->link[1] = r->comm;
  bool split_ok = false;
  auto split = (int (*)(void*, int, int, void**, void*))dlsym(h, "ncclCommSplit");
  if (split && nranks > 1) {
    void *a = nullptr, *b = nullptr;
    const int ra = split(r->comm, 0, rank, &a, nullptr);
    const int rb = ra ? ra : split(r->comm, 0, rank, &b, nullptr);
    if (!ra && !rb && a && b) { r->link[0] = a; r->link[1] = b; split_ok = true; }
    else {
      if (a) (void)r->CommDestroy(a);
      if (b) (void)r->CommDestroy(b);
    }
  }
  dove_comm_destroy(c);
  c->rccl_comm = r;
  c->rank = rank; c->nranks = nranks; c->send_fn = rccl_send; c->recv_fn = rccl_recv; c->xfer_user = r;
  c->halo_send_fn = rccl_halo_send; c->halo_recv_fn = rccl_halo_recv;
  c->halo_can_prepost = split_ok || nranks == 1;
  c->group_begin = rccl_group_begin; c->group_end = rccl_group_end;      // every exchange (pair swap, all-to-all) is one ncclGroup
  return DOVE_OK;
}
// frames [first, first + count) of a stage's output that THIS rank produces (stage 0: dove_vae_encode, n = F pixel frames ->
// latent frames; stage 1: dove_vae_decode, n = T latent frames -> pixel frames); the other frames of the output buffer are
// left untouched and are the other ranks' to deliver
extern "C" int dove_shard_frames(dove_ctx* c, int stage, int n, int* first, int* count) {
  DOVE_CHECK_ARG(c && first && count && n >= 1 && (stage == 0 || stage == 1), "dove_shard_frames: bad argument");
  RankPlan rp;
  rank_plan(c, stage == 0, n, &rp);
  *first = rp.rank_first[c->rank];
  *count = rp.rank_count[c->rank];
  return DOVE_OK;
}

extern "C" int dove_comm_useful_ranks(dove_ctx* c, int stage, int n) {
  if (!c || n < 1 || (stage != 0 && stage != 1)) return 0;
  // ranks that get work when `nranks` ranks share the stage: up to two per frame-batch (batches are split into paired pieces when there
  // are more ranks than batches); asked on a single-rank context: the most ranks that could
  std::vector<std::pair<int, int>> fb;
  frame_batches(n, stage == 0 ? c->cfg.vae_enc_batch : c->cfg.vae_dec_batch, &fb);
  std::vector<std::vector<Piece>> plan;
  return plan_pieces(fb, c->nranks > 1 ? c->nranks : 2 * (int)fb.size(), stage == 0, &plan);
}

extern "C" int dove_set_option(dove_ctx* c, int option, long long value) {
  DOVE_CHECK_ARG(c, "dove_set_option: null context");
  switch (option) {
    case DOVE_OPT_VAE_TILING: c->opt_tiling = value != 0; return DOVE_OK;
    case DOVE_OPT_VAE_SAMPLE_HEIGHT:
    case DOVE_OPT_VAE_SAMPLE_WIDTH:
      DOVE_CHECK_ARG(value >= 32 && value % 16 == 0 && value <= (1 << 16), "dove_set_option: sample size %lld must be a multiple of 16 in [32, 65536]", value);
      (option == DOVE_OPT_VAE_SAMPLE_HEIGHT ? c->sample_h : c->sample_w) = (int)value;
      return DOVE_OK;
    case DOVE_OPT_DIT_LINEAR_MXFP8:
      DOVE_CHECK_ARG(!c->finalized || (value != 0) == c->opt_linear_mx,
                     "dove_set_option: DOVE_OPT_DIT_LINEAR_MXFP8 must be chosen before dove_finalize_weights (the weights are quantised there)");
      c->opt_linear_mx = value != 0; return DOVE_OK;
    case DOVE_OPT_DIT_ATTN_MXFP8: c->opt_attn_mx = value != 0; return DOVE_OK;
    case DOVE_OPT_WEIGHT_SUMS: c->opt_weight_sums = value != 0; return DOVE_OK;
    case DOVE_OPT_VAE_STREAMS:
      DOVE_CHECK_ARG(value == 1 || value == 2, "dove_set_option: DOVE_OPT_VAE_STREAMS is 1 or 2");
      c->opt_vae_streams = (int)value; return DOVE_OK;
    default: break;
  }
  dove_set_error("dove_set_option: unknown option %d", option);
  return DOVE_EINVAL;
}
extern "C" long long dove_get_option(dove_ctx* c, int option) {
  if (!c) return -1;
  switch (option) {
    case DOVE_OPT_VAE_TILING: return c->opt_tiling;
    case DOVE_OPT_VAE_SAMPLE_HEIGHT: return c->sample_h;
    case DOVE_OPT_VAE_SAMPLE_WIDTH: return c->sample_w;
    case DOVE_OPT_DIT_LINEAR_MXFP8: return c->opt_linear_mx;
    case DOVE_OPT_DIT_ATTN_MXFP8: return c->opt_attn_mx;
    case DOVE_OPT_WEIGHT_SUMS: return c->opt_weight_sums;
    case DOVE_OPT_VAE_STREAMS: return c->opt_vae_streams;
    case DOVE_STAT_HALO_PREPOSTED: return c->stat_halo_preposted;
    case DOVE_STAT_HALO_BLOCKING: return c->stat_halo_blocking;
    case DOVE_STAT_HALO_SENT: return c->stat_halo_sent;
    case DOVE_STAT_HALO_COMMUNICATORS: return c->nranks > 1 ? (c->halo_can_prepost && c->rccl_comm ? 3 : (c->rccl_comm ? 1 : 0)) : 0;
    default: return -1;
  }
}
extern "C" int dove_set_weight(dove_ctx* c, const char* name, const void* dev_ptr, const long long* shape, int ndim, int dtype) {
  DOVE_CHECK_ARG(c && name && dev_ptr && shape && ndim >= 1 && ndim <= 5, "dove_set_weight: bad argument");
  DOVE_CHECK_ARG(dtype == DOVE_F32 || dtype == DOVE_BF16, "dove_set_weight: dtype must be DOVE_F32 or DOVE_BF16");
  DOVE_CHECK_ARG(!c->finalized, "dove_set_weight: weights are already finalized");
  Raw r; r.p = dev_ptr; r.dt = dtype; r.shape.assign(shape, shape + ndim);
  c->raw[name] = r;
  return DOVE_OK;
}
extern "C" int dove_finalize_weights(dove_ctx* c) {
  DOVE_CHECK_ARG(c && !c->finalized, "dove_finalize_weights: bad context");
  HIPCHK(hipSetDevice(c->device));
  const auto& cf = c->cfg;
  // ---- VAE: every conv by its diffusers module path (dove_amd/vae.py _pack) ----
  std::vector<std::string> names;
  for (auto& kv : c->raw) names.push_back(kv.first);
  for (auto& k : names) {
    if (k.compare(0, 8, "encoder.") && k.compare(0, 8, "decoder.")) continue;
    if (ends_with(k, ".conv.weight") && k.find(".conv_y.") == std::string::npos && k.find(".conv_b.") == std::string::npos) {
      const std::string n = k.substr(0, k.size() - strlen(".conv.weight"));
      // sub-pixel sums only for the upsamplers; pair sums for the first causal conv behind a TIME-doubling upsampler (decoder up-block
      // i > 0 whose predecessor compresses time: dove_amd/vae.py _pack)
      int ub = -1, n_td = 0;
      for (int r = cf.vae_temporal_compression; r > 1; r >>= 1) ++n_td;
      const bool pair = sscanf(n.c_str(), "decoder.up_blocks.%d.resnets.0.conv1", &ub) == 1 && ends_with(n, ".resnets.0.conv1") && ub > 0 && ub <= n_td;
      Packed p; CHK(pack(c, {k}, {n + ".conv.bias"}, &p, n.find(".upsamplers.") != std::string::npos, pair)); c->pc[n] = p;
    } else if (ends_with(k, ".conv_shortcut.weight")) {
      const std::string n = k.substr(0, k.size() - strlen(".weight"));
      Packed p; CHK(pack(c, {k}, {n + ".bias"}, &p)); c->pc[n] = p;
    } else if (ends_with(k, ".conv_y.conv.weight")) {
      const std::string n = k.substr(0, k.size() - strlen(".conv_y.conv.weight"));
      Packed p; CHK(pack(c, {k, n + ".conv_b.conv.weight"}, {n + ".conv_y.conv.bias", n + ".conv_b.conv.bias"}, &p)); c->pc[n + ".yb"] = p;
      float *g, *b; CHK(to_f32(c, n + ".norm_layer.weight", &g)); CHK(to_f32(c, n + ".norm_layer.bias", &b)); c->aff[n] = {g, b};
    } else if (ends_with(k, ".weight") && c->raw[k].shape.size() == 1 && k.find(".norm_layer.") == std::string::npos) {
      const std::string n = k.substr(0, k.size() - strlen(".weight"));
      float *g, *b; CHK(to_f32(c, k, &g)); CHK(to_f32(c, n + ".bias", &b)); c->aff[n] = {g, b};
    }
  }
  DOVE_CHECK_ARG(c->pc.count("encoder.conv_in") && c->pc.count("decoder.conv_out"), "dove_finalize_weights: VAE weights missing");
  {
    // encoder.conv_in with the 3x3 spatial taps unrolled into the input channels (dove_amd/vae.py _pack)
    const Raw* w; CHK(need(c, "encoder.conv_in.conv.weight", &w, 5));
    const int cout = (int)w->shape[0], C = (int)w->shape[1], kt = (int)w->shape[2];
    if (w->shape[3] == 3 && w->shape[4] == 3 && 9 * C <= 32) {
      Packed q; q.kt = kt; q.kh = 1; q.kw = 1; q.cin = 9 * C; q.cin_pad = 32; q.cout = cout; q.cout_pad = (int)ru(cout, 32);
      const size_t wbytes = (size_t)kt * q.cout_pad * 32 * 2;
      void* wp; CHK(dev_alloc(c, wbytes, &wp));
      HIPCHK(hipMemsetAsync(wp, 0, wbytes, 0));
      hipLaunchKernelGGL(pack_in_taps_kernel, dim3(256), dim3(256), 0, 0, w->p, w->dt, cout, C, kt, q.cout_pad, 32, (bf16_t*)wp);
      q.w = (bf16_t*)wp;
      q.bias = c->pc.at("encoder.conv_in").bias;                  // same bias vector (already padded to cout_pad)
      c->pc["encoder.conv_in.taps"] = q;
    }
  }
  {
    // decoder.conv_out split by spatial tap (dove_amd/vae.py _pack; include/dove_hip.h dove_conv_out_gather)
    const Raw* w; CHK(need(c, "decoder.conv_out.conv.weight", &w, 5));
    const int C = (int)w->shape[0], cin = (int)w->shape[1], kt = (int)w->shape[2];
    if (w->shape[3] == 3 && w->shape[4] == 3 && 9 * C <= 32) {
      Packed q; q.kt = kt; q.kh = 1; q.kw = 1; q.cin = cin; q.cin_pad = (cin <= 32 || cin % 64) ? (int)ru(cin, 32) : cin; q.cout = 9 * C; q.cout_pad = 32;
      const size_t wbytes = (size_t)kt * 32 * q.cin_pad * 2;
      void* wp; CHK(dev_alloc(c, wbytes, &wp));
      HIPCHK(hipMemsetAsync(wp, 0, wbytes, 0));
      hipLaunchKernelGGL(pack_taps_kernel, dim3(256), dim3(256), 0, 0, w->p, w->dt, C, cin, kt, q.cin_pad, (bf16_t*)wp);
      q.w = (bf16_t*)wp;
      c->pc["decoder.conv_out.taps"] = q;
      CHK(to_f32(c, "decoder.conv_out.conv.bias", &c->conv_out_bias));
    }
  }
  // ---- DiT (dove_amd/transformer.py _pack) ----
  const int D = cf.dit_heads * cf.dit_head_dim;
  CHK(pack(c, {"patch_embed.proj.weight"}, {"patch_embed.proj.bias"}, &c->pe_proj));
  CHK(pack(c, {"patch_embed.text_proj.weight"}, {"patch_embed.text_proj.bias"}, &c->pe_text));
  CHK(to_bf16(c, "time_embedding.linear_1.weight", &c->te1_w)); CHK(to_f32(c, "time_embedding.linear_1.bias", &c->te1_b));
  CHK(to_bf16(c, "time_embedding.linear_2.weight", &c->te2_w)); CHK(to_f32(c, "time_embedding.linear_2.bias", &c->te2_b));
  c->blocks.resize(cf.dit_num_layers);
  char b[96];
  for (int i = 0; i < cf.dit_num_layers; ++i) {
    DitBlock& k = c->blocks[i];
    snprintf(b, sizeof b, "transformer_blocks.%d.", i);
    const std::string p(b);
    CHK(to_bf16(c, p + "norm1.linear.weight", &k.mod1_w)); CHK(to_f32(c, p + "norm1.linear.bias", &k.mod1_b, 6 * D));
    CHK(to_bf16(c, p + "norm2.linear.weight", &k.mod2_w)); CHK(to_f32(c, p + "norm2.linear.bias", &k.mod2_b, 6 * D));
    CHK(to_f32(c, p + "norm1.norm.weight", &k.ln1_g, D)); CHK(to_f32(c, p + "norm1.norm.bias", &k.ln1_b, D));
    CHK(to_f32(c, p + "norm2.norm.weight", &k.ln2_g, D)); CHK(to_f32(c, p + "norm2.norm.bias", &k.ln2_b, D));
    CHK(to_f32(c, p + "attn1.norm_q.weight", &k.nq_g, 64)); CHK(to_f32(c, p + "attn1.norm_q.bias", &k.nq_b, 64));
    CHK(to_f32(c, p + "attn1.norm_k.weight", &k.nk_g, 64)); CHK(to_f32(c, p + "attn1.norm_k.bias", &k.nk_b, 64));
    CHK(pack(c, {p + "attn1.to_q.weight", p + "attn1.to_k.weight", p + "attn1.to_v.weight"}, {p + "attn1.to_q.bias", p + "attn1.to_k.bias", p + "attn1.to_v.bias"}, &k.qkv));
    CHK(pack(c, {p + "attn1.to_out.0.weight"}, {p + "attn1.to_out.0.bias"}, &k.out));
    CHK(pack(c, {p + "ff.net.0.proj.weight"}, {p + "ff.net.0.proj.bias"}, &k.ff1));
    CHK(pack(c, {p + "ff.net.2.weight"}, {p + "ff.net.2.bias"}, &k.ff2));
    if (c->opt_linear_mx) {          // dove_amd.ops.pack_linear_mx: the bf16-rounded weight quantised on the GPU; the bf16 copy is dropped
      CHK(to_mx(c, &k.qkv, &k.qkv8)); CHK(to_mx(c, &k.out, &k.out8)); CHK(to_mx(c, &k.ff1, &k.ff18)); CHK(to_mx(c, &k.ff2, &k.ff28));
    }
    void* m;
    CHK(dev_alloc(c, (size_t)4 * D * 4, &m)); k.m1 = (float*)m; CHK(dev_alloc(c, (size_t)2 * D * 4, &m)); k.g1 = (float*)m;
    CHK(dev_alloc(c, (size_t)4 * D * 4, &m)); k.m2 = (float*)m; CHK(dev_alloc(c, (size_t)2 * D * 4, &m)); k.g2 = (float*)m;
  }
  CHK(to_f32(c, "norm_final.weight", &c->nf_g, D)); CHK(to_f32(c, "norm_final.bias", &c->nf_b, D));
  CHK(to_bf16(c, "norm_out.linear.weight", &c->modout_w)); CHK(to_f32(c, "norm_out.linear.bias", &c->modout_b, 2 * D));
  CHK(to_f32(c, "norm_out.norm.weight", &c->no_g, D)); CHK(to_f32(c, "norm_out.norm.bias", &c->no_b, D));
  CHK(pack(c, {"proj_out.weight"}, {"proj_out.bias"}, &c->proj_out));
  void* m;
  CHK(dev_alloc(c, (size_t)4 * D * 4, &m)); c->final_mod = (float*)m;
  CHK(dev_alloc(c, (size_t)cf.dit_time_embed_dim * 4, &m)); c->emb = (float*)m;
  CHK(dev_alloc(c, (size_t)cf.dit_time_embed_dim * 4, &m)); c->e1 = (float*)m;
  CHK(dev_alloc(c, (size_t)D * 4, &m)); c->temb = (float*)m;
  CHK(dev_alloc(c, (size_t)6 * D * 4, &m)); c->vtmp = (float*)m;
  c->gn_ws_rows = 65536;                                 // 16 MiB: nb x frames x 256 rows of a tile batch's statistics pass
  CHK(dev_alloc(c, (size_t)c->gn_ws_rows * 64 * 4, &m)); c->gn_ws = (float*)m;
  HIPCHK(hipDeviceSynchronize());                             // the borrowed source tensors may be released by the caller now
  c->raw.clear();
  c->finalized = true;
  return DOVE_OK;
}

// Upper bound of the arena a whole dove_sr_clip needs for a [3, F, H, W] clip: the live set of the widest VAE stage
// (decoder up-block with C0 channels at full resolution: block input + normalised copy + conv output + shortcut, one frame-batch)
// plus the DiT's token buffers.
extern "C" size_t dove_workspace_bytes(dove_ctx* c, int F, int H, int W) {
  if (!c) return 0;
  const auto& cf = c->cfg;
  const int T = 1 + (F - 1) / cf.vae_temporal_compression;
  const long long fb = cf.vae_enc_batch + 1;                                      // frames of the largest frame-batch
  const long long top = fb * H * W * (long long)cf.vae_block_out_channels[0] * 2;      // one full-resolution C0 tensor, bf16
  // live set of the last up block (2*C0-channel input + its normalised copy + conv output) ~ 5 x top, every causal conv's 2-frame
  // cache ~ 3 x top over the decoder, staging of one decoded frame-batch; x 1.5 for first-fit fragmentation
  // plus, memory permitting, every causal conv's input of the previous frame-batch retained instead of copied (~ 20 x top)
  long long vae = 22 * top;                                                       // measured high water at 33x720x1280: 17.4 x top
  // enable_tiling(): the tiles of one shape run as one batch; overlapping tiles cover up to 6/5 x 5/4 = 1.5 x the frame
  if (c->opt_tiling) vae = vae * 3 / 2;
  else if (c->opt_vae_streams >= 2 && c->nranks == 1) vae = vae * 17 / 10;          // two frame-batches in flight (measured high water at 33x720x1280: see test_sr_clip_full_size_timing)
  const int D = cf.dit_heads * cf.dit_head_dim;
  const long long Td = T + (T % cf.dit_patch_t);
  const long long N = cf.dit_max_text + (Td / cf.dit_patch_t) * (H / 8 / cf.dit_patch) * (W / 8 / cf.dit_patch);
  long long dit = N * (long long)D * 2 * (1 + 1 + 3 + 4 + 1) + (1ll << 24);       // hs, n1, qkv, f1, slack
  long long io = 4ll * F * H * W * 2 * 2 + (long long)T * (H / 8) * (W / 8) * 64 * 4 * 4;
  return (size_t)((vae > dit ? vae : dit) + io + (64ll << 20));
}
extern "C" int dove_set_workspace(dove_ctx* c, void* dev_ptr, size_t bytes) {
  DOVE_CHECK_ARG(c, "dove_set_workspace: null context");
  if (c->arena.owned && c->arena.base) (void)hipFree(c->arena.base);
  {
    std::vector<hipEvent_t> pool = std::move(c->arena.pool);    // the event pool outlives the workspace
    c->arena = Arena();
    c->arena.pool = std::move(pool);
  }
  if (dev_ptr) { c->arena.base = (char*)dev_ptr; c->arena.cap = bytes; c->arena.owned = false; }
  else { void* p; HIPCHK(hipMalloc(&p, bytes)); c->arena.base = (char*)p; c->arena.cap = bytes; c->arena.owned = true; }
  c->arena.reset();
  return DOVE_OK;
}
extern "C" size_t dove_workspace_high_water(dove_ctx* c) { return c ? c->arena.high : 0; }


// ---- diffusers spatial tiling (AutoencoderKLCogVideoX.enable_tiling / tiled_encode / tiled_decode; ref :643-645 `--is_vae_st`;
// dove_amd/vae.py _tiled): every tile runs the whole frame-batched network with its own conv caches and its own GroupNorm scope,
// tiles are cross-faded IN PLACE with their already blended upper / left neighbours, cropped and concatenated. ----
struct TileGeom { int tile_h, tile_w, stride_h, stride_w, blend_h, blend_w, lim_h, lim_w; };
static void tiling_geometry(const dove_ctx* c, bool enc, TileGeom* g) {
  const int down = 1 << (c->cfg.vae_num_blocks - 1);
  const int smin_h = c->sample_h / 2, smin_w = c->sample_w / 2;
  const int lmin_h = (int)((double)smin_h / down), lmin_w = (int)((double)smin_w / down);
  const double of_h = 1.0 / 6.0, of_w = 1.0 / 5.0;                       // tile_overlap_factor_height / _width
  if (enc) {
    g->tile_h = smin_h; g->tile_w = smin_w;
    g->stride_h = (int)(smin_h * (1.0 - of_h)); g->stride_w = (int)(smin_w * (1.0 - of_w));
    g->blend_h = (int)(lmin_h * of_h); g->blend_w = (int)(lmin_w * of_w);
    g->lim_h = lmin_h - g->blend_h; g->lim_w = lmin_w - g->blend_w;
  } else {
    g->tile_h = lmin_h; g->tile_w = lmin_w;
    g->stride_h = (int)(lmin_h * (1.0 - of_h)); g->stride_w = (int)(lmin_w * (1.0 - of_w));
    g->blend_h = (int)(smin_h * of_h); g->blend_w = (int)(smin_w * of_w);
    g->lim_h = smin_h - g->blend_h; g->lim_w = smin_w - g->blend_w;
  }
}
static bool wants_tiling(const dove_ctx* c, bool enc, int H, int W) {
  if (!c->opt_tiling) return false;
  TileGeom g; tiling_geometry(c, enc, &g);
  return W > g.tile_w || H > g.tile_h;
}
// x [T][H][W][C] channels-last -> *out [T'][H'][W'][ld] (arena block of the caller's).  The tiles are independent until the blend, so all
// tiles of one shape (interior / bottom edge / right edge / corner: 12 + 4 + 3 + 1 at 720x1280) run as ONE batch - dove_conv_desc.nb and the
// *_nb operators, one launch per operator and shape class instead of one per tile, the largest class first (dove_amd/vae.py _tiled; per tile
// the arithmetic is that of a tile-by-tile loop, bit for bit).
// im2col_cin > 0 (encode): x is the im2col'ed clip and dove_tile_gather_bf16 zeroes the taps that reach outside a tile, so encoder.conv_in keeps
// its (3,1,1) form inside tiles; the decoder's tap-split conv_out is finished per tile batch by dove_conv_out_gather_cl (channels-last for the blend).
static int tiled(dove_ctx* c, const Tensor& x, bool enc, int im2col_cin, Tensor* out, void* stream) {
  TileGeom g; tiling_geometry(c, enc, &g);
  DOVE_CHECK_ARG(g.stride_h > 0 && g.stride_w > 0 && g.lim_h > 0 && g.lim_w > 0, "vae tiling: degenerate tile geometry (sample size %d x %d)", c->sample_h, c->sample_w);
  DOVE_CHECK_ARG(x.C % 8 == 0 && x.H < 32768 && x.W < 32768, "vae tiling: unsupported input layout");
  hipStream_t s = (hipStream_t)stream;
  const int batch = enc ? c->cfg.vae_enc_batch : c->cfg.vae_dec_batch;
  std::vector<std::pair<int, int>> fb;
  frame_batches(x.T, batch, &fb);
  std::vector<int> ii, jj;
  for (int i = 0; i < x.H; i += g.stride_h) ii.push_back(i);
  for (int j = 0; j < x.W; j += g.stride_w) jj.push_back(j);
  std::vector<std::vector<Tensor>> rows(ii.size(), std::vector<Tensor>(jj.size()));
  struct Cls { int th, tw; std::vector<std::pair<int, int>> members; };
  std::vector<Cls> classes;
  for (size_t a = 0; a < ii.size(); ++a)
    for (size_t b = 0; b < jj.size(); ++b) {
      const int th = std::min(g.tile_h, x.H - ii[a]), tw = std::min(g.tile_w, x.W - jj[b]);
      size_t k = 0;
      while (k < classes.size() && !(classes[k].th == th && classes[k].tw == tw)) ++k;
      if (k == classes.size()) classes.push_back({th, tw, {}});
      classes[k].members.push_back({(int)a, (int)b});
    }
  std::stable_sort(classes.begin(), classes.end(), [](const Cls& p, const Cls& q) {
    return (long long)p.th * p.tw * (long long)p.members.size() > (long long)q.th * q.tw * (long long)q.members.size(); });
  std::vector<void*> class_blocks;
  int t_total = 0;
  for (auto& se : fb) t_total += enc ? enc_batch_frames(c, se.second - se.first) : dec_batch_frames(c, se.second - se.first);
  int rc = 0;
  for (size_t k = 0; k < classes.size() && !rc; ++k) {
    const Cls& cl = classes[k];
    // at most 16 tiles per batch: activations scale with the batch (the 30 interior tiles of a 1088x1920 clip would hold > 100 GB live), and
    // 16 is past the point where the launches fill the chip; per tile the result does not depend on the batching
    constexpr size_t kTileBatchMax = 16;
    for (size_t m0 = 0; m0 < cl.members.size() && !rc; m0 += kTileBatchMax) {
      const int nb = (int)std::min<size_t>(kTileBatchMax, cl.members.size() - m0);
      int org_y[kTileBatchMax], org_x[kTileBatchMax];
      for (int n = 0; n < nb; ++n) { org_y[n] = ii[cl.members[m0 + n].first]; org_x[n] = jj[cl.members[m0 + n].second]; }
      clear_caches(c);
      c->nb = nb;
      Tensor cls_out;                                           // [nb][t_total][oh][ow][ld], allocated once the first batch says oh, ow, ld
      int t_done = 0;
      for (auto& se : fb) {
        const int nt = se.second - se.first;
        Tensor xb;
        if ((rc = alloc_t(c, nb * nt, cl.th, cl.tw, x.C, &xb))) break;
        if ((rc = dove_tile_gather_bf16(x.p, x.H, x.W, x.C, se.first, nt, cl.th, cl.tw, nb, org_y, org_x, im2col_cin, xb.p, stream))) { free_t(c, xb); break; }
        Tensor o;
        rc = enc ? encoder(c, xb, &o, stream) : decoder(c, xb, &o, stream);
        // the first conv does not own its input (a view in the un-tiled paths): release the batch tensor here.  Its conv cache was copied.
        free_t(c, xb);
        if (rc) break;
        if (!enc && c->conv_out_bias) {                           // fp32 partial planes of the tap-split conv_out -> the tile batch, channels-last
          Tensor g;
          if ((rc = alloc_t(c, o.T, o.H, o.W, 8, &g))) { free_t(c, o); break; }
          rc = dove_conv_out_gather_cl((const float*)o.p, o.C / 2, o.T, o.H, o.W, c->cfg.vae_out_channels, c->conv_out_bias, g.p, 8, stream);
          free_t(c, o);
          if (rc) { free_t(c, g); break; }
          o = g;
        }
        const int To = o.T / nb;
        if (!cls_out.p) {
          cls_out.T = nb * t_total; cls_out.H = o.H; cls_out.W = o.W; cls_out.C = o.C;
          cls_out.p = (bf16_t*)c->arena.alloc(cls_out.bytes(), true);
          if (!cls_out.p) { dove_set_error("workspace exhausted (vae tile outputs: %zu bytes)", cls_out.bytes()); rc = DOVE_EINVAL; free_t(c, o); break; }
          class_blocks.push_back(cls_out.p);
        }
        const long long oframe = (long long)o.H * o.W * o.C;
        rc = copy2d(cls_out.p + (long long)t_done * oframe, (size_t)t_total * oframe * 2, o.p, (size_t)To * oframe * 2, (size_t)To * oframe * 2, nb, s);
        t_done += To;
        free_t(c, o);
        if (rc) break;
      }
      clear_caches(c);
      c->nb = 1;
      if (!rc && t_done != t_total) { dove_set_error("vae tiling: a tile produced %d frames, expected %d", t_done, t_total); rc = DOVE_EINVAL; }
      if (rc) break;
      for (int n = 0; n < nb; ++n) {
        Tensor t; t.T = t_total; t.H = cls_out.H; t.W = cls_out.W; t.C = cls_out.C;
        t.p = cls_out.p + (long long)n * t_total * cls_out.H * cls_out.W * cls_out.C;
        rows[cl.members[m0 + n].first][cl.members[m0 + n].second] = t;
      }
    }
  }
  c->nb = 1;
  int Ho = 0, Wo = 0;
  if (!rc) {
    for (size_t i = 0; i < rows.size() && !rc; ++i)
      for (size_t j = 0; j < rows[i].size() && !rc; ++j) {
        Tensor& t = rows[i][j];
        if (i > 0) { const Tensor& a = rows[i - 1][j]; rc = dove_blend_edge_bf16(a.p, t.p, t.T, a.H, a.W, t.H, t.W, t.C, std::min(std::min(a.H, t.H), g.blend_h), 0, stream); }
        if (!rc && j > 0) { const Tensor& a = rows[i][j - 1]; rc = dove_blend_edge_bf16(a.p, t.p, t.T, a.H, a.W, t.H, t.W, t.C, std::min(std::min(a.W, t.W), g.blend_w), 1, stream); }
      }
    for (auto& r : rows) Ho += std::min(r[0].H, g.lim_h);
    for (auto& t : rows[0]) Wo += std::min(t.W, g.lim_w);
  }
  if (!rc) {
    const Tensor& t0 = rows[0][0];
    rc = alloc_t(c, t0.T, Ho, Wo, t0.C, out);
    int y0 = 0;
    for (size_t i = 0; i < rows.size() && !rc; ++i) {
      int x0 = 0;
      const int ch = std::min(rows[i][0].H, g.lim_h);
      for (size_t j = 0; j < rows[i].size() && !rc; ++j) {
        const Tensor& t = rows[i][j];
        const int cw = std::min(t.W, g.lim_w);
        for (int f = 0; f < t.T && !rc; ++f)
          rc = copy2d(out->p + (((long long)f * Ho + y0) * Wo + x0) * t.C, (size_t)Wo * t.C * 2, t.p + (long long)f * t.H * t.W * t.C, (size_t)t.W * t.C * 2,
                      (size_t)cw * t.C * 2, ch, s);
        x0 += cw;
      }
      y0 += ch;
    }
  }
  for (void* b : class_blocks) c->arena.release(b);
  return rc;
}

static int ensure_ws(dove_ctx* c, int F, int H, int W) {
  DOVE_CHECK_ARG(c && c->finalized, "context is not finalized (dove_finalize_weights)");
  HIPCHK(hipSetDevice(c->device));
  const size_t want = dove_workspace_bytes(c, F, H, W);
  // no workspace yet, or a library-owned one that is too small and idle: (re)allocate.  A caller-provided workspace is used as is
  // (allocation failures inside the stage then say how much was missing)
  if (!c->arena.base || (c->arena.owned && c->arena.cap < want && c->arena.live.empty())) CHK(dove_set_workspace(c, nullptr, want));
  return 0;
}

// ---- two-stream frame-batch loop (dove_ctx::opt_vae_streams) ----
struct VaeStreams {
  dove_ctx* c; hipStream_t caller;
  VaeStreams(dove_ctx* ctx, void* stream) : c(ctx), caller((hipStream_t)stream) {}
  int begin(size_t nitems) {
    c->vae_multi = false;
    if (c->opt_vae_streams < 2 || c->nranks > 1 || nitems < 2 || c->nb > 1) return 0;
    if (!c->vae_stream2) HIPCHK(hipStreamCreateWithFlags(&c->vae_stream2, hipStreamNonBlocking));
    if (!c->gn_ws2) { void* m; CHK(dev_alloc(c, (size_t)c->gn_ws_rows * 64 * 4, &m)); c->gn_ws2 = (float*)m; }
    Arena& a = c->arena;
    a.begin_tracking(caller, c->vae_stream2);
    c->cache_ev.clear();
    const long long e = a.next_event();                       // the second stream starts behind what the caller's stream holds so far (the converted clip)
    HIPCHK(hipEventRecord(a.pool[(size_t)e], caller));
    HIPCHK(hipStreamWaitEvent(c->vae_stream2, a.pool[(size_t)e], 0));
    c->vae_multi = true;
    return 0;
  }
  void* item(size_t i) {                                      // stream of work item i (and the arena's current stream with it)
    if (!c->vae_multi) return (void*)caller;
    c->arena.cur = (int)(i & 1);
    return (void*)c->arena.streams[i & 1];
  }
  int end() {                                                 // join: the caller's stream continues behind the second one; tracking off
    if (!c->vae_multi) return 0;
    Arena& a = c->arena;
    c->vae_multi = false;
    const long long e = a.next_event();
    const hipError_t r1 = hipEventRecord(a.pool[(size_t)e], c->vae_stream2), r2 = hipStreamWaitEvent(caller, a.pool[(size_t)e], 0);
    a.end_tracking(); c->cache_ev.clear();
    if (r1 != hipSuccess || r2 != hipSuccess) { dove_set_error("two-stream VAE: joining the streams failed"); return DOVE_ELAUNCH; }
    return 0;
  }
  ~VaeStreams() { (void)end(); }                              // also on an early error return: nothing of this stage stays in flight un-joined
};

// moments [2L][T][h][w] (dtype) of x [3][F][H][W] (dtype), frame-batched like diffusers' _encode; conv caches are per call
static int vae_encode_cl(dove_ctx* c, const void* x, int dtype, int F, int H, int W, Tensor* moments, void* stream) {
  const auto& cf = c->cfg;
  Tensor xcl;
  CHK(alloc_t(c, F, H, W, c->pc.at("encoder.conv_in").cin_pad, &xcl));
  if (wants_tiling(c, true, H, W)) {
    DOVE_CHECK_ARG(c->nranks == 1, "vae tiling is not combined with the multi-rank halo exchange");
    const bool taps = c->pc.count("encoder.conv_in.taps") != 0;
    if (taps) CHK(dove_cl_im2col3x3_from_ncthw(x, dtype, cf.vae_in_channels, F, H, W, xcl.C, 1.0f, 0.0f, xcl.p, stream));
    else CHK(dove_cl_from_ncthw(x, dtype, cf.vae_in_channels, (long long)F * H * W, xcl.C, 1.0f, 0.0f, xcl.p, stream));
    int rc = tiled(c, xcl, true, taps ? cf.vae_in_channels : 0, moments, stream);
    free_t(c, xcl);
    clear_caches(c);
    if (!rc && (moments->T != 1 + (F - 1) / cf.vae_temporal_compression || moments->H != H / 8 || moments->W != W / 8)) {
      dove_set_error("vae tiling: stitched moments are %d x %d x %d for a %d x %d x %d clip", moments->T, moments->H, moments->W, F, H, W);
      free_t(c, *moments);
      rc = DOVE_EINVAL;
    }
    return rc;
  }
  if (c->pc.count("encoder.conv_in.taps")) {
    CHK(dove_cl_im2col3x3_from_ncthw(x, dtype, cf.vae_in_channels, F, H, W, xcl.C, 1.0f, 0.0f, xcl.p, stream));
  } else {
    CHK(dove_cl_from_ncthw(x, dtype, cf.vae_in_channels, (long long)F * H * W, xcl.C, 1.0f, 0.0f, xcl.p, stream));
  }
  std::vector<std::pair<int, int>> fb;
  frame_batches(F, cf.vae_enc_batch, &fb);
  clear_caches(c);
  const int T = 1 + (F - 1) / cf.vae_temporal_compression, h = H / 8, w = W / 8;
  const int ld = c->pc.at("encoder.conv_out").cout_store();
  CHK(alloc_t(c, T, h, w, ld, moments));
  RankPlan rp;
  rank_plan(c, true, F, &rp);
  if (rp.total_out != T) { dove_set_error("dove_vae_encode: the rank plan yields %d latent frames, expected %d", rp.total_out, T); return DOVE_EINVAL; }
  c->stat_halo_preposted = c->stat_halo_blocking = c->stat_halo_sent = 0;
  VaeStreams vs(c, stream);
  CHK(vs.begin(rp.mine.size()));
  void* const caller_stream = stream;
  for (size_t i = 0; i < rp.mine.size(); ++i) {
    stream = vs.item(i);
    const Piece& pc = rp.mine[i];
    c->halo_recv = c->nranks > 1 && i == 0 && c->rank > 0;
    c->halo_send = c->nranks > 1 && i + 1 == rp.mine.size() && c->rank < rp.active - 1;
    set_piece(c, &pc);
    Tensor xb = xcl; xb.p = xcl.p + (long long)pc.s * H * W * xcl.C; xb.T = pc.e - pc.s;
    Tensor o;
    const bool first_recv = c->halo_recv;
    if (first_recv) CHK(halo_begin(c, "enc:" + std::to_string(F) + "x" + std::to_string(H) + "x" + std::to_string(W) + ":" + std::to_string(c->nranks) + ":" + std::to_string(c->rank), stream));
    int rc = encoder(c, xb, &o, stream);
    if (!rc && first_recv) rc = halo_end_first_item(c);
    c->halo_recv = c->halo_send = false;
    set_piece(c, nullptr);
    CHK(rc);
    HIPCHK(hipMemcpyAsync(moments->p + (long long)rp.out_first[i] * h * w * ld, o.p, o.bytes(), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    free_t(c, o);
  }
  stream = caller_stream;
  CHK(vs.end());
  free_t(c, xcl);
  clear_caches(c);
  CHK(halo_stage_end(c, stream));
  return 0;
}
extern "C" int dove_vae_encode(dove_ctx* c, const void* x, int dtype, int F, int H, int W, void* moments_out, int out_dtype, void* stream) {
  DOVE_CHECK_ARG(x && moments_out, "dove_vae_encode: null pointer");
  DOVE_CHECK_ARG(F >= 1 && H % 8 == 0 && W % 8 == 0 && H >= 8 && W >= 8, "dove_vae_encode: H and W must be multiples of 8");
  StageGuard guard(c);
  CHK(ensure_ws(c, F, H, W));
  Tensor m;
  CHK(vae_encode_cl(c, x, dtype, F, H, W, &m, stream));
  if (c->nranks == 1) {
    CHK(dove_ncthw_from_cl(m.p, m.C, 2 * c->cfg.vae_latent_channels, (long long)m.T * m.H * m.W, 1.0f, 0.0f, -INFINITY, INFINITY, moments_out, out_dtype, stream));
  } else {
    // only this rank's latent frames exist: convert them into a staging block and place them at their frame offset of [2L][T][h][w]
    int first = 0, count = 0;
    CHK(dove_shard_frames(c, 0, F, &first, &count));
    if (count > 0) {
      const size_t esz = out_dtype == DOVE_F32 ? 4 : 2;
      const int C2 = 2 * c->cfg.vae_latent_channels;
      const long long hw = (long long)m.H * m.W;
      void* tmp = c->arena.alloc((size_t)C2 * count * hw * esz);
      DOVE_CHECK_ARG(tmp, "workspace exhausted (encoder output staging)");
      CHK(dove_ncthw_from_cl(m.p + (long long)first * hw * m.C, m.C, C2, (long long)count * hw, 1.0f, 0.0f, -INFINITY, INFINITY, tmp, out_dtype, stream));
      CHK(copy2d((char*)moments_out + (size_t)first * hw * esz, (size_t)m.T * hw * esz, tmp, (size_t)count * hw * esz, (size_t)count * hw * esz, C2,
                 (hipStream_t)stream));
      c->arena.release(tmp);
    }
  }
  free_t(c, m);
  return DOVE_OK;
}
// frames the decoder returns for T latent frames: per latent frame-batch (diffusers' _decode batching) every temporal x2 stage maps
// t frames to 2t - 1 (t odd, > 1: the first frame is not doubled), 2t (t even) or 1 (t == 1); 1 + 4(T - 1) for the odd T the
// reference always has (clips are padded to 8N + 1 frames, ref :220-232)
extern "C" int dove_vae_decode_num_frames(dove_ctx* c, int T) {
  if (!c || T < 1) return 0;
  std::vector<std::pair<int, int>> fb;
  frame_batches(T, c->cfg.vae_dec_batch, &fb);
  int n_tdown = 0;
  for (int r = c->cfg.vae_temporal_compression; r > 1; r >>= 1) ++n_tdown;
  int F = 0;
  for (auto& se : fb) {
    int t = se.second - se.first;
    for (int i = 0; i < n_tdown && i < c->cfg.vae_num_blocks - 1; ++i) t = t > 1 ? (t % 2 ? 2 * t - 1 : 2 * t) : 1;
    F += t;
  }
  return F;
}
// z [L][T][h][w] (dtype; multiplied by `prescale` on the way in) -> video [3][F][8h][8w], F = dove_vae_decode_num_frames(T);
// range01: clamp(x*0.5+0.5, 0, 1)
extern "C" int dove_vae_decode(dove_ctx* c, const void* z, int dtype, int T, int h, int w, float prescale, int range01, void* video_out,
                               int out_dtype, void* stream) {
  DOVE_CHECK_ARG(z && video_out && T >= 1 && h >= 1 && w >= 1, "dove_vae_decode: bad argument");
  const auto& cf = c->cfg;
  const int F = dove_vae_decode_num_frames(c, T), H = 8 * h, W = 8 * w;
  StageGuard guard(c);
  CHK(ensure_ws(c, F, H, W));
  Tensor zcl;
  CHK(alloc_t(c, T, h, w, c->pc.at("decoder.conv_in").cin_pad, &zcl));
  CHK(dove_cl_from_ncthw(z, dtype, cf.vae_latent_channels, (long long)T * h * w, zcl.C, prescale, 0.0f, zcl.p, stream));
  if (wants_tiling(c, false, h, w)) {
    DOVE_CHECK_ARG(c->nranks == 1, "vae tiling is not combined with the multi-rank halo exchange");
    Tensor full;
    int rc = tiled(c, zcl, false, 0, &full, stream);
    free_t(c, zcl);
    clear_caches(c);
    CHK(rc);
    if (full.T != F || full.H != H || full.W != W) {
      dove_set_error("vae tiling: stitched output is %d x %d x %d, expected %d x %d x %d", full.T, full.H, full.W, F, H, W);
      free_t(c, full);
      return DOVE_EINVAL;
    }
    rc = dove_ncthw_from_cl(full.p, full.C, cf.vae_out_channels, (long long)F * H * W, range01 ? 0.5f : 1.0f, range01 ? 0.5f : 0.0f,
                            range01 ? 0.0f : -INFINITY, range01 ? 1.0f : INFINITY, video_out, out_dtype, stream);
    free_t(c, full);
    return rc;
  }
  std::vector<std::pair<int, int>> fb;
  frame_batches(T, cf.vae_dec_batch, &fb);
  clear_caches(c);
  const size_t esz = out_dtype == DOVE_F32 ? 4 : 2;
  RankPlan rp;
  rank_plan(c, false, T, &rp);
  if (rp.total_out != F) { dove_set_error("dove_vae_decode: the rank plan yields %d frames, expected %d", rp.total_out, F); return DOVE_EINVAL; }
  c->stat_halo_preposted = c->stat_halo_blocking = c->stat_halo_sent = 0;
  VaeStreams vs(c, stream);
  CHK(vs.begin(rp.mine.size()));
  void* const caller_stream = stream;
  for (size_t i = 0; i < rp.mine.size(); ++i) {
    stream = vs.item(i);
    const Piece& pc = rp.mine[i];
    c->halo_recv = c->nranks > 1 && i == 0 && c->rank > 0;
    c->halo_send = c->nranks > 1 && i + 1 == rp.mine.size() && c->rank < rp.active - 1;
    set_piece(c, &pc);
    Tensor zb = zcl; zb.p = zcl.p + (long long)pc.s * h * w * zcl.C; zb.T = pc.e - pc.s;
    Tensor o;
    const bool first_recv = c->halo_recv;
    if (first_recv) CHK(halo_begin(c, "dec:" + std::to_string(T) + "x" + std::to_string(h) + "x" + std::to_string(w) + ":" + std::to_string(c->nranks) + ":" + std::to_string(c->rank), stream));
    int rc = decoder(c, zb, &o, stream);
    if (!rc && first_recv) rc = halo_end_first_item(c);
    c->halo_recv = c->halo_send = false;
    set_piece(c, nullptr);
    CHK(rc);
    const int f0 = rp.out_first[i];
    // [C][F][H][W] output: this piece's frames are not contiguous per channel -> convert into a staging tensor, then strided copy
    void* tmp = c->arena.alloc((size_t)cf.vae_out_channels * o.T * H * W * esz);
    DOVE_CHECK_ARG(tmp, "workspace exhausted (decoder output staging)");
    if (c->conv_out_bias) {
      CHK(dove_conv_out_gather((const float*)o.p, o.C / 2, o.T, H, W, cf.vae_out_channels, c->conv_out_bias, range01 ? 0.5f : 1.0f,
                               range01 ? 0.5f : 0.0f, range01 ? 0.0f : -INFINITY, range01 ? 1.0f : INFINITY, tmp, out_dtype, stream));
    } else {
      CHK(dove_ncthw_from_cl(o.p, o.C, cf.vae_out_channels, (long long)o.T * H * W, range01 ? 0.5f : 1.0f, range01 ? 0.5f : 0.0f,
                             range01 ? 0.0f : -INFINITY, range01 ? 1.0f : INFINITY, tmp, out_dtype, stream));
    }
    CHK(copy2d((char*)video_out + (size_t)f0 * H * W * esz, (size_t)F * H * W * esz, tmp, (size_t)o.T * H * W * esz, (size_t)o.T * H * W * esz,
               cf.vae_out_channels, (hipStream_t)stream));
    c->arena.release(tmp);
    free_t(c, o);
  }
  stream = caller_stream;
  CHK(vs.end());
  free_t(c, zcl);
  clear_caches(c);
  CHK(halo_stage_end(c, stream));
  return DOVE_OK;
}

// The same forward with the ranks of the context's communicator sharing ONE sample (dove_amd/dist.py dit_forward_ulysses, Ulysses-style):
// the token-major residual stream [N, D] is sharded by ROWS for every row-local operator (LayerNormZero, the four linears with their gated
// residuals, QK-LayerNorm + RoPE) and by HEADS (heads / nranks each) for attention, with one all-to-all of Q', K', V^T before the attention
// kernel and one of its output after it per layer; the per-head score bound rides in the K blocks (dove_ulysses_place_bf16).  Every row and
// every head sees the single-GPU arithmetic, so v_out - complete on every rank - is bit-identical to the one-GPU call.
static int dit_forward_ulysses(dove_ctx* c, const void* hidden, int dtype, int T, int h, int w, const void* text, int L, int timestep,
                               const dove_dit_aux* aux, void* v_out, int out_dtype, void* stream) {
  const auto& cf = c->cfg;
  const int p = cf.dit_patch, pt = cf.dit_patch_t, Cc = cf.dit_in_channels, D = cf.dit_heads * cf.dit_head_dim, Hh = cf.dit_heads;
  const int R = c->nranks, me = c->rank;
  DOVE_CHECK_ARG(!c->opt_linear_mx && !c->opt_attn_mx, "dove_dit_forward: the sharded DiT runs the bf16 path (configs[4] is one clip per GPU)");
  DOVE_CHECK_ARG(Hh % R == 0, "dove_dit_forward: %d attention heads do not split over %d ranks", Hh, R);
  DOVE_CHECK_ARG(R <= 16, "dove_dit_forward: at most 16 ranks");
  hipStream_t s = (hipStream_t)stream;
  const int hloc = Hh / R;
  const long long nv = (long long)(T / pt) * (h / p) * (w / p), N = L + nv, npad = ru(N, 128);
  std::vector<long long> bounds(R + 1), counts(R);
  for (int i = 0; i <= R; ++i) bounds[i] = (long long)i * N / R;
  for (int i = 0; i < R; ++i) counts[i] = bounds[i + 1] - bounds[i];
  const long long r0 = bounds[me], r1 = bounds[me + 1], nloc = r1 - r0;
  DOVE_CHECK_ARG(nloc > 0, "dove_dit_forward: more ranks than tokens");
  const long long lt_loc = std::max<long long>(0, std::min<long long>(r1, L) - r0);      // local rows that are text rows (they come first)
  const long long v0 = std::max<long long>(r0, L) - L, v1 = std::max<long long>(r1, L) - L;   // local video rows [v0, v1)
  const float *cosp, *sinp;
  if (aux && aux->rope_cos && aux->rope_sin) { cosp = aux->rope_cos; sinp = aux->rope_sin; }
  else CHK(rope_tables(c, T / pt, h / p, w / p, &cosp, &sinp, stream));
  CHK(modulation(c, timestep, aux ? aux->timestep_proj : nullptr, stream));
  auto A = [&](size_t bytes) -> bf16_t* { return (bf16_t*)c->arena.alloc(bytes); };
  auto Z = [&](size_t bytes) -> bf16_t* { bf16_t* q = (bf16_t*)c->arena.alloc(bytes); if (q) (void)hipMemsetAsync(q, 0, bytes, s); return q; };
  bf16_t* hs = A((size_t)nloc * D * 2);
  bf16_t* n1 = A((size_t)nloc * D * 2);
  DOVE_CHECK_ARG(hs && n1, "workspace exhausted (DiT token buffers)");
  ConvOpt o; bf16_t* dummy;
  if (lt_loc) { o = ConvOpt(); o.out = hs; CHK(linear(c, (const bf16_t*)text + r0 * cf.dit_text_dim, lt_loc, c->pe_text, o, &dummy, stream)); }
  if (v1 > v0) {
    const int feat_pad = c->pe_proj.cin_pad;
    bf16_t* tok = A((size_t)nv * feat_pad * 2);
    DOVE_CHECK_ARG(tok, "workspace exhausted (DiT tokens)");
    if (feat_pad > Cc * pt * p * p) HIPCHK(hipMemsetAsync(tok, 0, (size_t)nv * feat_pad * 2, s));
    CHK(dove_patchify(hidden, dtype, T, Cc, h, w, pt, p, tok, feat_pad, stream));
    o = ConvOpt(); o.out = hs + (size_t)lt_loc * D;
    CHK(linear(c, tok + (size_t)v0 * feat_pad, v1 - v0, c->pe_proj, o, &dummy, stream));
    c->arena.release(tok);
  }
  const float* cos_l = v1 > v0 ? cosp + v0 * 64 : cosp;        // a text-only shard still hands a valid table to the kernel
  const float* sin_l = v1 > v0 ? sinp + v0 * 64 : sinp;
  // Rank-local head-major operands with row stride nloc + 1: rows [0, nloc) are data, so "my rows of rank j's heads" is one contiguous
  // all-to-all chunk; the extra row of a K head carries this rank's score-bound pair of that head to the rank that owns the head
  const size_t qkl = (size_t)Hh * (nloc + 1) * 64 * 2;
  bf16_t *Ql = Z(qkl), *Kl = Z(qkl), *Vl = Z(qkl);
  const size_t ab = (size_t)hloc * npad * 64 * 2;
  bf16_t *Qh = Z(ab), *Kh = Z(ab), *Vt = Z(ab);                 // pad rows / columns stay zero across the layers
  const size_t rb = (size_t)(N + R) * hloc * 64 * 2;
  bf16_t *rq = A(rb), *rk = A(rb), *rv = A(rb);
  bf16_t* att = A((size_t)N * hloc * 64 * 2);
  bf16_t* back = A((size_t)nloc * hloc * 64 * R * 2);
  bf16_t* att_loc = A((size_t)nloc * D * 2);
  float* norm2 = (float*)c->arena.alloc((size_t)Hh * 2 * sizeof(float));
  float* norm2_mine = (float*)c->arena.alloc((size_t)hloc * 2 * sizeof(float));
  DOVE_CHECK_ARG(Ql && Kl && Vl && Qh && Kh && Vt && rq && rk && rv && att && back && att_loc && norm2 && norm2_mine, "workspace exhausted (sharded attention buffers)");
  std::vector<size_t> s_off(R), s_cnt(R), r_off(R), r_cnt(R), b_soff(R), b_scnt(R), b_roff(R), b_rcnt(R);
  size_t ro = 0;
  for (int j = 0; j < R; ++j) {
    s_cnt[j] = (size_t)(nloc + 1) * hloc * 64 * 2; s_off[j] = (size_t)j * s_cnt[j];            // chunk j of Ql / Kl / Vl = head group j
    r_cnt[j] = (size_t)(counts[j] + 1) * hloc * 64 * 2; r_off[j] = ro; ro += r_cnt[j];
    b_scnt[j] = (size_t)counts[j] * hloc * 64 * 2; b_soff[j] = (size_t)bounds[j] * hloc * 64 * 2;   // rows of rank j in att [N][hloc*64]
    b_rcnt[j] = (size_t)nloc * hloc * 64 * 2; b_roff[j] = (size_t)j * b_rcnt[j];
  }
  const float qscale = (1.0f / sqrtf((float)cf.dit_head_dim)) * 1.4426950408889634f;
  for (auto& b : c->blocks) {
    CHK(dove_layernorm_modulate_bf16(hs, n1, nloc, D, cf.dit_norm_eps, b.ln1_g, b.ln1_b, b.m1, lt_loc, stream));
    bf16_t* qkv;
    CHK(linear(c, n1, nloc, b.qkv, ConvOpt(), &qkv, stream));
    CHK(dove_qkv_post_bf16(qkv, nloc, nloc + 1, Hh, 64, (int)lt_loc, b.nq_g, b.nq_b, b.nk_g, b.nk_b, cos_l, sin_l, qscale, 1e-6f, Ql, Kl, Vl, /*v_order=*/0, norm2, stream));
    c->arena.release(qkv);
    // the K heads' extra rows: two floats per head
    HIPCHK(hipMemcpy2DAsync(Kl + (size_t)nloc * 64, (size_t)(nloc + 1) * 64 * 2, norm2, 2 * sizeof(float), 2 * sizeof(float), Hh, hipMemcpyDeviceToDevice, s));
    CHK(all_to_all(c, Ql, s_off.data(), s_cnt.data(), rq, r_off.data(), r_cnt.data(), stream));
    CHK(all_to_all(c, Kl, s_off.data(), s_cnt.data(), rk, r_off.data(), r_cnt.data(), stream));
    CHK(all_to_all(c, Vl, s_off.data(), s_cnt.data(), rv, r_off.data(), r_cnt.data(), stream));
    CHK(dove_ulysses_place_bf16(rq, rk, rv, counts.data(), R, hloc, N, npad, Qh, Kh, Vt, norm2_mine, stream));
    CHK(dove_attention_fwd_bf16(Qh, Kh, Vt, att, N, npad, hloc, 64, (long long)hloc * 64, norm2_mine, stream));
    // heads -> rows: rank j gets rows [bounds[j], bounds[j+1]) of my heads; I get my rows of every head group
    CHK(all_to_all(c, att, b_soff.data(), b_scnt.data(), back, b_roff.data(), b_rcnt.data(), stream));
    for (int j = 0; j < R; ++j)                                 // [head group j][my rows][hloc*64] -> [my rows][all heads]
      CHK(copy2d(att_loc + (size_t)j * hloc * 64, (size_t)D * 2, back + (size_t)j * nloc * hloc * 64, (size_t)hloc * 64 * 2, (size_t)hloc * 64 * 2, (int)nloc, s));
    o = ConvOpt(); o.resid = hs; o.ldr = D; o.gate = b.g1; o.gate_split = lt_loc; o.out = hs;
    CHK(linear(c, att_loc, nloc, b.out, o, &dummy, stream));
    CHK(dove_layernorm_modulate_bf16(hs, n1, nloc, D, cf.dit_norm_eps, b.ln2_g, b.ln2_b, b.m2, lt_loc, stream));
    bf16_t* f1;
    o = ConvOpt(); o.act = 1;
    CHK(linear(c, n1, nloc, b.ff1, o, &f1, stream));
    o = ConvOpt(); o.resid = hs; o.ldr = D; o.gate = b.g2; o.gate_split = lt_loc; o.out = hs;
    CHK(linear(c, f1, nloc, b.ff2, o, &dummy, stream));
    c->arena.release(f1);
  }
  for (void* q : {(void*)Ql, (void*)Kl, (void*)Vl, (void*)Qh, (void*)Kh, (void*)Vt, (void*)rq, (void*)rk, (void*)rv, (void*)att, (void*)back, (void*)att_loc, (void*)norm2, (void*)norm2_mine})
    c->arena.release(q);
  // output head on my video rows, then every rank gathers all rows: the un-patchified velocity is complete everywhere
  const int width = c->proj_out.cout_store();
  bf16_t* oall = A((size_t)nv * width * 2);
  DOVE_CHECK_ARG(oall, "workspace exhausted (DiT head)");
  std::vector<size_t> g_off(R), g_cnt(R);
  for (int j = 0; j < R; ++j) {
    const long long a0 = std::max<long long>(bounds[j], L) - L, a1 = std::max<long long>(bounds[j + 1], L) - L;
    g_off[j] = (size_t)a0 * width * 2; g_cnt[j] = (size_t)(a1 - a0) * width * 2;
  }
  bf16_t* po = nullptr;
  if (v1 > v0) {
    bf16_t* xv = hs + (size_t)lt_loc * D;
    bf16_t* a1 = A((size_t)(v1 - v0) * D * 2);
    DOVE_CHECK_ARG(a1, "workspace exhausted (DiT head)");
    CHK(dove_layernorm_modulate_bf16(xv, a1, v1 - v0, D, cf.dit_norm_eps, c->nf_g, c->nf_b, nullptr, 0, stream));
    CHK(dove_layernorm_modulate_bf16(a1, n1, v1 - v0, D, cf.dit_norm_eps, c->no_g, c->no_b, c->final_mod, 0, stream));
    CHK(linear(c, n1, v1 - v0, c->proj_out, ConvOpt(), &po, stream));
    c->arena.release(a1);
  }
  CHK(all_gather_v(c, po, oall, g_off.data(), g_cnt.data(), stream));
  CHK(dove_unpatchify(oall, width, T, cf.dit_out_channels, h, w, pt, p, v_out, out_dtype, stream));
  c->arena.release(po); c->arena.release(oall); c->arena.release(n1); c->arena.release(hs);
  return DOVE_OK;
}

// CogVideoXTransformer3DModel.forward for one sample: hidden [T][C][h][w] (dtype), text [L][text_dim] bf16, timestep t ->
// v_out [T][C][h][w] (dtype).  aux (optional, for bit-exact comparisons with a host that computes them itself): rope cos / sin
// [Nv][64] fp32 and the bf16-rounded sinusoidal timestep projection [D] fp32, all device pointers.
extern "C" int dove_dit_forward(dove_ctx* c, const void* hidden, int dtype, int T, int h, int w, const void* text, int L, int timestep,
                                const dove_dit_aux* aux, void* v_out, int out_dtype, void* stream) {
  DOVE_CHECK_ARG(hidden && text && v_out, "dove_dit_forward: null pointer");
  const auto& cf = c->cfg;
  const int p = cf.dit_patch, pt = cf.dit_patch_t, Cc = cf.dit_in_channels, D = cf.dit_heads * cf.dit_head_dim, Hh = cf.dit_heads;
  DOVE_CHECK_ARG(T % pt == 0 && h % p == 0 && w % p == 0 && L >= 1, "dove_dit_forward: T / h / w must be multiples of the patch sizes");
  StageGuard guard(c);
  CHK(ensure_ws(c, 1 + 4 * (T - 1), 8 * h, 8 * w));
  if (c->nranks > 1) return dit_forward_ulysses(c, hidden, dtype, T, h, w, text, L, timestep, aux, v_out, out_dtype, stream);
  const long long nv = (long long)(T / pt) * (h / p) * (w / p), N = L + nv, npad = ru(N, 128);
  const float *cosp, *sinp;
  if (aux && aux->rope_cos && aux->rope_sin) { cosp = aux->rope_cos; sinp = aux->rope_sin; }
  else CHK(rope_tables(c, T / pt, h / p, w / p, &cosp, &sinp, stream));
  CHK(modulation(c, timestep, aux ? aux->timestep_proj : nullptr, stream));
  if (c->attn_n != N) {
    if (c->Qh) { (void)hipFree(c->Qh); (void)hipFree(c->Kh); (void)hipFree(c->Vt); }
    const size_t b = (size_t)Hh * npad * 64 * 2;
    HIPCHK(hipMalloc((void**)&c->Qh, b)); HIPCHK(hipMalloc((void**)&c->Kh, b)); HIPCHK(hipMalloc((void**)&c->Vt, b));
    HIPCHK(hipMemsetAsync(c->Qh, 0, b, (hipStream_t)stream)); HIPCHK(hipMemsetAsync(c->Kh, 0, b, (hipStream_t)stream)); HIPCHK(hipMemsetAsync(c->Vt, 0, b, (hipStream_t)stream));
    c->attn_n = N;
  }
  auto A = [&](size_t bytes) -> bf16_t* { return (bf16_t*)c->arena.alloc(bytes); };
  bf16_t* hs = A((size_t)N * D * 2);
  bf16_t* n1 = A((size_t)N * D * 2);
  const int feat_pad = c->pe_proj.cin_pad;
  bf16_t* tok = A((size_t)nv * feat_pad * 2);
  DOVE_CHECK_ARG(hs && n1 && tok, "workspace exhausted (DiT token buffers)");
  if (feat_pad > Cc * pt * p * p) HIPCHK(hipMemsetAsync(tok, 0, (size_t)nv * feat_pad * 2, (hipStream_t)stream));
  CHK(dove_patchify(hidden, dtype, T, Cc, h, w, pt, p, tok, feat_pad, stream));
  ConvOpt o; bf16_t* dummy;
  o = ConvOpt(); o.out = hs; CHK(linear(c, (const bf16_t*)text, L, c->pe_text, o, &dummy, stream));
  o = ConvOpt(); o.out = hs + (size_t)L * D; CHK(linear(c, tok, nv, c->pe_proj, o, &dummy, stream));
  c->arena.release(tok);
  const float qscale = (1.0f / sqrtf((float)cf.dit_head_dim)) * 1.4426950408889634f;
  float* norm2 = (float*)c->arena.alloc((size_t)Hh * 2 * sizeof(float));   // per-head score bound, dove_qkv_post_bf16 -> dove_attention_fwd_bf16
  DOVE_CHECK_ARG(norm2, "workspace exhausted (attention score bounds)");
  if (c->opt_attn_mx && c->attn8_n != N) {                      // e4m3 operands of dove_attention_fwd_mxfp8 (every padded row is rewritten per call)
    if (c->Q8) { (void)hipFree(c->Q8); (void)hipFree(c->K8); (void)hipFree(c->V8); (void)hipFree(c->Vs8); }
    const size_t b8 = (size_t)Hh * npad * 64;
    HIPCHK(hipMalloc((void**)&c->Q8, b8)); HIPCHK(hipMalloc((void**)&c->K8, b8)); HIPCHK(hipMalloc((void**)&c->V8, b8));
    HIPCHK(hipMalloc((void**)&c->Vs8, (size_t)Hh * (npad / 64) * 64 * 2));
    HIPCHK(hipMemsetAsync(c->Q8, 0, b8, (hipStream_t)stream)); HIPCHK(hipMemsetAsync(c->K8, 0, b8, (hipStream_t)stream));
    HIPCHK(hipMemsetAsync(c->V8, 0, b8, (hipStream_t)stream)); HIPCHK(hipMemsetAsync(c->Vs8, 0, (size_t)Hh * (npad / 64) * 64 * 2, (hipStream_t)stream));
    c->attn8_n = N;
  }
  // one of the block's four big linears: bf16 implicit GEMM, or (DOVE_OPT_DIT_LINEAR_MXFP8) activation quantised per call + block-scaled MFMA
  auto big = [&](const bf16_t* x, const Packed& pc, const PackedMx& p8, ConvOpt oo, bf16_t** out) -> int {
    if (!c->opt_linear_mx) return linear(c, x, N, pc, oo, out, stream);
    const int K = p8.K, Nn = p8.rows;
    uint8_t* xq = (uint8_t*)c->arena.alloc((size_t)N * K);
    uint32_t* xs = (uint32_t*)c->arena.alloc((size_t)(K / 256) * N * 2 * 4);
    bf16_t* y = oo.out ? oo.out : (bf16_t*)c->arena.alloc((size_t)N * Nn * 2);
    if (!xq || !xs || !y) { dove_set_error("workspace exhausted (MXFP8 linear)"); return DOVE_EINVAL; }
    CHK(dove_mx_quant_bf16(x, N, K, xq, xs, stream));
    CHK(dove_linear_mxfp8(xq, xs, p8.q, p8.s, p8.bias, oo.resid, oo.gate, y, N, Nn, K, Nn, oo.ldr, oo.gate_split, oo.act, stream));
    c->arena.release(xq); c->arena.release(xs);
    *out = y;
    return 0;
  };
  for (auto& b : c->blocks) {
    CHK(dove_layernorm_modulate_bf16(hs, n1, N, D, cf.dit_norm_eps, b.ln1_g, b.ln1_b, b.m1, L, stream));
    bf16_t* qkv;
    CHK(big(n1, b.qkv, b.qkv8, ConvOpt(), &qkv));
    if (c->opt_attn_mx) {
      CHK(dove_qkv_post_mxfp8(qkv, N, npad, Hh, 64, L, b.nq_g, b.nq_b, b.nk_g, b.nk_b, cosp, sinp, qscale, 1e-6f, c->Q8, c->K8, c->V8, c->Vs8, stream));
      c->arena.release(qkv);
      CHK(dove_attention_fwd_mxfp8(c->Q8, c->K8, c->V8, c->Vs8, n1, N, npad, Hh, 64, D, stream));
    } else {
      CHK(dove_qkv_post_bf16(qkv, N, npad, Hh, 64, L, b.nq_g, b.nq_b, b.nk_g, b.nk_b, cosp, sinp, qscale, 1e-6f, c->Qh, c->Kh, c->Vt, /*v_order=*/1, norm2, stream));
      c->arena.release(qkv);
      CHK(dove_attention_fwd_bf16(c->Qh, c->Kh, c->Vt, n1, N, npad, Hh, 64, D, norm2, stream));      // attention output reuses n1
    }
    o = ConvOpt(); o.resid = hs; o.ldr = D; o.gate = b.g1; o.gate_split = L; o.out = hs;
    CHK(big(n1, b.out, b.out8, o, &dummy));
    CHK(dove_layernorm_modulate_bf16(hs, n1, N, D, cf.dit_norm_eps, b.ln2_g, b.ln2_b, b.m2, L, stream));
    bf16_t* f1;
    o = ConvOpt(); o.act = 1;
    CHK(big(n1, b.ff1, b.ff18, o, &f1));
    o = ConvOpt(); o.resid = hs; o.ldr = D; o.gate = b.g2; o.gate_split = L; o.out = hs;
    CHK(big(f1, b.ff2, b.ff28, o, &dummy));
    c->arena.release(f1);
  }
  bf16_t* xv = hs + (size_t)L * D;
  bf16_t* a1 = A((size_t)nv * D * 2);
  DOVE_CHECK_ARG(a1, "workspace exhausted (DiT head)");
  CHK(dove_layernorm_modulate_bf16(xv, a1, nv, D, cf.dit_norm_eps, c->nf_g, c->nf_b, nullptr, 0, stream));
  CHK(dove_layernorm_modulate_bf16(a1, n1, nv, D, cf.dit_norm_eps, c->no_g, c->no_b, c->final_mod, 0, stream));
  bf16_t* po;
  CHK(linear(c, n1, nv, c->proj_out, ConvOpt(), &po, stream));
  CHK(dove_unpatchify(po, c->proj_out.cout_store(), T, cf.dit_out_channels, h, w, pt, p, v_out, out_dtype, stream));
  c->arena.release(po); c->arena.release(a1); c->arena.release(n1); c->arena.release(hs); c->arena.release(norm2);
  return DOVE_OK;
}

// process_video (ref :394-503) on device buffers: video_in [3][F][H][W] in [-1,1] (dtype), posterior noise [L][T][h][w] (noise_dtype),
// text [Ltxt][text_dim] bf16, timestep t with sqrt(alpha_t), sqrt(1 - alpha_t) of the scheduler -> video_out [3][F][H][W] in [0,1].
extern "C" int dove_sr_clip(dove_ctx* c, const void* video_in, int dtype, int F, int H, int W, const void* noise, int noise_dtype,
                            const void* text, int Ltxt, int timestep, float sqrt_alpha, float sqrt_one_minus_alpha, const dove_dit_aux* aux,
                            const dove_pre_noise* pre, void* video_out, int out_dtype, void* stream) {
  DOVE_CHECK_ARG(video_in && noise && text && video_out, "dove_sr_clip: null pointer");
  DOVE_CHECK_ARG(!pre || pre->eps, "dove_sr_clip: pre_noise without eps");
  StageGuard guard(c);
  DOVE_CHECK_ARG(F >= 1 && H % 16 == 0 && W % 16 == 0, "dove_sr_clip: H and W must be multiples of 16 (8x VAE, 2x patch)");
  CHK(ensure_ws(c, F, H, W));
  const auto& cf = c->cfg;
  const int Lc = cf.vae_latent_channels, h = H / 8, w = W / 8;
  hipStream_t s = (hipStream_t)stream;
  Tensor m;
  CHK(vae_encode_cl(c, video_in, dtype, F, H, W, &m, stream));
  const int T = m.T;
  const long long fsz = (long long)h * w;                       // elements of one latent frame of one channel
  if (c->nranks > 1) {
    // ONE clip on all ranks (BASELINE configs[2]): every rank encoded its frames; the posterior moments are gathered so that every rank
    // samples the SAME latent (the caller passes the same noise on every rank), the DiT runs sequence / head parallel and returns the whole
    // velocity everywhere, and dove_vae_decode below writes only this rank's frames of video_out (dove_shard_frames(ctx, 1, T, ...))
    RankPlan rp;
    rank_plan(c, true, F, &rp);
    std::vector<size_t> off(c->nranks), cnt(c->nranks);
    for (int j = 0; j < c->nranks; ++j) { off[j] = (size_t)rp.rank_first[j] * fsz * m.C * 2; cnt[j] = (size_t)rp.rank_count[j] * fsz * m.C * 2; }
    CHK(all_gather_v(c, (char*)m.p + off[c->rank], m.p, off.data(), cnt.data(), stream));
  }
  // sample = mean + std * noise (bf16, [L][T][h][w]), then * scaling_factor and the first-frame pad, as [T'][L][h][w] for the DiT
  bf16_t* samp = (bf16_t*)c->arena.alloc((size_t)Lc * T * fsz * 2);
  DOVE_CHECK_ARG(samp, "workspace exhausted");
  CHK(dove_posterior_sample(m.p, m.C, Lc, (long long)T * fsz, noise, noise_dtype, samp, DOVE_BF16, stream));
  free_t(c, m);
  const int ncopy = T % cf.dit_patch_t, Td = T + ncopy;
  bf16_t* lat = (bf16_t*)c->arena.alloc((size_t)Td * Lc * fsz * 2);     // [Td][L][h][w]
  bf16_t* scaled = (bf16_t*)c->arena.alloc((size_t)Lc * T * fsz * 2);
  DOVE_CHECK_ARG(lat && scaled, "workspace exhausted");
  CHK(dove_axpby(samp, samp, scaled, DOVE_BF16, (long long)Lc * T * fsz, cf.vae_scaling_factor, 0.0f, stream));   // sample * scaling_factor
  for (int t = 0; t < Td; ++t) {                                // permute(0,2,1,3,4) + prepend frame 0 `ncopy` times
    const int ts = t < ncopy ? 0 : t - ncopy;
    CHK(copy2d(lat + (long long)t * Lc * fsz, (size_t)fsz * 2, scaled + (long long)ts * fsz, (size_t)T * fsz * 2, (size_t)fsz * 2, Lc, s));
  }
  c->arena.release(samp); c->arena.release(scaled);
  if (pre) {       // --noise_step (ref :449-457): latent <- sqrt(a_n) * latent + sqrt(1 - a_n) * eps, eps [Td][L][h][w] like the latent
    bf16_t* noisy = (bf16_t*)c->arena.alloc((size_t)Td * Lc * fsz * 2);
    DOVE_CHECK_ARG(noisy, "workspace exhausted");
    const void* eps = pre->eps;
    bf16_t* eps16 = nullptr;
    if (pre->eps_dtype != DOVE_BF16) {                          // axpby takes both operands in one dtype: round eps like `.to(latent.dtype)`
      eps16 = (bf16_t*)c->arena.alloc((size_t)Td * Lc * fsz * 2);
      DOVE_CHECK_ARG(eps16, "workspace exhausted");
      hipLaunchKernelGGL(to_bf16_kernel, dim3(256), dim3(256), 0, s, pre->eps, pre->eps_dtype, (long long)Td * Lc * fsz, eps16);
      eps = eps16;
    }
    CHK(dove_axpby(lat, eps, noisy, DOVE_BF16, (long long)Td * Lc * fsz, pre->sqrt_alpha, pre->sqrt_one_minus_alpha, stream));
    c->arena.release(lat); c->arena.release(eps16);
    lat = noisy;
  }
  bf16_t* vel = (bf16_t*)c->arena.alloc((size_t)Td * Lc * fsz * 2);
  DOVE_CHECK_ARG(vel, "workspace exhausted");
  CHK(dove_dit_forward(c, lat, DOVE_BF16, Td, h, w, text, Ltxt, timestep, aux, vel, DOVE_BF16, stream));
  // get_velocity: x0 = sqrt(a) * latent - sqrt(1-a) * v
  bf16_t* x0 = (bf16_t*)c->arena.alloc((size_t)Td * Lc * fsz * 2);
  DOVE_CHECK_ARG(x0, "workspace exhausted");
  CHK(dove_axpby(lat, vel, x0, DOVE_BF16, (long long)Td * Lc * fsz, sqrt_alpha, -sqrt_one_minus_alpha, stream));
  c->arena.release(lat); c->arena.release(vel);
  // drop the padded frames, back to [L][T][h][w] for the decoder
  bf16_t* z = (bf16_t*)c->arena.alloc((size_t)Lc * T * fsz * 2);
  DOVE_CHECK_ARG(z, "workspace exhausted");
  for (int t = 0; t < T; ++t)
    CHK(copy2d(z + (long long)t * fsz, (size_t)T * fsz * 2, x0 + (long long)(t + ncopy) * Lc * fsz, (size_t)fsz * 2, (size_t)fsz * 2, Lc, s));
  c->arena.release(x0);
  CHK(dove_vae_decode(c, z, DOVE_BF16, T, h, w, 1.0f / cf.vae_scaling_factor, 1, video_out, out_dtype, stream));
  c->arena.release(z);
  return DOVE_OK;
}
