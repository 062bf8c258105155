/* libdove_hip.so -- C ABI of the MI355X-native (gfx950) operator library behind DOVE's one-step video-SR path.
 *
 * The reference (zhengchen1999/DOVE) has no FFI of its own: `inference_script.py::process_video`
 * (/root/reference/inference_script.py:394-503) drives a diffusers `CogVideoXPipeline`, and every FLOP is a
 * torch operator invoked inside diffusers modules.  The entry points below are what those modules would bind
 * instead of the torch operators -- each one cites the reference call site whose arithmetic it carries
 * (the module names are diffusers'; SURVEY.md App. A restates them).  INTEGRATION.md shows the ctypes stubs.
 *
 * Conventions
 *  - plain C: raw device pointers, sizes, a `hipStream_t` passed as `void*`; no torch / C++ types.
 *  - every function returns 0 on success, a negative DOVE_E* code otherwise; `dove_last_error()` holds the message
 *    (thread-local).  No exceptions cross the ABI.
 *  - all tensor memory is owned by the caller; launches are asynchronous on the given stream.
 *  - activations: channels-last bf16 ([T,H,W,C] for the VAE, [N,C] token-major for the DiT); statistics,
 *    biases, norm affine parameters and modulation vectors are fp32.
 */
#ifndef DOVE_HIP_H
#define DOVE_HIP_H

#include <stddef.h> /* size_t */

#ifdef __cplusplus
extern "C" {
#endif

#define DOVE_ABI_VERSION 15

/* dtype codes for boundary tensors */
#define DOVE_F32 0
#define DOVE_BF16 1

const char* dove_last_error(void);
int dove_abi_version(void);
/* name / CU count / total memory of HIP device `dev`; name_len bytes available in `name` */
int dove_device_info(int dev, char* name, int name_len, int* cu_count, long long* total_mem);

/* Implicit-GEMM convolution / linear layer (bf16 in, fp32 accumulate on MFMA, bf16 out).
 * Carries: CogVideoXCausalConv3d, Downsample3D / Upsample3D convs, resnet 1x1x1 shortcut, SpatialNorm conv_y/conv_b
 * (vae.encode / decode_latents, /root/reference/inference_script.py:408,500) and every nn.Linear of
 * CogVideoXTransformer3DModel (/root/reference/inference_script.py:483-489).
 *   x      [t_in, h_in, w_in, cin]      cin multiple of 32 (zero-padded channels)
 *   cache  [kt-1, h_in, w_in, cin] or NULL: temporal front halo = last frames of the previous frame-batch's
 *          input (diffusers `conv_cache`); NULL replicates frame 0.
 *   w      [kt*kh*kw][cout_pad][cin]    packed, cout_pad multiple of 32, padding rows zero
 *   out    [t_out, h_out, w_out, ldo]   channels [0, cout_store) written (cout_store multiple of 4)
 *   input row for output row oh and tap dh: (oh*stride + dh - pad_h) >> up   (`up`=1 folds a nearest x2 upsample)
 *   tmode (kt==1 only): 0 t_in=t, 1 t_in=t>>1, 2 t_in = t==0 ? 0 : 1+((t-1)>>1)   (Upsample3D time doubling)
 *   epilogue: v = acc + bias; act==1: gelu(tanh); resid: v = resid + (gate ? gate[class][c] * v : v), where
 *   class = (flat output pixel index < gate_split) ? 0 : 1 and gate is [2][cout_pad] fp32 (AdaLN-Zero gates). */
typedef struct dove_conv_desc {
  /* = sizeof(dove_conv_desc) of the header the CALLER was built against.  The descriptor is caller-allocated and grows with the
   * ABI (gn_partial at 2, out_f32 at 8): a binding written against an older header would hand over a shorter struct and the new
   * fields would be read from whatever follows it.  Every entry point that takes a descriptor rejects a size other than its own. */
  unsigned int struct_size;
  unsigned int reserved; /* 0 */
  const void* x;
  const void* cache;
  const void* w;
  const float* bias;
  const void* resid;
  const float* gate;
  void* out;
  int t_in, h_in, w_in, cin;
  int t_out, h_out, w_out, cout_pad, cout_store;
  int kt, kh, kw, stride, pad_h, pad_w, up, tmode, act;
  long long ldo, ldr, gate_split;
  /* optional fused GroupNorm(32) statistics of `out` (the nn.GroupNorm that consumes this conv's output in
   * CogVideoXResnetBlock3D / CogVideoXSpatialNorm3D): fp32 [dove_conv_gn_partial_rows(d)][32][2] partial (sum, sum of
   * squares) per group over the bf16-rounded stored values; reduce with dove_groupnorm_finalize_partials.  Only the
   * kernels for which dove_conv_gn_partial_rows() > 0 produce them; anything else with gn_partial != NULL is an error. */
  float* gn_partial;
  /* != 0: `out` is float [..][ldo] and receives the un-rounded fp32 accumulators (+ bias): partial sums another kernel finishes
   * (dove_conv_out_gather).  Only for plain convs (no act / resid) that dispatch to igemm_fast_kernel; an error otherwise. */
  int out_f32;
  /* nb > 1 (ABI 12): `nb` INDEPENDENT instances of this conv in one launch, laid out back to back along the frame axis -
   * x [nb * t_in, h_in, w_in, cin], out / resid [nb * t_out, ...], gn_partial instance-major (dove_conv_gn_partial_rows(d) / nb rows
   * each).  t_in / t_out stay PER-INSTANCE counts: causal temporal taps, the conv cache and tmode are evaluated inside an instance
   * (instance b's first frames read ITS cache frames, cache + b * cache_stride).  The spatial tiles of diffusers' tiled_encode /
   * tiled_decode (`--is_vae_st`, /root/reference/inference_script.py:642-645) are such instances: each tile sees zero padding at its
   * own border and keeps its own conv cache and GroupNorm scope, so same-shaped tiles run as ONE launch instead of one per tile.
   * Results are those of nb separate calls.  0 / 1 = one instance.  Not combined with gate. */
  int nb;
  /* elements between two instances' cache frames; 0 = (kt-1) * h_in * w_in * cin (a dense [nb][kt-1][h][w][cin] array).  A cache that is
   * a view of the previous frame-batch's input [nb][t_prev][h][w][cin] passes t_prev * h_in * w_in * cin. */
  long long cache_stride;
  /* optional (ABI 12), kt == 3 only: [2][kh*kw][cout_pad][cin] = the temporal weight sums w[0] + w[1] and w[0] + w[1] + w[2] (tap blocks of
   * `w`), formed in fp32 and rounded to bf16 ONCE at pack time.  Without a conv cache (cache == NULL: the first frame-batch of a clip, tile
   * or chunk) CogVideoXCausalConv3d pads the front with the REPLICATED first frame, so output frame 0 is (w0 + w1 + w2) x0 and output frame 1
   * is (w0 + w1) x0 + w2 x1: one and two temporal taps instead of three (3 % of the causal-conv MACs of a 33-frame clip).  Used by
   * conv3x3_halo4x_kernel when cache == NULL; ignored otherwise (NULL: three taps for every frame).  The sums differ from the three separate
   * products by one bf16 rounding of the summed weight - inside the operator tolerance, but a caller that needs the un-summed arithmetic
   * leaves the field NULL. */
  const void* w_first;
  /* optional (ABI 12), up == 1 with a 3x3 (kt == 1) kernel: [4 phases][2x2 taps][cout_pad][cin] = the SUB-PIXEL form of the upsample-fused conv.
   * A nearest x2 upsample followed by a 3x3 conv is, for output phase (py, px) = (oy & 1, ox & 1), a 2x2 conv on the low-res input:
   *   out[2y + py][2x + px] = sum_{a,b in {0,1}} w_sub[2 py + px][2 a + b] . in[y + py - 1 + a][x + px - 1 + b]   (zero outside the frame),
   *   w_sub[2 py + px][2 a + b] = sum of w[dh][dw] over the dh with (py + dh + 1) / 2 == py + a and the dw with (px + dw + 1) / 2 == px + b
   * (integer division), formed in fp32 and rounded to bf16 ONCE at pack time - 4 / 9 of the MACs of Upsample3D's conv.  Used when the low-res grid
   * is at least 16 x 32; NULL = the direct form (upsample folded into the addressing, all 9 taps). */
  const void* w_sub;
  /* optional (ABI 13), kt == 3 only: the caller DECLARES that the input frames of every instance come in bit-identical pairs - the frames
   * CogVideoXUpsample3D's time doubling produces (F.interpolate(scale_factor=2) along t, the first frame kept single when the frame-batch has an
   * odd length; decode_latents, /root/reference/inference_script.py:500), carried unchanged through the per-pixel SpatialNorm + SiLU to the
   * first causal conv of the next up-block.
   *   tdup == 1: pairs (0,1), (2,3), ... ; a conv cache, if any, holds an equal pair too (it is the end of the previous batch's input)
   *   tdup == 2: frame 0 single, pairs (1,2), (3,4), ... ; cache must be NULL (the head of a clip)
   * Two of the three causal taps of every output frame then read the same bits: the launch runs TWO temporal groups per frame instead of
   * three - (w0 + w1) x[t-1] + w2 x[t]  or  w0 x[t-2] + (w1 + w2) x[t]  by frame parity (frames whose three taps all read one frame: one
   * group on w_first's w0 + w1 + w2) - two thirds of the MACs.  w_pair = [2][kh*kw][cout_pad][cin]: the sums w[0] + w[1] and w[1] + w[2]
   * (tap blocks of `w`), formed in fp32 and rounded to bf16 ONCE at pack time, like w_first (which a cache-less launch needs as well).
   * Used by conv3x3_halo4x_kernel; every other kernel ignores the declaration and computes the three taps (the same function of equal frames up
   * to the rounding of the summed weights).  0 / NULL: no declaration. */
  const void* w_pair;
  int tdup;
  int reserved2; /* 0 */
} dove_conv_desc;
int dove_conv_igemm_bf16(const dove_conv_desc* d, void* stream);
/* name of the kernel this call dispatches to (one of igemm_kernel, igemm_fast_kernel, conv3x3_halo4x_kernel, gemm8p_kernel,
 * smallk_kernel) - for reporting (bench.py's per-kernel roofline) and for tests that pin which production shape
 * runs where; the rule is a pure function of the descriptor (no environment switches) */
const char* dove_conv_kernel_name(const dove_conv_desc* d);
/* Partial-tile launches a call makes besides its main launch (reporting / tests; ABI 15): 1 = the last 16 x 32 tile column is walked in
 * 32 x 16 tiles by a second launch - taken when the image ends within the first half of that column (W % 32 in 1..16) AND the two launches
 * need fewer rounds of the persistent grid than one launch (the 240 x 360 tiles of the tiled VAE).  Outputs and fused GroupNorm partial
 * sums do not depend on it (bit-identical). */
int dove_conv_partial_launches(const dove_conv_desc* d);
/* rows of gn_partial this call would write, 0 if its kernel does not fuse the statistics (caller then runs
 * dove_groupnorm_stats_bf16 on the output as before) */
long long dove_conv_gn_partial_rows(const dove_conv_desc* d);
/* stats [32][2] (mean, rstd) from partial [rows][32][2] (deterministic two-level fp64 combine).  count = elements per
 * group = npix * C/32; ws >= 256*64 DOUBLES (128 KiB, 8-byte aligned) of scratch: both levels are fp64. */
int dove_groupnorm_finalize_partials(const float* partial, long long rows, double count, float eps, void* ws, float* stats,
                                     void* stream);

/* nn.GroupNorm(32, C, eps) statistics over one frame-batch [npix, C] (diffusers CogVideoXResnetBlock3D norm1/norm2,
 * encoder.norm_out, and the norm_layer inside CogVideoXSpatialNorm3D).  stats = [32][2] (mean, rstd) fp32.
 * frame_pix = H*W of one frame (npix = frames * frame_pix; 0: treat the tensor as one frame): partial sums are formed per
 * (frame, fixed share of the frame), so a batch split into pieces of whole frames reduces to the same statistics.
 * partial_ws: >= ws_blocks*64 floats of scratch; ws_blocks >= frames * min(256, blocks a frame needs) - 4096 rows never constrain a
 * 9-frame batch; a scratch too small for the call is an ERROR (a lowered block count would make the sums depend on how many frames
 * share the call, which the split-invariance above forbids). */
int dove_groupnorm_stats_bf16(const void* x, long long npix, long long frame_pix, int C, float eps, void* partial_ws, int ws_blocks,
                              float* stats, void* stream);
/* Distributed form (dove_amd.dist, frame-batch split over a rank pair): raw per-group (sum, sum of squares) of one piece
 * as fp64 [32][2] - the ranks add their pieces' sums (and element counts) and call the finalize below, which is the same
 * fp64 mean / rstd arithmetic the single-call form ends with. */
int dove_groupnorm_sums_bf16(const void* x, long long npix, long long frame_pix, int C, void* partial_ws, int ws_blocks, double* sums,
                             void* stream);
/* count <= 0: sums has 65 entries and sums[64] is the element count (the message two ranks of a split frame-batch exchange and add) */
int dove_groupnorm_finalize_sums(const double* sums, double count, float eps, float* stats, void* stream);
/* the same raw sums from the partial rows a conv epilogue wrote (dove_conv_desc.gn_partial): no pass over the tensor */
int dove_groupnorm_sums_from_partials(const float* partial, long long rows, void* ws, double* sums, void* stream);
/* y = silu?( GN(x) [ * yb[z][0:C] + yb[z][C:2C] ] ): GroupNorm apply, optional SpatialNorm3D conditioning from the
 * [Tz,hz,wz,2C] table conv_y(zq)||conv_b(zq) on the latent grid (z = (tmap[t], h>>sshift, w>>sshift), i.e. the
 * nearest-neighbour resize of zq), optional SiLU. */
int dove_groupnorm_apply_bf16(const void* x, void* y, int T, int H, int W, int C, const float* stats,
                              const float* gamma, const float* beta, int silu, const void* yb, int hz, int wz,
                              int sshift, const int* tmap, void* stream);

/* ---- batched forms (ABI 12): `nb` independent instances back to back along the frame axis, each with its own GroupNorm scope - the
 * same-shaped spatial tiles of AutoencoderKLCogVideoX.tiled_encode / tiled_decode (`--is_vae_st`, /root/reference/inference_script.py:
 * 642-645) in ONE launch per operator instead of one per tile (dove_conv_desc.nb is the conv's form).  Each returns what nb separate
 * calls of the un-suffixed function would: the partial sums are keyed by (instance, frame, share of the frame) as before.
 *   stats_nb:             x [nb][npix][C] (npix = pixels of ONE instance, frame_pix of one frame) -> stats [nb][32][2];
 *                         partial_ws >= nb * frames * min(256, ceil(frame_pix / (8192 / (C/8)))) rows of 64 floats (an error otherwise)
 *   finalize_partials_nb: partial [nb][rows][32][2] (dove_conv_desc.gn_partial of a conv with nb instances; rows = rows of ONE
 *                         instance) -> stats [nb][32][2]; ws >= nb * 128 KiB when rows > 1024
 *   apply_nb:             x, y [nb * T, H, W, C], stats [nb][32][2], yb [nb * Tz, hz, wz, 2C], tmap [T] = frame map of ONE instance
 *   avgpool_time_nb:      x [nb][T][frame] -> y [nb][To][frame] */
int dove_groupnorm_stats_nb_bf16(const void* x, int nb, long long npix, long long frame_pix, int C, float eps, void* partial_ws,
                                 int ws_blocks, float* stats, void* stream);
int dove_groupnorm_finalize_partials_nb(const float* partial, long long rows, int nb, double count, float eps, void* ws, size_t ws_bytes,
                                        float* stats, void* stream);
int dove_groupnorm_apply_nb_bf16(const void* x, void* y, int nb, int T, int H, int W, int C, const float* stats, const float* gamma,
                                 const float* beta, int silu, const void* yb, int Tz, int hz, int wz, int sshift, const int* tmap,
                                 void* stream);
int dove_avgpool_time_nb_bf16(const void* x, int nb, int T, long long frame_elems, void* y, void* stream);

/* CogVideoXLayerNormZero / norm_final / AdaLayerNorm: y = LN(x)*gamma+beta, then *(1+scale)+shift with
 * mod = [2 row classes][shift|scale][D] fp32 (rows < split: class 0 = text); mod NULL -> plain LayerNorm. */
int dove_layernorm_modulate_bf16(const void* x, void* y, long long rows, int D, float eps, const float* gamma,
                                 const float* beta, const float* mod, long long split, void* stream);

/* Attention pre-processing of the fused QKV projection [N, 3*heads*64]: per-head LayerNorm(64) on q,k
 * (attn1.norm_q / norm_k), interleaved-pair RoPE on rows >= text_len (apply_rotary_emb), q *= qscale,
 * head-major outputs Qh,Kh [heads][Npad][64] and Vt [heads][64][Npad] (pad must be pre-zeroed).
 * v_order: key order of the Vt rows.  0 = natural.  1 = quad-swapped, what dove_attention_fwd_bf16 reads (Npad % 16 == 0): every
 * 16 consecutive keys are stored [0-3, 8-11, 4-7, 12-15] - the order in which the QK^T MFMA leaves the probabilities in a lane,
 * so that P needs no cross-lane exchange before the PV MFMA.  dove_vt_quad_swap_bf16 converts rows [rows][Npad] between the two
 * orders in place (an involution; for hosts that assemble Vt from natural-order pieces, e.g. after an all-to-all). */
/* norm2 (may be NULL): float [heads][2], receives max over the N rows of |q row|^2 and |k row|^2 of every head, taken from the STORED
 * (bf16-rounded, rotated, scaled) values - what dove_attention_fwd_bf16 needs to bound its scores.  The array is cleared and filled
 * on the stream; a host that shards the rows takes the element-wise maximum of the ranks' arrays (exact, so the sharded result
 * stays bit-identical). */
int dove_qkv_post_bf16(const void* qkv, long long N, long long Npad, int heads, int head_dim, int text_len,
                       const float* gq, const float* bq, const float* gk, const float* bk, const float* cosT,
                       const float* sinT, float qscale, float eps, void* Qh, void* Kh, void* Vt, int v_order, float* norm2, void* stream);
int dove_vt_quad_swap_bf16(void* Vt, long long rows, long long Npad, void* stream);
/* Receive side of the Q' / K' / V^T all-to-all of the sequence/head-parallel DiT (dove_amd.dist; not in the reference: one clip over
 * several GPUs): rq, rk = per source rank i the block [hloc][counts[i]][64], rv = [hloc][64][counts[i]] (natural key order), blocks in
 * rank order -> Qh, Kh [hloc][Npad][64] and Vt [hloc][64][Npad] quad-swapped with zero pad columns: the operands of
 * dove_attention_fwd_bf16 for this rank's hloc heads over all N = sum(counts) rows.  counts is a HOST array of `world` entries. */
/* norm2_out (may be NULL; ABI 12): float [hloc][2].  When given, every source block carries ONE extra row per head (rq, rk: [hloc][counts[i] + 1][64])
 * / column (rv: [hloc][64][counts[i] + 1]) behind its rows, and the extra row of the K block starts with two floats: that rank's
 * (max |q row|^2, max |k row|^2) of the head over ITS rows (dove_qkv_post_bf16's norm2).  norm2_out = their maximum over the ranks = what one
 * GPU computes over all rows, exactly - the attention's score bound rides in the all-to-all payload instead of a collective of its own. */
int dove_ulysses_place_bf16(const void* rq, const void* rk, const void* rv, const long long* counts, int world, int hloc, long long N,
                            long long Npad, void* Qh, void* Kh, void* Vt, float* norm2_out, void* stream);

/* F.scaled_dot_product_attention (no mask, non-causal) on the operands above, Vt in QUAD-SWAPPED key order; Qh carries
 * scale*log2(e).  O [N][ldo] token-major, head h at columns [64h, 64h+64).
 * norm2 (may be NULL; IN/OUT since ABI 14): float [heads][2] = max squared row norms of this call's Qh / Kh heads (dove_qkv_post_bf16).
 * Every head whose two numbers are finite runs on the software-pipelined kernel WITHOUT a softmax shift (shift-invariance: 2^s itself, the
 * constant cancels in O / l; -13 % kernel time against the running maximum at N = 18 226).  That is exact to rounding while a row's sum
 * l = sum_j 2^s_ij stays in [2^-80, 2^100] - guaranteed when the Cauchy-Schwarz bound b = 1.01 sqrt(norm2[h][0] norm2[h][1]) <= 80, and
 * checked per row otherwise: a head with a row outside the window is marked norm2[h][0] = NaN and recomputed, whole, with the running
 * maximum inside the same call.  Heads with a non-finite entry on entry and calls with norm2 == NULL use the running maximum.  On return
 * (stream order) norm2 therefore tells which kernel produced each head (dove_attention_head_paths).  The values only select the path: a
 * stale array costs time, never correctness (ABI <= 13 used b as the shift itself and needed it to describe THESE Qh / Kh). */
int dove_attention_fwd_bf16(const void* Qh, const void* Kh, const void* Vt, void* O, long long N, long long Npad,
                            int heads, int head_dim, long long ldo, float* norm2, void* stream);
/* Which kernel produced each head of a dove_attention_fwd_bf16 call: norm2_host = HOST copy of that call's norm2 taken after it completed
 * (NULL: the call had none); path[h] = 1 "attn_pipe_kernel" (no shift), 0 "attn_fwd_kernel" (running maximum).  Pure host function.
 * dove_attention_path_name(path) names them. */
int dove_attention_head_paths(const float* norm2_host, int heads, int* path);
const char* dove_attention_path_name(int path);

/* layout glue at the [B,C,T,H,W] boundary (B = 1) */
int dove_cl_from_ncthw(const void* x, int dtype, int C, long long npix, int Cp, float scale, float shift, void* y,
                       void* stream);
int dove_ncthw_from_cl(const void* x, long long ld, int C, long long npix, float scale, float shift, float lo,
                       float hi, void* y, int dtype, void* stream);
/* encoder.conv_in (3 -> 128 channels, 3x3x3) with the spatial taps moved into the input channels: [C][T][H][W] -> [T][H][W][Cp] bf16,
 * y[..][(dy*3+dx)*C + c] = x[c][t][h+dy-1][w+dx-1] * scale + shift (0 outside the frame; channels >= 9C zero).  The conv then is a (3,1,1)
 * conv on 27 real input channels (weights re-laid the same way) instead of a 3x3x3 conv on 3 real channels padded to 32. */
int dove_cl_im2col3x3_from_ncthw(const void* x, int dtype, int C, int T, int H, int W, int Cp, float scale, float shift, void* y,
                                 void* stream);
/* decoder.conv_out (128 -> 3 channels, 3x3x3) split by spatial tap: a (3,1,1) conv with 27 (padded 32) output channels
 * P[t][y][x][(dy*3+dx)*C + c] = sum_{kt,ci} x[t+kt-2][y][x][ci] w[c][ci][kt][dy][dx] (dove_conv_igemm_bf16 with out_f32 = 1: K = 3*Cin
 * instead of 27*Cin per staged pixel, and no 32/3 padding waste on the MFMAs) followed by this gather:
 * y[c][t][oy][ox] = clamp(bf16(bias[c] + sum_{dy,dx} P[t][oy+dy-1][ox+dx-1][(dy*3+dx)*C + c]) * scale + shift, lo, hi), zero padding at
 * the frame border; the bf16 rounding is the conv's own output rounding in the reference. */
int dove_conv_out_gather(const float* p, long long ldp, int T, int H, int W, int C, const float* bias, float scale, float shift, float lo,
                         float hi, void* y, int dtype, void* stream);
/* CogVideoXDownsample3D temporal average pool (odd T keeps the first frame) */
int dove_avgpool_time_bf16(const void* x, int T, long long frame_elems, void* y, void* stream);
/* DiagonalGaussianDistribution.sample(): out[c] = mean + exp(0.5*clamp(logvar,-30,20)) * noise  (ref :409) */
int dove_posterior_sample(const void* moments, long long ld, int latent_channels, long long npix, const void* noise,
                          int noise_dtype, void* out, int out_dtype, void* stream);
/* out = a*x + b*y : CogVideoXDPMScheduler.get_velocity / add_noise (ref :457,491-493) */
int dove_axpby(const void* x, const void* y, void* out, int dtype, long long n, float a, float b, void* stream);
/* CogVideoXPatchEmbed gather / transformer un-patchify between [T,C,h,w] and tokens [Nv][C*pt*p*p] */
int dove_patchify(const void* x, int dtype, int T, int C, int H, int W, int pt, int p, void* tokens, long long ld,
                  void* stream);
int dove_unpatchify(const void* tokens, long long ld, int T, int C, int H, int W, int pt, int p, void* y, int dtype,
                    void* stream);
/* The same gather for a SPATIAL TILE of diffusers' tiled_decode (`--is_vae_st`, /root/reference/inference_script.py:642-645): the tile is
 * cross-faded with its neighbours in the channels-last layout before the clip changes layout, so the result stays channels-last:
 * y[t][oy][ox][c] = bf16(bias[c] + sum ...) for c < C, 0 for C <= c < ldy; bf16 [T,H,W,ldy], C <= ldy <= 32, ldy % 4 == 0; no range map.
 * T may be nb x frames (tile-major batch): frames are independent.  (ABI 15) */
int dove_conv_out_gather_cl(const float* p, long long ldp, int T, int H, int W, int C, const float* bias, void* y, int ldy, void* stream);
/* Spatial tiles of diffusers' tiled_encode / tiled_decode as ONE tile-major batch (dove_conv_desc.nb): out [nb][nt][th][tw][C] <-
 * x [T][H][W][C] bf16 channels-last, frames [t0, t0 + nt), tile n at (oy[n], ox[n]) (HOST arrays, nb <= 64), C % 8 == 0.
 * im2col_cin > 0: x is the im2col'ed clip of dove_cl_im2col3x3_from_ncthw with C = im2col_cin input channels; the channels of taps that
 * reach outside the TILE are zeroed at the tile's border pixels (each tile sees zero padding at its own border), so that the batch equals
 * the im2col of the cropped tiles bit for bit and encoder.conv_in runs in its (3,1,1) form inside tiles too.  (ABI 15) */
int dove_tile_gather_bf16(const void* x, int H, int W, int C, int t0, int nt, int th, int tw, int nb, const int* oy, const int* ox,
                          int im2col_cin, void* out, void* stream);
/* AutoencoderKLCogVideoX.blend_v / blend_h of the spatial-tiling path (`enable_tiling`, ref :644-645): in-place linear
 * cross-fade of the first `extent` rows (axis 0) / columns (axis 1) of tile b with the last ones of its neighbour a;
 * channels-last tiles [T,H,W,ld], ld multiple of 4. */
int dove_blend_edge_bf16(const void* a, void* b, int T, int Ha, int Wa, int Hb, int Wb, int ld, int extent, int axis,
                         void* stream);
/* Script-level pre/post-processing of /root/reference/inference_script.py on the GPU:
 * preprocess: frames [F0,H0,W0,3] u8 -> [3, F0+pad_f, (H0+pad_h)*up, (W0+pad_w)*up] in [-1,1]: pad F by repeating the last frame,
 *   pad H/W with zeros bottom/right (ref :220-232), bilinear x`up` with align_corners=False (ref :672), x/255*2-1 (ref :674).
 * postprocess: video [3,F,H,W] in [0,1] -> frames [Fo,Ho,Wo,3] u8 = trunc(clamp(x*255,0,255)), cropping the padding
 *   (ref :238-246, :124). */
int dove_preprocess_u8(const void* frames, int F0, int H0, int W0, int pad_f, int pad_h, int pad_w, int upscale, void* out,
                       int out_dtype, void* stream);
int dove_postprocess_u8(const void* video, int dtype, int F, int H, int W, int Fo, int Ho, int Wo, void* out, void* stream);
/* M = 1 linear with optional SiLU on the input (time_embedding MLP, norm*.linear modulation vectors) */
int dove_gemv_bf16(const void* W, const float* bias, const float* x, int in_features, int out_features, int act_in,
                   float* y, void* stream);

/* ---- MXFP8 linears (BASELINE configs[4]: fp8 MFMA path of the DiT, PSNR-gated against the bf16 path) --------------------
 * OCP microscaling FP8: e4m3fn elements, one E8M0 (power-of-two) scale per 32 consecutive K elements of a row, for BOTH
 * operands of an nn.Linear of CogVideoXBlock (attn1.to_q/k/v fused, attn1.to_out.0, ff.net.0.proj, ff.net.2;
 * /root/reference/inference_script.py:483-489).  The scales are applied inside v_mfma_scale_f32_32x32x64_f8f6f4.
 * dove_mx_quant_bf16: x bf16 [rows][K] (K % 256 == 0) -> q u8 [rows][K] and scales u32 [K/256][rows][2]; word (c, r, h),
 *   byte u = E8M0 scale of row r's block 8c + 2u + h; scale = 2^ceil(log2(amax/448)) (no clipping), q = RNE_e4m3(x/scale).
 * dove_linear_mxfp8: out [M][ldo] bf16 = epilogue((xq.xs) (wq.ws)^T + bias) with the epilogue of dove_conv_igemm_bf16
 *   (act 1 = GELU(tanh); resid / gate [2][N] fp32 / gate_split as there); N % 256 == 0, K % 256 == 0. */
int dove_mx_quant_bf16(const void* x, long long rows, int K, void* q, void* scales, void* stream);
int dove_linear_mxfp8(const void* xq, const void* xs, const void* wq, const void* ws, const float* bias, const void* resid,
                      const float* gate, void* out, long long M, int N, int K, long long ldo, long long ldr, long long gate_split,
                      int act, void* stream);

/* MXFP8 attention (csrc/attention_mx.hip; same configs[4] variant): dove_qkv_post_bf16's pre-processing with e4m3 outputs -
 * Q8 [heads][Npad][64] = e4m3(q * qscale * 8) (fixed block scale 2^-3), K8 [heads][Npad][64] = e4m3(k) (scale 2^0),
 * V8t [heads][64][Npad] e4m3 with one E8M0 scale per (d, 32 consecutive keys): Vs u8 [heads][Npad/64][64][2].
 * Every row of the padded buffers is written (pad = zeros); Npad % 128 == 0.
 * dove_attention_fwd_mxfp8: softmax(Q K^T) V on those operands, probabilities quantised per (query, 64-key tile) in
 * registers; O as dove_attention_fwd_bf16. */
int dove_qkv_post_mxfp8(const void* qkv, long long N, long long Npad, int heads, int head_dim, int text_len,
                        const float* gq, const float* bq, const float* gk, const float* bk, const float* cosT,
                        const float* sinT, float qscale, float eps, void* Q8, void* K8, void* V8t, void* Vs,
                        void* stream);
int dove_attention_fwd_mxfp8(const void* Q8, const void* K8, const void* V8t, const void* Vs, void* O, long long N,
                             long long Npad, int heads, int head_dim, long long ldo, void* stream);

/* ---- T5 text encoder operators (non-empty prompts only: `pipe.text_encoder(ids)[0]`, /root/reference/inference_script.py:429-444;
 * transformers' T5EncoderModel of CogVideoX1.5: T5-v1.1-XXL, 24 blocks, d_model 4096, 64 heads x 64, gated-GELU d_ff 10240) ----
 * T5LayerNorm: y = weight * x * rsqrt(mean(x^2) + eps) (fp32 statistics; no mean subtraction, no bias) */
int dove_rmsnorm_bf16(const void* x, void* y, long long rows, int D, float eps, const float* weight, void* stream);
/* T5DenseGatedActDense: y [M][F] = gelu_new(x[:, :F]) * x[:, F:2F] on the fused wi_0 || wi_1 projection */
int dove_gated_gelu_bf16(const void* x, void* y, long long M, int F, void* stream);
/* T5Attention (encoder self-attention, no mask): out [N][ldo] = softmax(q k^T + bias[h]) v per head; q/k/v token-major with
 * row stride ld (views into the fused projection), head h at columns [64h, 64h+64); NO 1/sqrt(d) scaling; bias [H][N][N]
 * fp32 = relative_attention_bias gathered by bucket; N <= 1024 */
int dove_attention_bias_bf16(const void* q, const void* k, const void* v, long long ld, const float* bias, void* out, long long ldo,
                             int N, int heads, int head_dim, void* stream);

/* ---- graph-level entry points: a context owns packed weights + one workspace arena and runs whole stages ----
 * One context per (process, device); NOT thread-safe; calls are asynchronous on the given stream except where a host-side
 * table has to be uploaded (first call with a new shape / timestep).  Weight tensors passed to dove_set_weight are BORROWED until
 * dove_finalize_weights returns (it repacks them into library-owned memory: implicit-GEMM layout for every conv / linear,
 * fused QKV, SpatialNorm conv_y || conv_b, fp32 norm parameters).  Names are diffusers' state-dict names of
 * AutoencoderKLCogVideoX (`encoder.*`, `decoder.*`) and CogVideoXTransformer3DModel (everything else); SURVEY.md App. E. */
typedef struct dove_ctx dove_ctx;
typedef struct dove_model_config {
  /* vae/config.json */
  int vae_in_channels, vae_out_channels, vae_latent_channels, vae_num_blocks, vae_block_out_channels[8], vae_layers_per_block,
      vae_temporal_compression, vae_enc_batch /* num_sample_frames_batch_size */, vae_dec_batch /* num_latent_frames_batch_size */;
  float vae_norm_eps, vae_scaling_factor;
  /* transformer/config.json */
  int dit_heads, dit_head_dim, dit_num_layers, dit_in_channels, dit_out_channels, dit_patch, dit_patch_t, dit_text_dim,
      dit_time_embed_dim, dit_max_text, dit_flip_sin_to_cos;
  float dit_norm_eps, dit_freq_shift;
} dove_model_config;
/* optional precomputed inputs of the DiT (all device pointers, may be NULL): RoPE tables [Nv][64] fp32 as
 * `prepare_rotary_positional_embeddings` returns them (ref :364-392) and the bf16-rounded sinusoidal timestep projection [D] fp32;
 * NULL = computed inside (host libm + upload, cached per shape / timestep) */
typedef struct dove_dit_aux { const float* rope_cos; const float* rope_sin; const float* timestep_proj; } dove_dit_aux;
int dove_create(int hip_device, const dove_model_config* cfg, dove_ctx** out);
/* ---- one clip on several GPUs (BASELINE configs[2]) ----
 * One context per rank (process or thread, one GPU each).  With a communicator set,
 *  - dove_vae_encode / dove_vae_decode run only THIS rank's share of the frame-batches (diffusers' num_sample_frames_batch_size /
 *    num_latent_frames_batch_size batching): every CogVideoXCausalConv3d of the rank's first work item receives its conv_cache (the last
 *    kt-1 input frames the previous item would have left) from rank-1, and sends its own to rank+1 after the rank's last - point-to-point, in
 *    layer order, on the context's own receive / send streams (ABI 14): the transport callbacks are handed THOSE streams for halos, the
 *    caller's stream only waits on events (receives are pre-posted from the second call with a given stage and shape on).  With at most as many ranks as frame-batches the items are whole batches; with MORE ranks (ABI 12)
 *    batches are split into PAIRED PIECES on consecutive ranks - 8 ranks on a 33-frame clip decode 5,4,4,4,4,4,4,4 frames, BASELINE's
 *    "frame-chunk = 4" - and every GroupNorm of a piece swaps 65 doubles (per-group sum, sum of squares, element count) with its partner, so the
 *    statistics are the whole batch's.  Only the frames dove_shard_frames reports are written to the output buffer;
 *  - dove_dit_forward (ABI 12) shards ONE sample over the ranks: rows of the token-major residual stream for every row-local operator,
 *    heads for attention, one all-to-all of Q' / K' / V^T and one of the attention output per layer (the score bound rides in the K blocks);
 *    v_out is complete on every rank.  dit_heads must be a multiple of nranks;
 *  - dove_sr_clip (ABI 12) chains them: encode, gather of the posterior moments, the sharded DiT, decode - every rank passes the SAME clip,
 *    noise and text and gets ITS frames of video_out (dove_shard_frames(ctx, 1, T, ...)); gathering them is the caller's.
 * Every result is bit-identical to the single-GPU call (tests/test_graph_gpu.py plays 2 - 8 ranks as threads on one GPU).
 * dove_comm_init: RCCL transport (librccl opened at run time; the 128-byte id from dove_comm_unique_id on rank 0, distributed by the host;
 * every symmetric exchange is one ncclGroup; the halos of the two directions travel on two further communicators made with ncclCommSplit).
 * dove_comm_init_custom: any transport - send / recv of `bytes` device bytes to / from rank `peer`, ordered on `stream`, matched in order per
 * (source, destination) pair.  `stream` is the caller's for the symmetric exchanges and one of the context's two internal streams for
 * halos: a transport that hands data from the sender's stream to the receiver's must order the two itself (an event per message).  Because
 * a custom transport has no channels, halo receives are posted ahead only for work items whose GroupNorm partner is not rank - 1 (the pair
 * sums would share the halos' message order); under RCCL the halos have communicators of their own and are always posted ahead. */
typedef int (*dove_xfer_fn)(void* user, int peer, void* dev_ptr, size_t bytes, void* stream);
/* Bracket of ONE exchange (ABI 12).  The paired-piece VAE swaps 65 doubles with a partner per GroupNorm and the sharded DiT runs all-to-alls:
 * symmetric patterns in which every rank sends and receives.  A transport whose send rendezvous with the peer's receive (RCCL) must see the
 * calls of such an exchange as one group (dove_comm_init does this with ncclGroupStart / ncclGroupEnd); a custom transport either BUFFERS its
 * sends (then no bracket is needed: the library issues all sends of an exchange before its receives, and in a pair swap the lower rank
 * sends first) or provides the two callbacks here. */
typedef int (*dove_group_fn)(void* user, void* stream);
int dove_comm_set_group(dove_ctx* ctx, dove_group_fn begin, dove_group_fn end);
int dove_comm_unique_id(void* out128);
int dove_comm_init(dove_ctx* ctx, const void* nccl_unique_id, int rank, int nranks);
int dove_comm_init_custom(dove_ctx* ctx, int rank, int nranks, dove_xfer_fn send, dove_xfer_fn recv, void* user);
void dove_comm_destroy(dove_ctx* ctx);
/* stage 0 = dove_vae_encode (n = pixel frames F -> latent frames), stage 1 = dove_vae_decode (n = latent frames T -> pixel frames) */
int dove_shard_frames(dove_ctx* ctx, int stage, int n, int* first, int* count);
/* number of ranks that get work for this stage and clip length under the context's communicator (whole batches, or paired pieces when there
 * are more ranks than batches); on a single-rank context: the most ranks that could (two per splittable frame-batch) */
int dove_comm_useful_ranks(dove_ctx* ctx, int stage, int n);
void dove_destroy(dove_ctx* ctx);
/* ---- options of the graph level: the reference's switches that change what a stage computes ----
 * DOVE_OPT_VAE_TILING (value 0 / 1): pipe.vae.enable_tiling() (`--is_vae_st`, /root/reference/inference_script.py:643-645; every
 *   published number of the reference ran with it): dove_vae_encode / dove_vae_decode / dove_sr_clip run diffusers' spatial tiling
 *   whenever the clip is larger than one tile - tile = sample/2 px (sample/16 latents), strides 5/6 and 4/5 of it, linear
 *   cross-fade of the overlaps, every tile with its own GroupNorm scope and conv caches.  Not combined with a multi-rank context.
 * DOVE_OPT_VAE_SAMPLE_HEIGHT / _WIDTH (px): vae/config.json sample_height / sample_width (defaults 480 / 720).
 * DOVE_OPT_DIT_LINEAR_MXFP8 (0 / 1; BASELINE configs[4], not a reference option; never the headline dtype): attn1.to_q/k/v,
 *   attn1.to_out.0, ff.net.0.proj and ff.net.2 of every block in MXFP8.  Choose BEFORE dove_finalize_weights: the weights are
 *   quantised there and their bf16 copies dropped.
 * DOVE_OPT_DIT_ATTN_MXFP8 (0 / 1; same variant): attention on dove_qkv_post_mxfp8 / dove_attention_fwd_mxfp8; any time.
 * DOVE_OPT_WEIGHT_SUMS: see below.
 * Stage results with an option set are bit-identical to the Python facade with the same switch (tests/test_graph_gpu.py). */
#define DOVE_OPT_VAE_TILING 1
#define DOVE_OPT_VAE_SAMPLE_HEIGHT 2
#define DOVE_OPT_VAE_SAMPLE_WIDTH 3
#define DOVE_OPT_DIT_LINEAR_MXFP8 4
#define DOVE_OPT_DIT_ATTN_MXFP8 5
/* DOVE_OPT_WEIGHT_SUMS (0 / 1, default 1; any time): 0 = the VAE never hands the pack-time weight sums (dove_conv_desc.w_first / w_sub /
 * w_pair) to its convolutions - every launch computes the reference's per-tap arithmetic (+7.6 % conv MACs), for a caller who wants a
 * checkpoint validated without the one extra bf16 rounding of the summed weights.  The Python facade's switch: pipe.vae.weight_sums. */
#define DOVE_OPT_WEIGHT_SUMS 6
/* DOVE_OPT_VAE_STREAMS (1 / 2, default 2; any time; ABI 14): 2 = the frame-batches of an un-tiled, single-rank dove_vae_encode / dove_vae_decode
 * alternate between the caller's stream and one internal stream, ordered by one event per causal conv (the conv cache is the only dependency
 * between frame-batches); the caller's stream continues behind both when the stage returns.  Same kernels on the same inputs: bit-identical
 * to 1; +1.5 % per clip; the arena's high water grows by about a half (dove_workspace_bytes accounts for it).  The Python facade's switch:
 * pipe.vae.n_streams. */
#define DOVE_OPT_VAE_STREAMS 7
/* read-only counters of the LAST dove_vae_encode / dove_vae_decode of a multi-rank context (dove_get_option; ABI 14): halos this rank received
 * from receives posted before the stage's first kernel / posted where they were consumed (the first pass over a (stage, shape) records the
 * list, later passes pre-post it), halos sent; DOVE_STAT_HALO_COMMUNICATORS: 3 = RCCL with one communicator per halo direction + the main
 * one, 1 = RCCL without ncclCommSplit (one communicator, nothing pre-posted), 0 = custom transport / single rank. */
#define DOVE_STAT_HALO_PREPOSTED 100
#define DOVE_STAT_HALO_BLOCKING 101
#define DOVE_STAT_HALO_SENT 102
#define DOVE_STAT_HALO_COMMUNICATORS 103
int dove_set_option(dove_ctx* ctx, int option, long long value);
long long dove_get_option(dove_ctx* ctx, int option); /* -1: unknown option */
int dove_set_weight(dove_ctx* ctx, const char* name, const void* dev_ptr, const long long* shape, int ndim, int dtype);
int dove_finalize_weights(dove_ctx* ctx);
/* arena bytes a [3,F,H,W] clip needs (upper bound); dove_set_workspace(ctx, NULL, n) lets the library allocate, a non-NULL pointer
 * lends caller memory; with neither the first stage call allocates dove_workspace_bytes itself */
size_t dove_workspace_bytes(dove_ctx* ctx, int F, int H, int W);
int dove_set_workspace(dove_ctx* ctx, void* dev_ptr, size_t bytes);
size_t dove_workspace_high_water(dove_ctx* ctx);
/* pipe.vae.encode(x).latent_dist.parameters (ref :408): x [3][F][H][W] -> moments [2L][T][H/8][W/8] */
int dove_vae_encode(dove_ctx* ctx, const void* x, int dtype, int F, int H, int W, void* moments_out, int out_dtype, void* stream);
/* pipe.transformer(...)[0] for one sample (ref :483-489): hidden [T][C][h][w], text [L][text_dim] bf16 -> v [T][C][h][w] */
int dove_dit_forward(dove_ctx* ctx, const void* hidden, int dtype, int T, int h, int w, const void* text, int L, int timestep,
                     const dove_dit_aux* aux, void* v_out, int out_dtype, void* stream);
/* pipe.vae.decode(z * prescale).sample (ref :500-501; range01 != 0 fuses (x*0.5+0.5).clamp(0,1)): z [L][T][h][w] -> [3][F][8h][8w],
 * F = dove_vae_decode_num_frames(T) = 1 + 4(T-1) for the odd T of 8N+1-frame clips */
int dove_vae_decode_num_frames(dove_ctx* ctx, int T);
int dove_vae_decode(dove_ctx* ctx, const void* z, int dtype, int T, int h, int w, float prescale, int range01, void* video_out,
                    int out_dtype, void* stream);
/* `--noise_step n` (ref :449-457; 0 = off = NULL): the latent handed to the DiT is first noised,
 * latent <- sqrt_alpha * latent + sqrt_one_minus_alpha * eps with the scheduler's alpha_n (cast to the latent dtype first, like
 * diffusers' add_noise) and eps [T'][L][h][w] (the DiT's input layout, T' = T + T % patch_t) drawn by the caller. */
typedef struct dove_pre_noise { const void* eps; int eps_dtype; float sqrt_alpha, sqrt_one_minus_alpha; } dove_pre_noise;
/* process_video (ref :394-503) with the posterior noise injected and the scheduler's sqrt(alpha_t), sqrt(1 - alpha_t) (alpha cast
 * to bf16 first, like diffusers): video_in [3][F][H][W] in [-1,1] -> video_out [3][F][H][W] in [0,1]; pre_noise may be NULL */
int dove_sr_clip(dove_ctx* ctx, const void* video_in, int dtype, int F, int H, int W, const void* noise, int noise_dtype, const void* text,
                 int text_len, int timestep, float sqrt_alpha, float sqrt_one_minus_alpha, const dove_dit_aux* aux,
                 const dove_pre_noise* pre_noise, void* video_out, int out_dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DOVE_HIP_H */
