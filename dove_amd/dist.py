"""Multi-GPU execution of the one-step SR path: one process per GPU, torch.distributed ("nccl" = RCCL over
xGMI on the GPU box, "gloo" in CPU tests).  SURVEY.md 8(e).

(A) Chunk farm -- the reference's own semantics: every (time-chunk x spatial-tile) of
    /root/reference/inference_script.py:690-703 is an independent ``process_video`` call, so work items are dealt
    round-robin to ranks; NO data-path collective, only the final stitch (a sum of disjoint pieces).  This is what
    ``bench.py --gpus N`` scales (one clip per GPU, weak scaling).

(B) Halo-exact single clip -- not in the reference; the distributed form of diffusers' ``conv_cache``.  The VAE's
    frame-batches (8 px-frames / 2 latent frames) are split into contiguous groups, one
    group per rank; every k_t=3 CausalConv3d of the first local batch receives its 2-frame temporal halo (the last
    two INPUT frames of that conv in the previous rank's last batch) with a point-to-point recv from rank-1, and
    the last local batch sends its own to rank+1 -- neighbour traffic on ONE xGMI link, in layer order, so ranks
    run as a wavefront skewed by one layer.  GroupNorm statistics are per frame-batch; with at most as many ranks as
    frame-batches batches are never split, no reduction is needed and results are bit-identical to the single-GPU path.
    With more ranks (8 GPUs on a 33-frame clip: BASELINE's "frame-chunk = 4") a batch is split into two pieces on a rank
    pair which combines its GroupNorm sums with one send/recv per norm (``plan_pieces``).
    The DiT attends over all tokens of the clip (not frame-separable): moments are all-gathered and the DiT runs either
    replicated or (C) sequence/head-parallel.

(C) Ulysses-style DiT for the single-clip mode: the [N, 3072] residual stream is sharded by ROWS (tokens; text rows
    first) for everything that is row-local - LayerNormZero, the QKV / out / FFN linears with their gated residuals,
    QK-LayerNorm + RoPE - and by HEADS for attention (48 heads / R ranks): one all_to_all moves Q', K', V^T from
    "my rows, all heads" to "all rows, my heads" before the flash-attention kernel and one moves its output back, per
    layer (4 x N x 3072 x 2 B / R per rank and layer over xGMI).  Every row and every head sees exactly the arithmetic of
    the single-GPU path, so the result is bit-identical (gloo test with 2 and 3 ranks); `process_video_sharded` chains
    (B) encode -> (C) DiT -> (B) decode.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import tiling
from .inference import process_video
from .vae import DiagonalGaussianDistribution, frame_batches


# ---- (A) chunk farm ---------------------------------------------------------------------------------
def owns(i: int, n: int, rank: int, world: int) -> bool:
    """Round-robin ownership of work item i of n."""
    return i % world == rank


@torch.no_grad()
def run_clip_distributed(pipe, video, *, group=None, chunk_len=0, overlap_t=8, tile_size_hw=(0, 0), overlap_hw=(32, 32),
                         empty_prompt_embedding=None, sr_noise_step=399, seeds=None):
    """Reference chunk/tile loop with the work items sharded over the ranks of ``group``; every rank returns the full
    stitched [1,3,F,H,W] fp32 host tensor (pieces are disjoint, so the merge is an all-reduce SUM) and the write
    count, on which the reference's exact-once coverage check is applied."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    items = tiling.plan(video.shape, chunk_len, overlap_t, tile_size_hw, overlap_hw)
    out = torch.zeros(video.shape, dtype=torch.float32)
    wc = torch.zeros(video.shape, dtype=torch.int32)
    for i, ((t0, t1, h0, h1, w0, w1), region) in enumerate(items):
        if not owns(i, len(items), rank, world):
            continue
        gen = None
        if seeds is not None:   # per-item generator so the result does not depend on which rank ran the item
            gen = torch.Generator(device=pipe.vae.device).manual_seed(int(seeds) + i)
        piece = process_video(pipe, video[:, :, t0:t1, h0:h1, w0:w1], sr_noise_step=sr_noise_step,
                              empty_prompt_embedding=empty_prompt_embedding, generator=gen)
        tiling.stitch(out, wc, piece.float().cpu(), region)
    backend = dist.get_backend(group)
    if backend == "nccl":
        dev = pipe.vae.device
        o, w = out.to(dev), wc.to(dev)
        dist.all_reduce(o, group=group)
        dist.all_reduce(w, group=group)
        out, wc = o.cpu(), w.cpu()
    else:
        dist.all_reduce(out, group=group)
        dist.all_reduce(wc, group=group)
    tiling.check_coverage(wc)
    return out, wc


# ---- (B) halo-exact VAE -------------------------------------------------------------------------------
def split_batches(batches, world):
    """Contiguous groups of frame-batches, one per rank (earlier ranks get the extras)."""
    n = len(batches)
    base, extra = divmod(n, world)
    out, s = [], 0
    for r in range(world):
        k = base + (1 if r < extra else 0)
        out.append(batches[s:s + k])
        s += k
    return out


class HaloCache(dict):
    """conv_cache dict whose misses on the first local batch are filled by a recv from rank-1 and whose final
    entries are sent to rank+1.  ``phase`` is set by the VAE driver loop: 'first' / 'last' / 'both' / 'mid'."""

    def __init__(self, group, rank, world):
        super().__init__()
        self.group, self.rank, self.world = group, rank, world
        self.phase = "mid"
        self.bytes_sent = 0

    def _wire(self, t):
        return t.view(torch.uint8) if dist.get_backend(self.group) != "nccl" else t

    def fetch(self, name, like_shape, device):
        """Halo for conv ``name`` of the first local batch (None on rank 0: replicate-first-frame padding)."""
        if self.phase in ("first", "both") and self.rank > 0:
            buf = torch.empty(like_shape, dtype=torch.bfloat16, device=device)
            dist.recv(self._wire(buf), src=dist.get_global_rank(self.group, self.rank - 1) if self.group else self.rank - 1,
                      group=self.group)
            return buf
        return self.get(name)

    def publish(self, name, new):
        self[name] = new
        if self.phase in ("last", "both") and self.rank < self.world - 1:
            dist.send(self._wire(new.contiguous()), dst=dist.get_global_rank(self.group, self.rank + 1) if self.group else self.rank + 1,
                      group=self.group)
            self.bytes_sent += new.numel() * 2


def plan_pieces(batches, world, kind):
    """Work list per rank for the halo-exact VAE.  With at most as many ranks as frame-batches every rank gets a contiguous
    group of whole batches (``split_batches``).  With more ranks, batches are split in two PIECES handed to consecutive
    ranks (a rank pair): the conv halos flow rank -> rank+1 exactly as between batches, the GroupNorm statistics of the
    batch are combined across the pair, and Upsample3D's first-frame rule is told which piece starts the batch.
    Split points keep every temporal 2:1 pooling pair inside one piece: px-frame batches (``kind == "enc"``) of 8k(+1)
    frames split at 4k(+1); latent batches (``kind == "dec"``) of 2 or 3 frames split before the last frame.
    Returns (per-rank list of dicts {s, e, role, partner, odd}, number of active ranks)."""
    nb = len(batches)
    if world <= nb:
        groups = split_batches(batches, world)
        return [[dict(s=s, e=e, role=None, partner=None) for s, e in g_] for g_ in groups], sum(1 for g_ in groups if g_)
    out, r = [[] for _ in range(world)], 0
    for i, (s, e) in enumerate(batches):
        n = e - s
        spare = (world - r) - (nb - i)               # ranks we can still spend on splitting
        if kind == "enc":
            head = (n % 2) + 4 * ((n - n % 2) // 8)
            can = n - (n % 2) >= 8 and (n - n % 2) % 8 == 0
        else:
            head = n - 1
            can = n >= 2
        if spare >= 1 and can:
            odd = n % 2 == 1
            out[r].append(dict(s=s, e=s + head, role="head" if odd else "tail", partner=r + 1, lower=True))
            out[r + 1].append(dict(s=s + head, e=e, role="tail", partner=r, lower=False))
            r += 2
        else:
            out[r].append(dict(s=s, e=e, role=None, partner=None))
            r += 1
    return out, r


def _run_sharded(vae, x_cl, batches, world, rank, group, fn, kind="enc"):
    """Run ``fn(tensor, cache)`` over this rank's frame-batches or pieces of frame-batches (``plan_pieces``)."""
    from . import ops
    plan, active = plan_pieces(batches, world, kind)
    mine = plan[rank]
    cache = HaloCache(group, rank, active)
    outs = []

    def gsrc(r):
        return dist.get_global_rank(group, r) if group else r

    for i, pc in enumerate(mine):
        first, last = i == 0, i == len(mine) - 1
        cache.phase = "both" if first and last else ("first" if first else ("last" if last else "mid"))
        if pc["partner"] is not None:
            partner, lower = pc["partner"], pc["lower"]

            def hook(x, partner=partner, lower=lower):
                # whole-batch GroupNorm statistics = my piece's (sum, sumsq, count) + the partner's, same fp64 finalize
                sums = ops.groupnorm_sums_of(x)          # from the producing conv's epilogue rows when it wrote any
                cnt = float(x.numel() // 32)
                mine_msg = torch.cat([sums.reshape(-1), torch.tensor([cnt], dtype=torch.float64, device=sums.device)])
                theirs = torch.empty_like(mine_msg)
                if lower:
                    dist.send(mine_msg, dst=gsrc(partner), group=group)
                    dist.recv(theirs, src=gsrc(partner), group=group)
                else:
                    dist.recv(theirs, src=gsrc(partner), group=group)
                    dist.send(mine_msg, dst=gsrc(partner), group=group)
                a, b = (mine_msg, theirs) if lower else (theirs, mine_msg)      # same summation order on both ranks
                tot = a + b
                return ops.groupnorm_from_sums(tot[:64].reshape(32, 2).contiguous(), float(tot[64]), vae.eps)

            vae._gn_hook, vae._piece_role = hook, pc["role"]
        try:
            outs.append(fn(x_cl[pc["s"]:pc["e"]], cache))
        finally:
            vae._gn_hook, vae._piece_role = None, None
    return outs, cache


def _gather_time(parts, group, world, device):
    """All-gather variable-length [T_r, ...] tensors along T (every rank gets the whole clip)."""
    local = torch.cat(parts, dim=0) if parts else None
    shape = [torch.zeros(4, dtype=torch.int64, device=device) for _ in range(world)]
    mine = torch.tensor(list(local.shape) if local is not None else [0, 0, 0, 0], dtype=torch.int64, device=device)
    dist.all_gather(shape, mine, group=group)
    tail = next(tuple(int(v) for v in s[1:]) for s in shape if int(s[0]) > 0)
    tmax = max(int(s[0]) for s in shape)
    pad = torch.zeros((tmax,) + tail[:-1] + (tail[-1] * 2,), dtype=torch.uint8, device=device)   # bf16 as bytes on the wire
    if local is not None:
        pad[: local.shape[0]] = local.contiguous().view(torch.uint8)
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([b[: int(s[0])] for b, s in zip(bufs, shape)], dim=0).view(torch.bfloat16)


def _bcast_from_first(t, group):
    """Every rank ends up with rank 0's tensor (bf16 moves as bytes: gloo has no bf16 wire type)."""
    src = dist.get_global_rank(group, 0) if group else 0
    wire = t.contiguous()
    dist.broadcast(wire.view(torch.uint8) if wire.dtype == torch.bfloat16 else wire, src=src, group=group)
    return wire


class _SharedPosterior(DiagonalGaussianDistribution):
    """Posterior of the sharded encode.  Every rank patchifies only ITS rows of the sampled latent, so all ranks must
    hold the same sample: a draw from each rank's own device RNG (what ``sample()`` does by default, like diffusers) would
    silently mix shards of different samples.  The noise - injected, drawn from ``generator`` or from the global RNG - is
    therefore always rank 0's, broadcast over the group."""

    def __init__(self, moments_cl, latent_channels, dtype, group):
        super().__init__(moments_cl, latent_channels, dtype)
        self._group = group

    def sample(self, generator=None, noise=None):
        T, h, w, _ = self._m[0].shape
        if noise is None:
            noise = torch.randn(len(self._m), self._L, T, h, w, generator=generator, device=self._m[0].device, dtype=self._dtype)
        noise = _bcast_from_first(noise.to(self._m[0].device), self._group)
        return super().sample(noise=noise)


class _SharedScheduler:
    """``pipe.scheduler`` for the sharded path: the `--noise_step` pre-noising draws eps with torch.randn_like on every
    rank (ref :449-457); rank 0's draw is the one all ranks use."""

    def __init__(self, sched, group):
        self._sched, self._group, self.config = sched, group, sched.config

    def get_velocity(self, *a, **k):
        return self._sched.get_velocity(*a, **k)

    def add_noise(self, original_samples, noise, timesteps):
        return self._sched.add_noise(original_samples, _bcast_from_first(noise, self._group), timesteps)


@torch.no_grad()
def encode_sharded(vae, x, group=None):
    """vae.encode with frame-batches sharded over ranks + temporal halo exchange.  Every rank returns the full
    posterior (moments all-gathered), bit-identical to ``vae.encode(x)``."""
    from . import ops
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    assert x.shape[0] == 1, "sharded path runs B = 1 (the reference's batch)"
    x = x.to(vae.device).contiguous()
    x_cl = ops.cl_from_ncthw(x[0], vae.pc["encoder.conv_in"].cin_pad)
    outs, cache = _run_sharded(vae, x_cl, frame_batches(x_cl.shape[0], vae.enc_batch), world, rank, group, vae._encoder)
    moments = _gather_time(outs, group, world, vae.device)
    vae.last_halo_bytes = cache.bytes_sent
    return _SharedPosterior([moments], vae.lat, vae.dtype, group)


@torch.no_grad()
def decode_sharded(vae, z, group=None, _range01=False, _prescale=1.0):
    """vae.decode with latent frame-batches sharded over ranks; every rank returns the full [1,3,F,H,W] video."""
    from . import ops
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    assert z.shape[0] == 1
    z = z.to(vae.device).contiguous()
    z_cl = ops.cl_from_ncthw(z[0], vae.pc["decoder.conv_in"].cin_pad, scale=_prescale)
    outs, cache = _run_sharded(vae, z_cl, frame_batches(z_cl.shape[0], vae.dec_batch), world, rank, group, vae._decoder, "dec")
    full = _gather_time(outs, group, world, vae.device)          # [F,H,W,4] channels-last
    post = dict(scale=0.5, shift=0.5, lo=0.0, hi=1.0) if _range01 else {}
    vae.last_halo_bytes = cache.bytes_sent
    return ops.ncthw_from_cl(full, vae.config["out_channels"], vae.dtype, **post)[None]


# ---- (C) sequence / head parallel DiT -------------------------------------------------------------------
def _row_bounds(n, world):
    return [(i * n) // world for i in range(world + 1)]


def _a2a(out_splits, inp, in_splits, group):
    """all_to_all_single on a flat bf16 tensor (moved as bytes: gloo has no bf16 wire type)."""
    out = torch.empty(sum(out_splits), dtype=inp.dtype, device=inp.device)
    dist.all_to_all_single(out.view(torch.uint8), inp.contiguous().view(torch.uint8), [2 * v for v in out_splits],
                           [2 * v for v in in_splits], group=group)
    return out


@torch.no_grad()
def dit_forward_ulysses(tr, hidden, text, t, rope, group=None):
    """One sample of CogVideoXTransformer3DModel.forward (same arguments as ``tr._forward_one``) with rows sharded over the
    ranks of ``group`` and attention head-parallel.  Every rank returns the full [T,C,H,W] prediction."""
    import math

    from . import ops
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    D, heads = tr.D, tr.heads
    if heads % world:
        raise ValueError(f"{heads} attention heads do not split over {world} ranks")
    hloc = heads // world
    dev = tr.device
    T, Cc, H, W = hidden.shape
    p, pt = tr.p, tr.pt
    hidden = hidden.to(dev).contiguous()
    text = text.to(dev, torch.bfloat16).contiguous()
    cos, sin = (r.to(dev, torch.float32).contiguous() for r in rope)
    Lt = text.shape[0]
    nv = (T // pt) * (H // p) * (W // p)
    N = Lt + nv
    npad = (N + 127) // 128 * 128
    bounds = _row_bounds(N, world)
    r0, r1 = bounds[rank], bounds[rank + 1]
    counts = [bounds[i + 1] - bounds[i] for i in range(world)]
    nloc = r1 - r0
    assert nloc > 0, "more ranks than tokens"
    lt_loc = max(0, min(r1, Lt) - r0)                          # local rows that are text rows (they come first)
    v0, v1 = max(r0, Lt) - Lt, max(r1, Lt) - Lt                # local video rows [v0, v1)
    blocks_mod, final_mod = tr._modulation(t)

    hs = torch.empty(nloc, D, dtype=torch.bfloat16, device=dev)
    if lt_loc:
        ops.linear(text[r0:r0 + lt_loc].contiguous(), tr.pe_text, out=hs[:lt_loc])
    if v1 > v0:
        tok = ops.patchify(hidden, pt, p, tr.pe_proj.cin_pad)
        ops.linear(tok[v0:v1].contiguous(), tr.pe_proj, out=hs[lt_loc:])
    cos_l, sin_l = cos[v0:v1].contiguous(), sin[v0:v1].contiguous()
    if v1 == v0:                                               # a text-only shard still hands a valid table to the kernel
        cos_l, sin_l = cos[:1].contiguous(), sin[:1].contiguous()
    nlpad = (nloc + 127) // 128 * 128
    z = lambda *shp: torch.zeros(*shp, dtype=torch.bfloat16, device=dev)   # noqa: E731
    Ql, Kl, Vl = z(heads, nlpad, 64), z(heads, nlpad, 64), z(heads, 64, nlpad)
    Qh, Kh, Vt = z(hloc, npad, 64), z(hloc, npad, 64), z(hloc, 64, npad)
    qscale = (tr.hd ** -0.5) * math.log2(math.e)
    blk_in = [c * hloc * 64 for c in counts]                    # elements I receive from each source rank

    def rows_to_heads(loc, transposed):
        # loc: [heads, nloc, 64] (or [heads, 64, nloc]); destination j gets my rows of ITS heads
        src = (loc[:, :, :nloc] if transposed else loc[:, :nloc]).contiguous()
        got = _a2a(blk_in, src.view(-1), [nloc * hloc * 64] * world, group)
        parts, off = [], 0
        for c in counts:
            blk = got[off:off + c * hloc * 64]
            parts.append(blk.view(hloc, 64, c) if transposed else blk.view(hloc, c, 64))
            off += c * hloc * 64
        return torch.cat(parts, dim=2 if transposed else 1)

    for blk, md in zip(tr.blocks, blocks_mod):
        n1 = ops.layernorm_modulate(hs, blk["ln1"][0], blk["ln1"][1], tr.eps, md["m1"], lt_loc)
        qkv = ops.linear(n1, blk["qkv"])
        ops.qkv_post(qkv, nloc, nlpad, heads, lt_loc, blk["nq"][0], blk["nq"][1], blk["nk"][0], blk["nk"][1], cos_l, sin_l,
                     qscale, 1e-6, Ql, Kl, Vl)
        Qh[:, :N] = rows_to_heads(Ql, False)
        Kh[:, :N] = rows_to_heads(Kl, False)
        Vt[:, :, :N] = rows_to_heads(Vl, True)
        att = torch.empty(N, hloc * 64, dtype=torch.bfloat16, device=dev)
        ops.attention(Qh, Kh, Vt, N, npad, hloc, att)
        # heads -> rows: rank j gets rows [bounds[j], bounds[j+1]) of my heads; I get my rows of every head group
        back = _a2a([nloc * hloc * 64] * world, att.view(-1), blk_in, group)
        att_loc = torch.cat([back[i * nloc * hloc * 64:(i + 1) * nloc * hloc * 64].view(nloc, hloc * 64) for i in range(world)],
                            dim=1).contiguous()
        ops.linear(att_loc, blk["out"], resid=hs, gate=md["gate1"], gate_split=lt_loc, out=hs)
        n2 = ops.layernorm_modulate(hs, blk["ln2"][0], blk["ln2"][1], tr.eps, md["m2"], lt_loc, out=n1)
        f1 = ops.linear(n2, blk["ff1"], act=1)
        ops.linear(f1, blk["ff2"], resid=hs, gate=md["gate2"], gate_split=lt_loc, out=hs)

    width = tr.proj_out.cout_store
    o_loc = torch.zeros(0, width, dtype=torch.bfloat16, device=dev)
    if v1 > v0:
        xv = hs[lt_loc:].contiguous()
        xv = ops.layernorm_modulate(xv, tr.norm_final[0], tr.norm_final[1], tr.eps)
        xv = ops.layernorm_modulate(xv, tr.norm_out[0], tr.norm_out[1], tr.eps, final_mod, 0)
        o_loc = ops.linear(xv, tr.proj_out)
    vcounts = [max(bounds[i + 1], Lt) - max(bounds[i], Lt) for i in range(world)]
    vmax = max(vcounts)
    pad = torch.zeros(vmax, width, dtype=torch.bfloat16, device=dev)
    pad[: o_loc.shape[0]] = o_loc
    bufs = [torch.empty(vmax, width * 2, dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(bufs, pad.view(torch.uint8), group=group)
    o = torch.cat([b[:c] for b, c in zip(bufs, vcounts)], dim=0).view(torch.bfloat16)
    return ops.unpatchify(o.contiguous(), T, Cc, H, W, pt, p, tr.dtype)


class _ShardedVAE:
    def __init__(self, vae, group):
        self._vae, self._group = vae, group
        self.device, self.dtype, self.config = vae.device, vae.dtype, vae.config

    def encode(self, x):
        class _O:
            latent_dist = encode_sharded(self._vae, x, self._group)
        return _O()


class _ShardedTransformer:
    def __init__(self, tr, group):
        self._tr, self._group, self.config = tr, group, tr.config

    def __call__(self, hidden_states, encoder_hidden_states, timestep, image_rotary_emb=None, return_dict=False, **_):
        ts = [int(v) for v in timestep.reshape(-1).tolist()]
        outs = [dit_forward_ulysses(self._tr, hidden_states[b], encoder_hidden_states[b], ts[b if len(ts) > 1 else 0],
                                    image_rotary_emb, self._group) for b in range(hidden_states.shape[0])]
        return (torch.stack(outs),)


class _ShardedPipe:
    """Duck-typed ``pipe`` whose VAE and transformer run sharded over ``group`` (what process_video touches, nothing more)."""

    def __init__(self, pipe, group):
        self._pipe, self._group = pipe, group
        self.vae = _ShardedVAE(pipe.vae, group)
        self.transformer = _ShardedTransformer(pipe.transformer, group)
        self.scheduler = _SharedScheduler(pipe.scheduler, group)
        self.tokenizer, self.text_encoder = pipe.tokenizer, pipe.text_encoder

    def decode_latents(self, latents, _range01=False):
        z = latents.permute(0, 2, 1, 3, 4).contiguous()
        return decode_sharded(self._pipe.vae, z, self._group, _range01=_range01,
                              _prescale=1.0 / float(self._pipe.vae.config["scaling_factor"]))


@torch.no_grad()
def process_video_sharded(pipe, video, *, group=None, **kw):
    """``process_video`` on ONE clip with every stage sharded over the ranks of ``group``: halo-exact VAE (B) and
    sequence/head-parallel DiT (C).  All ranks must pass the same clip; random draws (posterior sample, optional
    pre-noising) are rank 0's, broadcast, so the ranks stay consistent whatever their RNG states.  All ranks return the
    full SR clip, bit-identical to the single-GPU result given rank 0's noise."""
    return process_video(_ShardedPipe(pipe, group), video, **kw)
