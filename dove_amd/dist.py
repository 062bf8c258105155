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
    run as a wavefront skewed by one layer.  GroupNorm statistics are per frame-batch, batches are never split
    across ranks, hence no reduction is needed and results are bit-identical to the single-GPU path.
    The DiT attends over all tokens of the clip (not frame-separable): moments are all-gathered and the DiT runs
    replicated; only the VAE (71 % of the FLOPs) is sharded.
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


def _run_sharded(vae, x_cl, batches, world, rank, group, fn):
    """Run ``fn(batch_tensor, cache)`` over this rank's contiguous group of frame-batches."""
    groups = split_batches(batches, world)
    mine = groups[rank]
    active = sum(1 for g_ in groups if g_)        # ranks beyond the number of frame-batches own nothing and exchange nothing
    cache = HaloCache(group, rank, active)
    outs = []
    for i, (s, e) in enumerate(mine):
        first, last = i == 0, i == len(mine) - 1
        cache.phase = "both" if first and last else ("first" if first else ("last" if last else "mid"))
        outs.append(fn(x_cl[s:e], cache))
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
    return DiagonalGaussianDistribution([moments], vae.lat, vae.dtype)


@torch.no_grad()
def decode_sharded(vae, z, group=None, _range01=False, _prescale=1.0):
    """vae.decode with latent frame-batches sharded over ranks; every rank returns the full [1,3,F,H,W] video."""
    from . import ops
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    assert z.shape[0] == 1
    z = z.to(vae.device).contiguous()
    z_cl = ops.cl_from_ncthw(z[0], vae.pc["decoder.conv_in"].cin_pad, scale=_prescale)
    outs, cache = _run_sharded(vae, z_cl, frame_batches(z_cl.shape[0], vae.dec_batch), world, rank, group, vae._decoder)
    full = _gather_time(outs, group, world, vae.device)          # [F,H,W,4] channels-last
    post = dict(scale=0.5, shift=0.5, lo=0.0, hi=1.0) if _range01 else {}
    vae.last_halo_bytes = cache.bytes_sent
    return ops.ncthw_from_cl(full, vae.config["out_channels"], vae.dtype, **post)[None]
