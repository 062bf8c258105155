"""AutoencoderKLCogVideoX facade over the HIP operators.

Exposes what /root/reference/inference_script.py touches on ``pipe.vae`` (``.device``, ``.dtype``,
``.config.scaling_factor``, ``.config.block_out_channels``, ``.encode(x).latent_dist.sample()``,
``.decode(z).sample``, ``.enable_slicing()``, ``.enable_tiling()``; :407-409,467,500,644-645) and computes
what diffusers' 3D causal VAE computes: frame-batched (8 px-frames / 2 latent
frames) encoder/decoder with per-conv temporal caches, GroupNorm statistics scoped to one frame-batch.

Internal layout is channels-last bf16 [T,H,W,C]; the graph below only allocates tensors and calls the C-ABI
operators of libdove_hip.so (dove_amd.ops).  B > 1 is looped (the reference always runs B = 1).
"""
from __future__ import annotations

import math
import re

import torch

from . import ops
from .config import AttrDict


def frame_batches(num_frames: int, batch: int):
    """diffusers `_encode`/`_decode` batch rule: 33 @8 -> 9,8,8,8 ; 9 @2 -> 3,2,2,2."""
    n = max(num_frames // batch, 1)
    rem = num_frames % batch
    return [(batch * i + (0 if i == 0 else rem), min(batch * (i + 1) + rem, num_frames)) for i in range(n)]


def spatial_norm_tmap(t_f: int, t_z: int):
    """Frame index of zq used for frame t of f under SpatialNorm3D's nearest resize (odd T>1 splits frame 0)."""
    if t_f > 1 and t_f % 2 == 1:
        if t_z == 1:
            return [0] * t_f
        return [0] + [1 + ((t - 1) * (t_z - 1)) // (t_f - 1) for t in range(1, t_f)]
    return [(t * t_z) // t_f for t in range(t_f)]


class DiagonalGaussianDistribution:
    """Posterior over latents; moments are kept channels-last per batch element, sampled on the GPU."""

    def __init__(self, moments_cl, latent_channels, dtype):
        self._m = moments_cl          # list of [T,h,w,2L] bf16
        self._L = latent_channels
        self._dtype = dtype

    @property
    def parameters(self):
        return torch.stack([ops.ncthw_from_cl(m, 2 * self._L, self._dtype) for m in self._m])

    @property
    def mean(self):
        return self.parameters[:, : self._L]

    def mode(self):
        return self.mean

    def sample(self, generator=None, noise=None):
        """mean + exp(0.5*clamp(logvar,-30,20)) * N(0,1).  ``noise`` ([B,L,T,h,w]) may be injected for
        reproducible parity runs; otherwise it is drawn like diffusers' randn_tensor (global RNG of the device)."""
        T, h, w, _ = self._m[0].shape
        B = len(self._m)
        if noise is None:
            noise = torch.randn(B, self._L, T, h, w, generator=generator, device=self._m[0].device, dtype=self._dtype)
        noise = noise.to(self._m[0].device).contiguous()
        return torch.stack([ops.posterior_sample(m, self._L, noise[b], self._dtype) for b, m in enumerate(self._m)])


class _Out:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class StreamCache(dict):
    """conv_cache for the two-stream mode: besides the halo tensors it carries, per causal conv, the HIP event recorded
    (on the producing batch's stream) at the moment that conv's INPUT was complete -- the consuming batch, running on the
    other stream, waits on it right before its own conv.  That is the only cross-batch dependency of the VAE."""

    def __init__(self):
        super().__init__()
        self.events = {}
        self.stream = None


class AutoencoderKLCogVideoX:
    def __init__(self, config: dict, state_dict: dict, device="cuda", dtype=torch.bfloat16):
        self.config = AttrDict(config)
        self.device = torch.device(device)
        self.dtype = dtype
        self.use_slicing = False
        self.use_tiling = False
        c = self.config
        self.boc = list(c["block_out_channels"])
        self.layers = c.get("layers_per_block", 3)
        self.eps = c.get("norm_eps", 1e-6)
        self.lat = c["latent_channels"]
        self.n_tdown = int(math.log2(c.get("temporal_compression_ratio", 4)))
        self.enc_batch = c.get("num_sample_frames_batch_size", 8)
        self.dec_batch = c.get("num_latent_frames_batch_size", 2)
        if c.get("norm_num_groups", 32) != 32:
            raise NotImplementedError("HIP GroupNorm kernels are built for 32 groups")
        # n_streams = 2 alternates frame-batches on two HIP streams (bit-identical; tests/test_e2e_gpu.py::test_two_stream_vae_is_bit_identical):
        # one batch's HBM-bound GroupNorm / glue kernels run beside the other batch's MFMA-bound convs (the persistent conv3x3_halo4x owns a CU
        # per workgroup, so two convs mostly serialise).  Default since round 6 (+1.4 % per clip; profiles/r06_streams_ab.log).  Per-kernel
        # durations taken with two streams (HIP events, rocprofv3) include co-scheduling waits: bench.py takes its roofline figures from a
        # one-stream pass of the same run and says so.
        self.n_streams = 2
        self._streams = None
        # False: no conv is handed a pack-time weight sum (first-frame temporal sums, sub-pixel upsample sums, frame-pair sums): every launch
        # computes the reference's per-tap arithmetic - for validating a checkpoint without the one extra bf16 rounding of the summed weights
        self.weight_sums = True
        # enable_tiling(): run all tiles of one shape as one batch (False: one tile at a time, the round-3 loop - kept for the A/B)
        self.tile_batching = True
        self.tile_streams = 2
        self.tile_batch_streams = 2       # inside a tile class: frame-batches go round-robin to this many streams (1: one after the other)
        self._tile_helpers = []
        self._tile_stream = None
        self.tile_batch_max = 16
        self._pack(state_dict)

    # ---- weights ---------------------------------------------------------------------------------
    def _pack(self, sd):
        dev = self.device
        self.pc, self.aff = {}, {}
        for k in sd:
            if k.endswith(".conv.weight") and ".conv_y." not in k and ".conv_b." not in k:
                n = k[: -len(".conv.weight")]
                # sub-pixel sums only where an upsample-fused launch reads them; pair sums for the first causal conv behind a TIME-doubling
                # upsampler (decoder up-block i > 0 whose predecessor compresses time: its input frames are bit-identical pairs)
                m = re.fullmatch(r"decoder\.up_blocks\.(\d+)\.resnets\.0\.conv1", n)
                self.pc[n] = ops.pack_conv(sd[k], sd[n + ".conv.bias"], dev, sub=".upsamplers." in n,
                                           pair=bool(m) and 0 < int(m.group(1)) <= self.n_tdown)
            elif k.endswith(".conv_shortcut.weight"):
                n = k[: -len(".weight")]
                self.pc[n] = ops.pack_conv(sd[k], sd[n + ".bias"], dev)
            elif k.endswith(".conv_y.conv.weight"):
                n = k[: -len(".conv_y.conv.weight")]  # spatial norm: one [2C, L] 1x1x1 conv producing Y || B
                w = torch.cat([sd[k], sd[n + ".conv_b.conv.weight"]], dim=0)
                b = torch.cat([sd[n + ".conv_y.conv.bias"], sd[n + ".conv_b.conv.bias"]], dim=0)
                self.pc[n + ".yb"] = ops.pack_conv(w, b, dev)
                self.aff[n] = (sd[n + ".norm_layer.weight"].to(dev, torch.float32).contiguous(),
                               sd[n + ".norm_layer.bias"].to(dev, torch.float32).contiguous())
            elif k.endswith(".weight") and sd[k].dim() == 1 and ".norm_layer." not in k:
                n = k[: -len(".weight")]
                self.aff[n] = (sd[k].to(dev, torch.float32).contiguous(), sd[n + ".bias"].to(dev, torch.float32).contiguous())
        # decoder.conv_out split by spatial tap (include/dove_hip.h dove_conv_out_gather): with 3 output channels a direct 3x3x3
        # conv pads Cout 3 -> 32 and stages every input pixel 27 times; as a (3,1,1) conv with the 9 spatial taps as 27 output
        # channels it stages them 3 times and wastes 5 of 32 MFMA columns, the 9-tap shifted sum rides in the layout kernel
        # encoder.conv_in the other way round (dove_cl_im2col3x3_from_ncthw): 3 input channels pad to 32 with 29 zeros; with the 9 spatial
        # taps unrolled into the input channels (27 real) it is a (3,1,1) conv - a ninth of the MFMA work and of the staged bytes
        wi = sd["encoder.conv_in.conv.weight"]                      # [Cout, Cin, 3, 3, 3]
        self._conv_in_split = wi.shape[3] == 3 and wi.shape[4] == 3 and 9 * wi.shape[1] <= 32
        if self._conv_in_split:
            w27 = wi.float().permute(0, 3, 4, 1, 2).reshape(wi.shape[0], 9 * wi.shape[1], wi.shape[2], 1, 1)   # column (dy*3+dx)*Cin + c
            self.pc["encoder.conv_in.taps"] = ops.pack_conv(w27, sd["encoder.conv_in.conv.bias"], dev)
        w = sd["decoder.conv_out.conv.weight"]                      # [Cout, Cin, 3, 3, 3]
        co, ci, kt, kh, kw = w.shape
        self._conv_out_split = kh == 3 and kw == 3 and 9 * co <= 32
        if self._conv_out_split:
            w27 = w.float().permute(3, 4, 0, 1, 2).reshape(kh * kw * co, ci, kt, 1, 1)    # row (dy*3+dx)*Cout + c
            self.pc["decoder.conv_out.taps"] = ops.pack_conv(w27, None, dev)
            self.conv_out_bias = sd["decoder.conv_out.conv.bias"].to(dev, torch.float32).contiguous()

    # ---- toggles kept for API parity (ref :644-645) ---------------------------------------------
    def enable_slicing(self):
        self.use_slicing = True

    def disable_slicing(self):
        self.use_slicing = False

    def enable_tiling(self, *a, **k):
        self.use_tiling = True

    def disable_tiling(self):
        self.use_tiling = False

    def to(self, *a, **k):
        return self

    # ---- building blocks ---------------------------------------------------------------------------
    def _cconv(self, x, name, cache, **kw):
        """CogVideoXCausalConv3d with conv_cache: the front halo is the last kt-1 input frames of the previous batch."""
        pc = self.pc[name]
        kw["weight_sums"] = self.weight_sums
        if pc.kt == 1:
            return ops.conv(x, pc, nb=self._nb, **kw)
        k = pc.kt - 1
        if self._nb > 1:
            # nb same-shaped tiles in one launch (_tiled): x [nb*T, H, W, C], the cache a [nb, k, H, W, C] view of the previous
            # frame-batch's input - instance b's first frames read cache[b]
            nb = self._nb
            x5 = x.view(nb, x.shape[0] // nb, *x.shape[1:])
            prev = cache.get(name)
            events = getattr(cache, "events", None)    # StreamCache: the class's frame-batches alternate between two HIP streams (_tiled)
            if events is not None:
                ev_prev = events.get(name)
                ev = torch.cuda.Event()
                ev.record(cache.stream)
                events[name] = ev
                if ev_prev is not None:
                    cache.stream.wait_event(ev_prev)
                    prev.record_stream(cache.stream)
            if x5.shape[1] >= k:
                new = x5[:, -k:]
            else:
                pad = prev if prev is not None else x5[:, :1].expand(-1, k, -1, -1, -1)
                new = torch.cat([pad, x5], dim=1)[:, -k:].contiguous()
            cache[name] = new
            return ops.conv(x, pc, cache=prev, nb=nb, **kw)
        fetch = getattr(cache, "fetch", None)      # dove_amd.dist.HaloCache: halo arrives from rank-1 over xGMI
        prev = fetch(name, (k,) + tuple(x.shape[1:]), x.device) if fetch else cache.get(name)
        events = getattr(cache, "events", None)    # StreamCache: frame-batches alternate between two HIP streams
        if events is not None:
            ev_prev = events.get(name)
            ev = torch.cuda.Event()
            ev.record(cache.stream)                # everything that produced x (this conv's input) is ordered before it
            events[name] = ev
            if ev_prev is not None:
                cache.stream.wait_event(ev_prev)
                prev.record_stream(cache.stream)   # keep the other stream's tensor alive for this stream's read
        tdup = kw.get("tdup", 0)
        if tdup == 1 and prev is not None and not fetch and not cache.get(name + "#pair", False):
            # ``tdup = 1`` also declares the CACHE a bit-identical pair.  It is one exactly when the previous batch's input was doubled and at
            # least as long as the halo (its last two frames are then a pair); a shorter batch leaves a slid window (prev[1], x0), which is
            # not.  frame_batches cannot produce that sequence today, but the declaration is checked here, not assumed
            kw["tdup"] = 0
        if x.shape[0] >= k:
            new = x[-k:]      # a view: conv inputs are never written again, the batch tensor simply stays alive
        else:  # fewer frames than the halo: slide the padded window
            pad = prev if prev is not None else x[:1].expand(k, -1, -1, -1)
            new = torch.cat([pad, x], dim=0)[-k:].clone()
        if fetch:
            cache.publish(name, new)
        else:
            cache[name] = new
            cache[name + "#pair"] = bool(tdup) and x.shape[0] >= k      # doubled input (tmode 1: pairs from frame 0; 2: from frame 1, odd length)
        return ops.conv(x, pc, cache=prev, **kw)

    # set by dove_amd.dist while a rank runs a PIECE of a frame-batch (the batch is split over a rank pair):
    #   _gn_hook(x) -> stats over the whole batch (pieces' sums combined);  _piece_role: "head" / "tail" for Upsample3D's
    #   first-frame rule, which diffusers derives from the batch's frame-count parity
    _gn_hook = None
    _piece_role = None
    # > 1 while _tiled runs nb same-shaped tiles as ONE batch: every activation is [nb*T, H, W, C] (tile-major), every operator
    # call carries nb, GroupNorm statistics and conv caches stay per tile
    _nb = 1

    def _norm_silu(self, x, name, zq=None):
        nb = self._nb
        if self._gn_hook is not None:
            stats = self._gn_hook(x)
        else:
            stats = ops.groupnorm_stats_of(x, self.eps, nb)   # fused into the producing conv's epilogue when it could be
        g, b = self.aff[name]
        if zq is None:
            return ops.groupnorm_apply(x, stats, g, b, silu=True, nb=nb)
        yb = ops.conv(zq, self.pc[name + ".yb"])            # [Tz,hz,wz,2C] on the latent grid (pointwise: instances need no care)
        ratio = x.shape[1] // zq.shape[1]
        assert x.shape[1] == zq.shape[1] * ratio and x.shape[2] == zq.shape[2] * ratio and ratio & (ratio - 1) == 0
        return ops.groupnorm_apply(x, stats, g, b, silu=True, yb=yb, sshift=ratio.bit_length() - 1,
                                   tmap=spatial_norm_tmap(x.shape[0] // nb, zq.shape[0] // nb), nb=nb)

    def _resnet(self, x, name, cache, zq=None, tdup=0):
        """``tdup``: x came out of a time-doubling Upsample3D (its frames are bit-identical pairs; 1 / 2 = the upsampler's tmode) - the
        per-pixel norm1 keeps the pairs (same statistics, same zq frame for both halves of a pair), conv1 is told (ops.conv tdup)."""
        h = self._norm_silu(x, name + ".norm1", zq)
        h = self._cconv(h, name + ".conv1", cache, gn_eps=self.eps, tdup=tdup)   # feeds norm2
        h = self._norm_silu(h, name + ".norm2", zq)
        if name + ".conv_shortcut" in self.pc:
            x = ops.conv(x, self.pc[name + ".conv_shortcut"])   # pointwise: the same call for one tile or a batch of tiles
        return self._cconv(h, name + ".conv2", cache, resid=x, gn_eps=self.eps)   # feeds the next block's norm1 / norm_out

    def _downsample(self, x, name, compress_time):
        if compress_time:
            x = ops.avgpool_time(x, self._nb)
        return ops.conv(x, self.pc[name], stride=2, pad=(0, 0), nb=self._nb)

    def _upsample(self, x, name, compress_time):
        T = x.shape[0] // self._nb
        if compress_time and self._piece_role is not None:
            # a piece of a split frame-batch: the piece that starts an odd-length batch keeps its first frame single,
            # every other piece doubles all of its frames (also a single-frame piece, which is not a 1-frame batch)
            tmode, t_out = (2, 2 * T - 1) if self._piece_role == "head" else (1, 2 * T)
        elif compress_time and T > 1:
            tmode, t_out = (2, 2 * T - 1) if T % 2 == 1 else (1, 2 * T)
        else:
            tmode, t_out = 0, T
        return ops.conv(x, self.pc[name], up=1, tmode=tmode, t_out=t_out, pad=(1, 1), gn_eps=self.eps, nb=self._nb,
                        weight_sums=self.weight_sums), tmode

    def _encoder(self, x, cache, split_in=False):
        """``split_in``: x is the im2col'ed input of ``ops.cl_im2col3x3_from_ncthw`` (a spatial tile: through ``ops.tile_gather``, which
        zeroes the taps that reach outside the tile)."""
        h = self._cconv(x, "encoder.conv_in.taps" if split_in and self._conv_in_split else "encoder.conv_in", cache)
        nb = len(self.boc)
        for i in range(nb):
            for j in range(self.layers):
                h = self._resnet(h, f"encoder.down_blocks.{i}.resnets.{j}", cache)
            if i < nb - 1:
                h = self._downsample(h, f"encoder.down_blocks.{i}.downsamplers.0", i < self.n_tdown)
        for j in range(2):
            h = self._resnet(h, f"encoder.mid_block.resnets.{j}", cache)
        h = self._norm_silu(h, "encoder.norm_out")
        return self._cconv(h, "encoder.conv_out", cache)

    def _decoder(self, z, cache, split_out=False):
        """``split_out``: return the tap-split conv_out's fp32 partial planes [T,H,W,32] for ``ops.conv_out_gather`` (``_cl`` inside a
        spatial tile) instead of the conv_out result (the sharded decode's gather wants the real output)."""
        h = self._cconv(z, "decoder.conv_in", cache)
        for j in range(2):
            h = self._resnet(h, f"decoder.mid_block.resnets.{j}", cache, zq=z)
        nb = len(self.boc)
        tdup = 0
        for i in range(nb):
            for j in range(self.layers + 1):
                h = self._resnet(h, f"decoder.up_blocks.{i}.resnets.{j}", cache, zq=z, tdup=tdup if j == 0 else 0)
            if i < nb - 1:
                h, tdup = self._upsample(h, f"decoder.up_blocks.{i}.upsamplers.0", i < self.n_tdown)
        h = self._norm_silu(h, "decoder.norm_out", zq=z)
        if split_out and self._conv_out_split:
            return self._cconv(h, "decoder.conv_out.taps", cache, out_f32=True)
        return self._cconv(h, "decoder.conv_out", cache)

    # ---- diffusers spatial tiling (enable_tiling; SURVEY.md App. A.4, all published numbers use --is_vae_st) --------
    def _tiling_params(self):
        c = self.config
        down = 2 ** (len(self.boc) - 1)
        smin_h, smin_w = c.get("sample_height", 480) // 2, c.get("sample_width", 720) // 2
        return dict(smin_h=smin_h, smin_w=smin_w, lmin_h=int(smin_h / down), lmin_w=int(smin_w / down), of_h=1 / 6, of_w=1 / 5)

    def _tiled(self, x_cl, tile_h, tile_w, stride_h, stride_w, blend_h, blend_w, lim_h, lim_w, batch, fn, im2col_cin=0):
        """Tile loop of diffusers tiled_encode/tiled_decode on a channels-last [T,H,W,C] tensor: each tile runs the whole
        frame-batched network with its own conv caches (GroupNorm statistics become per tile); tiles are cross-faded
        IN PLACE with their already blended upper / left neighbours, cropped and concatenated.  ``im2col_cin`` > 0: x_cl is the im2col'ed
        clip (encoder.conv_in in its (3,1,1) form); ops.tile_gather zeroes the taps that reach outside a tile at its border, so a tile
        still sees zero padding at ITS border."""
        T, H, W, _ = x_cl.shape
        ii, jj = list(range(0, H, stride_h)), list(range(0, W, stride_w))
        rows = [[None] * len(jj) for _ in ii]
        if self.tile_batching:
            # the tiles are independent until the blend: all tiles of one shape (interior / bottom edge / right edge / corner) run as
            # ONE batch - a launch per operator and shape class instead of one per tile (20 tiles -> 4 classes at 720x1280), the
            # largest class first so that the host queues the small classes' launches while the GPU works on the big one.
            # Per tile the arithmetic is that of the loop below, bit for bit (tests/test_e2e_gpu.py).
            classes = {}
            for a, i in enumerate(ii):
                for b, j in enumerate(jj):
                    classes.setdefault((min(tile_h, H - i), min(tile_w, W - j)), []).append((a, b))
            # at most tile_batch_max tiles per batch: activations scale with the batch (30 interior tiles of a 1088x1920 clip would hold
            # 118 GB live; 16 is past the point where the launches fill the chip).  Per tile the result does not depend on the batching
            split = {}
            for (shape, members) in classes.items():
                for k in range(0, len(members), self.tile_batch_max):
                    split[shape + (k,)] = members[k:k + self.tile_batch_max]
            classes = split
            order = sorted(classes.items(), key=lambda kv: -kv[0][0] * kv[0][1] * len(kv[1]))
            # tile_streams = 2: the largest class stays on the caller's stream, the edge classes (a fifth of the work at 720x1280) run
            # on a second HIP stream - their launches fill the CUs the big class leaves idle in its partly filled last rounds and at
            # the deep levels (a 30x45 latent tile is 2 x 2 conv tiles), and vice versa
            side = None
            main = torch.cuda.current_stream(self.device) if x_cl.is_cuda else None
            if self.tile_streams > 1 and len(order) > 1 and x_cl.is_cuda:
                if self._tile_stream is None:
                    self._tile_stream = torch.cuda.Stream(device=self.device)
                side = self._tile_stream
                side.wait_stream(main)                          # x_cl was produced on the caller's stream
            try:
                for k, ((th, tw, _), members) in enumerate(order):
                    nb = len(members)
                    base = side if (side is not None and k > 0) else main
                    fbs = frame_batches(T, batch)
                    # tile_batch_streams = 2: the class's frame-batches alternate between its own stream and a helper, ordered only by the
                    # per-conv events of StreamCache (what _run_batches does for the untiled clip): the small launches of a 240x360 tile leave
                    # more of the chip idle at their ends than a 720x1280 frame's, and the other batch's kernels fill it
                    helpers = []
                    if self.tile_batch_streams > 1 and len(fbs) > 1 and base is not None:
                        slot = 0 if base is main else 1
                        nh = min(self.tile_batch_streams, len(fbs)) - 1
                        while len(self._tile_helpers) <= slot:
                            self._tile_helpers.append([])
                        while len(self._tile_helpers[slot]) < nh:
                            self._tile_helpers[slot].append(torch.cuda.Stream(device=self.device))
                        helpers = self._tile_helpers[slot][:nh]
                        for hs in helpers:
                            hs.wait_stream(base)
                            x_cl.record_stream(hs)
                    lanes = [base] + helpers                    # frame-batch bi runs on lanes[bi % len(lanes)]
                    cache, parts = (StreamCache() if helpers else {}), []
                    self._nb = nb
                    with torch.cuda.stream(base):
                        try:
                            for bi, (s, e) in enumerate(fbs):
                                st = lanes[bi % len(lanes)]
                                if helpers:
                                    cache.stream = st
                                with torch.cuda.stream(st):
                                    xb = ops.tile_gather(x_cl, s, e - s, th, tw, [(ii[a], jj[b]) for a, b in members], im2col_cin)   # [nb * t, th, tw, C]
                                    o = fn(xb, cache)
                                    parts.append(o.view(nb, o.shape[0] // nb, *o.shape[1:]))
                        finally:
                            self._nb = 1
                            for hs in helpers:
                                base.wait_stream(hs)
                        if helpers:
                            for o_ in parts:
                                o_.record_stream(base)
                        out = torch.cat(parts, dim=1) if len(parts) > 1 else parts[0].contiguous()                # [nb, T', oh, ow, C]
                    if side is not None and k > 0:
                        out.record_stream(main)                     # blended / cropped on the caller's stream below
                    for n, (a, b) in enumerate(members):
                        rows[a][b] = out[n]
            finally:
                # also when a class raised: kernels of the side stream may still be reading x_cl, which the caller is free to drop
                if side is not None:
                    main.wait_stream(side)
        else:
            for a, i in enumerate(ii):
                for b, j in enumerate(jj):
                    cache, parts = {}, []
                    for s, e in frame_batches(T, batch):
                        parts.append(fn(ops.tile_gather(x_cl, s, e - s, min(tile_h, H - i), min(tile_w, W - j), [(i, j)], im2col_cin), cache))
                    rows[a][b] = torch.cat(parts, dim=0) if len(parts) > 1 else parts[0]
        out_rows = []
        for i, row in enumerate(rows):
            out_row = []
            for j, tile in enumerate(row):
                if i > 0:
                    ops.blend_edge(rows[i - 1][j], tile, min(rows[i - 1][j].shape[1], tile.shape[1], blend_h), 0)
                if j > 0:
                    ops.blend_edge(row[j - 1], tile, min(row[j - 1].shape[2], tile.shape[2], blend_w), 1)
                out_row.append(tile[:, :lim_h, :lim_w])
            out_rows.append(torch.cat(out_row, dim=2))
        return torch.cat(out_rows, dim=1).contiguous()

    def _tiled_encode(self, x_cl, im2col_cin=0):
        p = self._tiling_params()
        st_h, st_w = int(p["smin_h"] * (1 - p["of_h"])), int(p["smin_w"] * (1 - p["of_w"]))
        bl_h, bl_w = int(p["lmin_h"] * p["of_h"]), int(p["lmin_w"] * p["of_w"])
        fn = (lambda xb, cache: self._encoder(xb, cache, split_in=True)) if im2col_cin else self._encoder
        return self._tiled(x_cl, p["smin_h"], p["smin_w"], st_h, st_w, bl_h, bl_w, p["lmin_h"] - bl_h, p["lmin_w"] - bl_w,
                           self.enc_batch, fn, im2col_cin)

    def _tiled_decode(self, z_cl):
        p = self._tiling_params()
        st_h, st_w = int(p["lmin_h"] * (1 - p["of_h"])), int(p["lmin_w"] * (1 - p["of_w"]))
        bl_h, bl_w = int(p["smin_h"] * p["of_h"]), int(p["smin_w"] * p["of_w"])
        fn = self._decoder
        if self._conv_out_split:
            # the tap-split conv_out inside a tile: the 9-tap shifted sum is per frame, i.e. per tile (taps beyond the tile border add
            # nothing = the tile's own zero padding), and stays channels-last for the blend (8 channels: 3 real)
            cout, bias = self.config["out_channels"], self.conv_out_bias
            fn = lambda zb, cache: ops.conv_out_gather_cl(self._decoder(zb, cache, split_out=True), cout, bias, 8)   # noqa: E731
        return self._tiled(z_cl, p["lmin_h"], p["lmin_w"], st_h, st_w, bl_h, bl_w, p["smin_h"] - bl_h, p["smin_w"] - bl_w,
                           self.dec_batch, fn)

    # ---- two-stream execution of the frame-batches ------------------------------------------------------------
    def _run_batches(self, x_cl, batch, fn, post=None):
        """Run ``fn`` (encoder or decoder) over the frame-batches.  With ``n_streams >= 2`` the batches go round-robin to that many HIP
        streams ordered only by per-conv events, so one batch's HBM-bound GroupNorm kernels overlap another batch's
        MFMA-bound convolutions; results are bit-identical to the single-stream order."""
        batches = frame_batches(x_cl.shape[0], batch)
        if self.n_streams < 2 or len(batches) < 2 or not x_cl.is_cuda:
            cache, outs = {}, []
            for s, e in batches:
                o = fn(x_cl[s:e], cache)
                outs.append(post(o) if post else o)
            return outs
        if self._streams is None or len(self._streams) != self.n_streams:
            self._streams = [torch.cuda.Stream(device=self.device) for _ in range(self.n_streams)]
        main = torch.cuda.current_stream(self.device)
        cache, outs = StreamCache(), []
        for st in self._streams:
            st.wait_stream(main)                   # x_cl was produced on the caller's stream
        for i, (s, e) in enumerate(batches):
            st = self._streams[i % len(self._streams)]
            cache.stream = st
            with torch.cuda.stream(st):
                xb = x_cl[s:e]
                xb.record_stream(st)
                o = fn(xb, cache)
                o = post(o) if post else o
                outs.append(o)
        for st in self._streams:
            main.wait_stream(st)
        for o in outs:
            o.record_stream(main)
        return outs

    # ---- public API -----------------------------------------------------------------------------
    @torch.no_grad()
    def encode(self, x: torch.Tensor, return_dict: bool = True):
        """x [B,3,F,H,W] in [-1,1] -> .latent_dist (posterior over [B,L,1+(F-1)/4,H/8,W/8])."""
        if x.dim() != 5:
            raise ValueError("expected [B,C,F,H,W]")
        tp = self._tiling_params()
        tiled = self.use_tiling and (x.shape[-1] > tp["smin_w"] or x.shape[-2] > tp["smin_h"])
        x = x.to(self.device).contiguous()
        cin_pad = self.pc["encoder.conv_in"].cin_pad
        moments = []
        for b in range(x.shape[0]):
            if tiled:
                if self._conv_in_split:
                    moments.append(self._tiled_encode(ops.cl_im2col3x3_from_ncthw(x[b], self.pc["encoder.conv_in.taps"].cin_pad),
                                                      im2col_cin=x.shape[1]))
                else:
                    moments.append(self._tiled_encode(ops.cl_from_ncthw(x[b], cin_pad)))
                continue
            if self._conv_in_split:
                x_cl = ops.cl_im2col3x3_from_ncthw(x[b], self.pc["encoder.conv_in.taps"].cin_pad)
                outs = self._run_batches(x_cl, self.enc_batch, lambda xb, cache: self._encoder(xb, cache, split_in=True))
            else:
                outs = self._run_batches(ops.cl_from_ncthw(x[b], cin_pad), self.enc_batch, self._encoder)
            moments.append(torch.cat(outs, dim=0) if len(outs) > 1 else outs[0])
        dist = DiagonalGaussianDistribution(moments, self.lat, self.dtype)
        return _Out(latent_dist=dist) if return_dict else (dist,)

    @torch.no_grad()
    def decode(self, z: torch.Tensor, return_dict: bool = True, _range01: bool = False, _prescale: float = 1.0):
        """z [B,L,T,h,w] (already divided by scaling_factor) -> .sample [B,3,1+4(T-1),8h,8w]."""
        z = z.to(self.device).contiguous()
        cin_pad = self.pc["decoder.conv_in"].cin_pad
        cout = self.config["out_channels"]
        post = dict(scale=0.5, shift=0.5, lo=0.0, hi=1.0) if _range01 else {}
        tp = self._tiling_params()
        tiled = self.use_tiling and (z.shape[-1] > tp["lmin_w"] or z.shape[-2] > tp["lmin_h"])
        vids = []
        for b in range(z.shape[0]):
            z_cl = ops.cl_from_ncthw(z[b], cin_pad, scale=_prescale)
            if tiled:
                vids.append(ops.ncthw_from_cl(self._tiled_decode(z_cl), cout, self.dtype, **post))
                continue
            if self._conv_out_split:
                outs = self._run_batches(z_cl, self.dec_batch, lambda zb, cache: self._decoder(zb, cache, split_out=True),
                                         post=lambda o: ops.conv_out_gather(o, cout, self.conv_out_bias, self.dtype, **post))
            else:
                outs = self._run_batches(z_cl, self.dec_batch, self._decoder, post=lambda o: ops.ncthw_from_cl(o, cout, self.dtype, **post))
            vids.append(torch.cat(outs, dim=1) if len(outs) > 1 else outs[0])
        sample = torch.stack(vids)
        return _Out(sample=sample) if return_dict else (sample,)
