"""Oracle: CogVideoX 3D causal VAE (encode / decode), torch-CPU.  TEST INFRASTRUCTURE ONLY.

Restates what ``pipe.vae.encode(video).latent_dist`` (/root/reference/inference_script.py:408)
and ``pipe.decode_latents`` (/root/reference/inference_script.py:500) compute inside the
un-vendored ``diffusers`` AutoencoderKLCogVideoX (parity unpinned, see
oracle/__init__.py).  Tensors are [B, C, T, H, W] like the reference.  Weights are a flat dict
keyed by the diffusers state-dict names.

``dtype=torch.bfloat16`` emulates the reference's bf16 run (every module output rounded to bf16,
torch upcasts inside group_norm); ``torch.float32`` is the ground truth the PSNR gates use.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def frame_batches(num_frames: int, batch: int):
    """diffusers' `_encode` / `_decode` frame-batch rule:
    33 px-frames @8 -> [0,9),[9,17),[17,25),[25,33);  9 latent frames @2 -> [0,3),[3,5),[5,7),[7,9)."""
    n = max(num_frames // batch, 1)
    rem = num_frames % batch
    out = []
    for i in range(n):
        start = batch * i + (0 if i == 0 else rem)
        end = batch * (i + 1) + rem
        out.append((start, min(end, num_frames)))
    return out


class OracleVAE:
    def __init__(self, cfg: dict, weights: dict, dtype=torch.float32):
        self.cfg = cfg
        self.dtype = dtype
        from .dit import WeightView
        self.w = WeightView(weights, dtype)
        self.groups = cfg.get("norm_num_groups", 32)
        self.eps = cfg.get("norm_eps", 1e-6)
        self.boc = list(cfg["block_out_channels"])
        self.layers = cfg.get("layers_per_block", 3)
        tcr = cfg.get("temporal_compression_ratio", 4)
        self.n_tdown = {1: 0, 2: 1, 4: 2, 8: 3}[tcr]
        self.enc_batch = cfg.get("num_sample_frames_batch_size", 8)
        self.dec_batch = cfg.get("num_latent_frames_batch_size", 2)

    # ---- primitives -----------------------------------------------------------------
    def causal_conv(self, x, name, cache: dict):
        """CogVideoXCausalConv3d: front-only temporal pad of kt-1 frames taken
        from the previous frame-batch's input (conv_cache) or the first frame replicated."""
        w = self.w[name + ".conv.weight"]
        b = self.w[name + ".conv.bias"]
        kt = w.shape[2]
        if kt > 1:
            prev = cache.get(name)
            if prev is None:
                prev = x[:, :, :1].repeat(1, 1, kt - 1, 1, 1)
            x = torch.cat([prev, x], dim=2)
            cache[name] = x[:, :, -(kt - 1):].clone()
        p = (w.shape[3] - 1) // 2
        return F.conv3d(x, w, b, padding=(0, p, p))

    def group_norm(self, x, name):
        return F.group_norm(x, self.groups, self.w[name + ".weight"], self.w[name + ".bias"], self.eps)

    def spatial_norm(self, f, zq, name):
        """CogVideoXSpatialNorm3D: GN(f) * conv_y(zq^) + conv_b(zq^), zq nearest-
        resized to f's (T,H,W); for odd T>1 the first frame is resized separately."""
        if f.shape[2] > 1 and f.shape[2] % 2 == 1:
            z_first = F.interpolate(zq[:, :, :1], size=(1,) + tuple(f.shape[-2:]))
            z_rest = F.interpolate(zq[:, :, 1:], size=(f.shape[2] - 1,) + tuple(f.shape[-2:]))
            zq = torch.cat([z_first, z_rest], dim=2)
        else:
            zq = F.interpolate(zq, size=tuple(f.shape[-3:]))
        y = F.conv3d(zq, self.w[name + ".conv_y.conv.weight"], self.w[name + ".conv_y.conv.bias"])
        b = self.w[name + ".conv_b.conv.weight"], self.w[name + ".conv_b.conv.bias"]
        bb = F.conv3d(zq, b[0], b[1])
        nf = F.group_norm(f, self.groups, self.w[name + ".norm_layer.weight"],
                          self.w[name + ".norm_layer.bias"], self.eps)
        return nf * y + bb

    def resnet(self, x, name, cache, zq=None):
        cin = x.shape[1]
        h = self.spatial_norm(x, zq, name + ".norm1") if zq is not None else self.group_norm(x, name + ".norm1")
        h = F.silu(h)
        h = self.causal_conv(h, name + ".conv1", cache)
        h = self.spatial_norm(h, zq, name + ".norm2") if zq is not None else self.group_norm(h, name + ".norm2")
        h = F.silu(h)
        h = self.causal_conv(h, name + ".conv2", cache)
        cout = h.shape[1]
        if cin != cout:
            x = F.conv3d(x, self.w[name + ".conv_shortcut.weight"], self.w[name + ".conv_shortcut.bias"])
        return x + h

    def downsample(self, x, name, compress_time):
        B, C, T, H, W = x.shape
        if compress_time:
            xx = x.permute(0, 3, 4, 1, 2).reshape(B * H * W, C, T)
            if T % 2 == 1:
                first, rest = xx[..., :1], xx[..., 1:]
                if rest.shape[-1] > 0:
                    rest = F.avg_pool1d(rest, 2, 2)
                xx = torch.cat([first, rest], dim=-1)
            else:
                xx = F.avg_pool1d(xx, 2, 2)
            T = xx.shape[-1]
            x = xx.reshape(B, H, W, C, T).permute(0, 3, 4, 1, 2)
        x = F.pad(x, (0, 1, 0, 1))
        xf = x.permute(0, 2, 1, 3, 4).reshape(B * T, C, H + 1, W + 1)
        xf = F.conv2d(xf, self.w[name + ".conv.weight"], self.w[name + ".conv.bias"], stride=2)
        return xf.reshape(B, T, C, xf.shape[-2], xf.shape[-1]).permute(0, 2, 1, 3, 4)

    def upsample(self, x, name, compress_time):
        B, C, T, H, W = x.shape
        if compress_time:
            if T > 1 and T % 2 == 1:
                first = F.interpolate(x[:, :, 0], scale_factor=2.0)
                rest = F.interpolate(x[:, :, 1:], scale_factor=2.0)
                x = torch.cat([first[:, :, None], rest], dim=2)
            elif T > 1:
                x = F.interpolate(x, scale_factor=2.0)
            else:
                x = F.interpolate(x[:, :, 0], scale_factor=2.0)[:, :, None]
        else:
            xf = x.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, W)
            xf = F.interpolate(xf, scale_factor=2.0)
            x = xf.reshape(B, T, C, 2 * H, 2 * W).permute(0, 2, 1, 3, 4)
        B, C, T, H, W = x.shape
        xf = x.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, W)
        xf = F.conv2d(xf, self.w[name + ".conv.weight"], self.w[name + ".conv.bias"], padding=1)
        return xf.reshape(B, T, C, H, W).permute(0, 2, 1, 3, 4)

    # ---- encoder / decoder ----------------------------------------------------------
    def encoder(self, x, cache):
        h = self.causal_conv(x, "encoder.conv_in", cache)
        nb = len(self.boc)
        for i in range(nb):
            for j in range(self.layers):
                h = self.resnet(h, f"encoder.down_blocks.{i}.resnets.{j}", cache)
            if i < nb - 1:
                h = self.downsample(h, f"encoder.down_blocks.{i}.downsamplers.0", i < self.n_tdown)
        for j in range(2):
            h = self.resnet(h, f"encoder.mid_block.resnets.{j}", cache)
        h = F.silu(self.group_norm(h, "encoder.norm_out"))
        return self.causal_conv(h, "encoder.conv_out", cache)

    def decoder(self, z, cache):
        h = self.causal_conv(z, "decoder.conv_in", cache)
        for j in range(2):
            h = self.resnet(h, f"decoder.mid_block.resnets.{j}", cache, zq=z)
        nb = len(self.boc)
        for i in range(nb):
            for j in range(self.layers + 1):
                h = self.resnet(h, f"decoder.up_blocks.{i}.resnets.{j}", cache, zq=z)
            if i < nb - 1:
                h = self.upsample(h, f"decoder.up_blocks.{i}.upsamplers.0", i < self.n_tdown)
        h = F.silu(self.spatial_norm(h, z, "decoder.norm_out"))
        return self.causal_conv(h, "decoder.conv_out", cache)

    # ---- diffusers spatial tiling (enable_tiling; reference runs it via --is_vae_st, ref :643-645) ----
    def tiling_params(self):
        sh, sw = self.cfg.get("sample_height", 480), self.cfg.get("sample_width", 720)
        down = 2 ** (len(self.boc) - 1)
        p = dict(smin_h=sh // 2, smin_w=sw // 2, of_h=1 / 6, of_w=1 / 5)
        p["lmin_h"], p["lmin_w"] = int(p["smin_h"] / down), int(p["smin_w"] / down)
        return p

    @staticmethod
    def blend_v(a, b, extent):
        extent = min(a.shape[3], b.shape[3], extent)
        for y in range(extent):
            b[:, :, :, y, :] = a[:, :, :, -extent + y, :] * (1 - y / extent) + b[:, :, :, y, :] * (y / extent)
        return b

    @staticmethod
    def blend_h(a, b, extent):
        extent = min(a.shape[4], b.shape[4], extent)
        for x in range(extent):
            b[:, :, :, :, x] = a[:, :, :, :, -extent + x] * (1 - x / extent) + b[:, :, :, :, x] * (x / extent)
        return b

    def _tiled(self, x, tile_h, tile_w, stride_h, stride_w, blend_h, blend_w, lim_h, lim_w, batch, fn):
        """Shared tile loop of diffusers tiled_encode / tiled_decode: every tile runs the full frame-batched network with
        its own conv caches; tiles are blended IN PLACE with their already-blended upper / left neighbours, cropped
        and concatenated."""
        rows = []
        for i in range(0, x.shape[3], stride_h):
            row = []
            for j in range(0, x.shape[4], stride_w):
                cache, parts = {}, []
                for s, e in frame_batches(x.shape[2], batch):
                    parts.append(fn(x[:, :, s:e, i:i + tile_h, j:j + tile_w], cache))
                row.append(torch.cat(parts, dim=2))
            rows.append(row)
        out_rows = []
        for i, row in enumerate(rows):
            out_row = []
            for j, tile in enumerate(row):
                if i > 0:
                    tile = self.blend_v(rows[i - 1][j], tile, blend_h)
                if j > 0:
                    tile = self.blend_h(row[j - 1], tile, blend_w)
                out_row.append(tile[:, :, :, :lim_h, :lim_w])
            out_rows.append(torch.cat(out_row, dim=4))
        return torch.cat(out_rows, dim=3)

    @torch.no_grad()
    def tiled_encode(self, x):
        p = self.tiling_params()
        st_h, st_w = int(p["smin_h"] * (1 - p["of_h"])), int(p["smin_w"] * (1 - p["of_w"]))
        bl_h, bl_w = int(p["lmin_h"] * p["of_h"]), int(p["lmin_w"] * p["of_w"])
        return self._tiled(x.to(self.dtype), p["smin_h"], p["smin_w"], st_h, st_w, bl_h, bl_w, p["lmin_h"] - bl_h,
                           p["lmin_w"] - bl_w, self.enc_batch, self.encoder)

    @torch.no_grad()
    def tiled_decode(self, z):
        p = self.tiling_params()
        st_h, st_w = int(p["lmin_h"] * (1 - p["of_h"])), int(p["lmin_w"] * (1 - p["of_w"]))
        bl_h, bl_w = int(p["smin_h"] * p["of_h"]), int(p["smin_w"] * p["of_w"])
        return self._tiled(z.to(self.dtype), p["lmin_h"], p["lmin_w"], st_h, st_w, bl_h, bl_w, p["smin_h"] - bl_h,
                           p["smin_w"] - bl_w, self.dec_batch, self.decoder)

    @torch.no_grad()
    def encode(self, x, tiling=False):
        """[B,3,F,H,W] -> posterior parameters [B, 2*latent, T, H/8, W/8] (mean || logvar)."""
        p = self.tiling_params()
        if tiling and (x.shape[-1] > p["smin_w"] or x.shape[-2] > p["smin_h"]):
            return self.tiled_encode(x)
        x = x.to(self.dtype)
        cache, outs = {}, []
        for s, e in frame_batches(x.shape[2], self.enc_batch):
            outs.append(self.encoder(x[:, :, s:e], cache))
        return torch.cat(outs, dim=2)

    @staticmethod
    def sample(params, noise):
        """DiagonalGaussianDistribution.sample with the noise injected by the caller."""
        mean, logvar = params.chunk(2, dim=1)
        logvar = logvar.clamp(-30.0, 20.0)
        return mean + torch.exp(0.5 * logvar) * noise.to(mean.dtype)

    @torch.no_grad()
    def decode(self, z, tiling=False):
        """[B,latent,T,h,w] (already divided by scaling_factor) -> [B,3,F,H,W]."""
        p = self.tiling_params()
        if tiling and (z.shape[-1] > p["lmin_w"] or z.shape[-2] > p["lmin_h"]):
            return self.tiled_decode(z)
        z = z.to(self.dtype)
        cache, outs = {}, []
        for s, e in frame_batches(z.shape[2], self.dec_batch):
            outs.append(self.decoder(z[:, :, s:e], cache))
        return torch.cat(outs, dim=2)
