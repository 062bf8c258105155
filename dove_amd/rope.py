"""cos/sin tables of diffusers' ``get_3d_rotary_pos_embed(grid_type="slice")`` as requested by
``prepare_rotary_positional_embeddings`` (/root/reference/inference_script.py:364-392).  SURVEY.md App. A.5-6.
Constant per latent shape -> cached; the rotation itself runs in dove_qkv_post_bf16."""
from __future__ import annotations

import functools

import torch


def _axis(dim: int, n: int, theta: float):
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float32)[: dim // 2] / dim))
    ang = torch.outer(torch.arange(n, dtype=torch.float32), freqs)
    return ang.cos().repeat_interleave(2, dim=1), ang.sin().repeat_interleave(2, dim=1)


@functools.lru_cache(maxsize=8)
def _tables(embed_dim: int, gh: int, gw: int, gt: int, theta: float):
    dt, dh = embed_dim // 4, embed_dim // 8 * 3
    axes = [_axis(dt, gt, theta), _axis(dh, gh, theta), _axis(dh, gw, theta)]
    out = []
    for k in (0, 1):
        t = axes[0][k][:, None, None, :].expand(gt, gh, gw, dt)
        h = axes[1][k][None, :, None, :].expand(gt, gh, gw, dh)
        w = axes[2][k][None, None, :, :].expand(gt, gh, gw, dh)
        out.append(torch.cat([t, h, w], dim=-1).reshape(gt * gh * gw, embed_dim).contiguous())
    return tuple(out)


def get_3d_rotary_pos_embed(embed_dim, crops_coords, grid_size, temporal_size, theta: int = 10000, use_real: bool = True,
                            grid_type: str = "linspace", max_size=None, device=None):
    if grid_type != "slice" or not use_real:
        raise NotImplementedError("DOVE's path uses grid_type='slice', use_real=True")
    gh, gw = grid_size
    if max_size is not None and (max_size[0] < gh or max_size[1] < gw):
        raise ValueError("grid exceeds max_size")
    cos, sin = _tables(int(embed_dim), int(gh), int(gw), int(temporal_size), float(theta))
    if device is not None:
        cos, sin = cos.to(device), sin.to(device)
    return cos, sin


def prepare_rotary_positional_embeddings(height, width, num_frames, transformer_config, vae_scale_factor_spatial, device):
    """Same signature and shape rules as the reference's helper (ref :364-392)."""
    p, pt = transformer_config.patch_size, transformer_config.patch_size_t
    gh = height // (vae_scale_factor_spatial * p)
    gw = width // (vae_scale_factor_spatial * p)
    base = num_frames if pt is None else (num_frames + pt - 1) // pt
    return get_3d_rotary_pos_embed(embed_dim=transformer_config.attention_head_dim, crops_coords=None, grid_size=(gh, gw),
                                   temporal_size=base, grid_type="slice", max_size=(gh, gw), device=device)
