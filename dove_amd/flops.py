"""Algorithmic work of the one-step SR operator: MACs of every dense contraction
(convs, linears, QK^T and PV), excluding norms / activations / elementwise.  Used by bench.py for the roofline
figures and by tests/test_oracle_structure.py to reproduce the paper's 504.81 T MACs
(/root/reference/assets/Quantitative-2.png; 504.60 T here without the attention matmuls)."""
from __future__ import annotations

from .vae import frame_batches


def _tdown(T):
    return 1 + (T - 1) // 2 if T % 2 else T // 2


def _tup(T):
    return T if T == 1 else (2 * T - 1 if T % 2 else 2 * T)


def vae_encode_macs(cfg, F, H, W):
    boc = cfg["block_out_channels"]
    L = cfg.get("layers_per_block", 3)
    lat = cfg["latent_channels"]
    n_td = {1: 0, 2: 1, 4: 2, 8: 3}[cfg.get("temporal_compression_ratio", 4)]
    total = 0
    for s, e in frame_batches(F, cfg.get("num_sample_frames_batch_size", 8)):
        T, h, w = e - s, H, W
        total += 27 * cfg["in_channels"] * boc[0] * T * h * w
        ch = boc[0]
        for i, co in enumerate(boc):
            P = T * h * w
            for j in range(L):
                total += 27 * ch * co * P + 27 * co * co * P + (ch * co * P if ch != co else 0)
                ch = co
            if i < len(boc) - 1:
                if i < n_td:
                    T = _tdown(T)
                h, w = (h - 2) // 2 + 1, (w - 2) // 2 + 1
                total += 9 * ch * ch * T * h * w
        P = T * h * w
        total += 4 * 27 * ch * ch * P + 27 * ch * 2 * lat * P
    return total


def vae_decode_macs(cfg, T_lat, h, w):
    boc = cfg["block_out_channels"][::-1]
    L = cfg.get("layers_per_block", 3) + 1
    lat = cfg["latent_channels"]
    n_td = {1: 0, 2: 1, 4: 2, 8: 3}[cfg.get("temporal_compression_ratio", 4)]
    total = 0
    for s, e in frame_batches(T_lat, cfg.get("num_latent_frames_batch_size", 2)):
        T, hh, ww = e - s, h, w
        Pz = T * hh * ww                      # SpatialNorm 1x1 convs run on the upsampled zq in the reference:
        P = Pz                                # count them at f's resolution like App. D
        ch = boc[0]
        total += 27 * lat * ch * P
        total += 4 * 27 * ch * ch * P + 4 * 2 * lat * ch * P
        for i, co in enumerate(boc):
            for j in range(L):
                total += 27 * ch * co * P + 27 * co * co * P + (ch * co * P if ch != co else 0)
                total += 2 * lat * ch * P + 2 * lat * co * P
                ch = co
            if i < len(boc) - 1:
                if i < n_td:
                    T = _tup(T)
                hh, ww = 2 * hh, 2 * ww
                P = T * hh * ww
                total += 9 * ch * ch * P
        total += 2 * lat * ch * P + 27 * ch * cfg["out_channels"] * P
    return total


def dit_macs(cfg, T_lat_padded, h, w, text_len=226, attention=True):
    D = cfg["num_attention_heads"] * cfg["attention_head_dim"]
    p, pt = cfg["patch_size"], cfg["patch_size_t"] or 1
    nv = (T_lat_padded // pt) * (h // p) * (w // p)
    N = nv + text_len
    lin = cfg["num_layers"] * N * (4 * D * D + 8 * D * D)
    att = cfg["num_layers"] * 2 * N * N * D if attention else 0
    te = cfg["time_embed_dim"]
    small = (cfg["in_channels"] * p * p * pt * D * nv + cfg["text_embed_dim"] * D * text_len + D * cfg["out_channels"] * p * p * pt * nv
             + cfg["num_layers"] * 2 * te * 6 * D + te * 2 * D + D * te + te * te)
    return dict(linears=lin, attention=att, small=small, tokens=N)


def clip_macs(vae_cfg, dit_cfg, F, H, W, text_len=226):
    """Total MACs of process_video on a [1,3,F,H,W] clip (F = 8N+1, H and W multiples of 16)."""
    down = 2 ** (len(vae_cfg["block_out_channels"]) - 1)
    tcr = vae_cfg.get("temporal_compression_ratio", 4)
    h, w = H // down, W // down
    T = 1 + (F - 1) // tcr
    pt = dit_cfg["patch_size_t"] or 1
    Tp = T + (T % pt)
    d = dit_macs(dit_cfg, Tp, h, w, text_len)
    enc, dec = vae_encode_macs(vae_cfg, F, H, W), vae_decode_macs(vae_cfg, T, h, w)
    total = enc + dec + d["linears"] + d["attention"] + d["small"]
    return dict(encode=enc, decode=dec, dit_linears=d["linears"], attention=d["attention"], dit_small=d["small"], total=total,
                flop=2 * total, tokens=d["tokens"])
