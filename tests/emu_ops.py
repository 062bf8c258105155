"""TEST INFRASTRUCTURE: torch restatement of every operator in include/dove_hip.h, with the same Python
signatures as dove_amd.ops.  Two uses:
  * `-m "not gpu"`: monkeypatched over dove_amd.ops so the host graphs (VAE frame-batching + conv caches,
    SpatialNorm frame maps, DiT row classes, modulation regrouping, layouts) can be checked against the oracle
    on CPU -- the HIP library is never involved there and no product code imports this file;
  * `-m gpu`: the per-operator expected value for the HIP kernels on identical inputs.
Semantics follow the kernels: fp32 arithmetic, bf16 rounding at operator outputs."""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from dataclasses import dataclass

BF = torch.bfloat16


@dataclass
class EmuConv:
    """The emulation's OWN weight container: the bf16-rounded weight in its natural Conv3d layout [cout, cin, kt, kh, kw]
    (no tap-major repacking, no channel padding) - deliberately independent of dove_amd.ops.pack_conv, so a packing bug
    in the product (tap order, padded rows, transposed operands) shows up in every operator case of tests/test_ops_gpu.py.
    Only the channel-padding RULE of the interface (how many channels the activation tensors carry) is restated."""
    wn: torch.Tensor
    bias: torch.Tensor | None
    kt: int
    kh: int
    kw: int
    cin: int
    cin_pad: int
    cout: int
    cout_pad: int

    @property
    def cout_store(self) -> int:
        return (self.cout + 3) // 4 * 4


def pack_conv(weight, bias, device="cpu"):
    w = weight.detach().float()
    if w.dim() == 2:
        w = w[:, :, None, None, None]
    elif w.dim() == 4:
        w = w[:, :, None]
    cout, cin, kt, kh, kw = w.shape
    cin_pad = cin if (cin > 32 and cin % 64 == 0) else (cin + 31) // 32 * 32      # include/dove_hip.h: activation channel rule
    cout_pad = (cout + 31) // 32 * 32
    return EmuConv(w.to(BF).float(), None if bias is None else bias.detach().float(), kt, kh, kw, cin, cin_pad, cout, cout_pad)


def _natural(pc):
    """(weight [cout, cin, kt, kh, kw] fp32, bias [cout] fp32 | None) of either container.  A product PackedConv (CPU
    host-graph tests run the product's packer) is read back through its documented layout [tap][cout_pad][cin_pad]."""
    if isinstance(pc, EmuConv):
        return pc.wn, pc.bias
    w = pc.w.float().view(pc.kt, pc.kh, pc.kw, pc.cout_pad, pc.cin_pad).permute(3, 4, 0, 1, 2)[: pc.cout, : pc.cin]
    return w, (None if pc.bias is None else pc.bias.float()[: pc.cout])


def _frame_index(t, tmode):
    return t if tmode == 0 else (t >> 1 if tmode == 1 else (0 if t == 0 else 1 + ((t - 1) >> 1)))


def conv(x, pc, *, cache=None, stride=1, pad=(None, None), up=0, tmode=0, t_out=None, hw_out=None, resid=None,
         gate=None, gate_split=0, act=0, ldo=None, out=None, gn_eps=None, out_f32=False, nb=1, tdup=0, weight_sums=True):
    if tdup:
        # dove_conv_desc.tdup is a DECLARATION by the caller (dove_amd/vae.py: the conv behind a time-doubling Upsample3D): the emulation
        # computes the plain per-tap conv - the same function of such frames - and checks that the declaration is true, bit for bit
        assert pc.kt == 3 and not (tdup == 2 and cache is not None)
        xs = x.view(nb, x.shape[0] // nb, *x.shape[1:])
        Ti = xs.shape[1]
        pairs = [(i, i + 1) for i in range(0, Ti - 1, 2)] if tdup == 1 else [(i, i + 1) for i in range(1, Ti - 1, 2)]
        assert (Ti % 2 == 0) if tdup == 1 else (Ti % 2 == 1), (tdup, Ti)
        for a, b in pairs:
            assert torch.equal(xs[:, a], xs[:, b]), f"tdup={tdup}: frames {a} and {b} of the conv input are not bit-identical"
        if cache is not None:
            cc = cache if cache.dim() == 5 else cache[None]
            assert torch.equal(cc[:, 0], cc[:, 1]), "tdup: the conv cache is not an equal pair"
    if nb > 1:
        # dove_conv_desc.nb: nb independent instances back to back along the frame axis = nb separate calls (that IS the contract)
        assert out is None and gate is None and x.shape[0] % nb == 0
        xs = x.view(nb, x.shape[0] // nb, *x.shape[1:])
        rs = None if resid is None else resid.reshape(nb, -1, *resid.shape[1:])
        return torch.cat([conv(xs[b], pc, cache=None if cache is None else cache[b], stride=stride, pad=pad, up=up, tmode=tmode, t_out=t_out,
                               hw_out=hw_out, resid=None if rs is None else rs[b], act=act, ldo=ldo, gn_eps=gn_eps, out_f32=out_f32)
                          for b in range(nb)], dim=0)
    T, H, W, Cx = x.shape
    assert Cx == pc.cin_pad
    ph = (pc.kh - 1) // 2 if pad[0] is None else pad[0]
    pw = (pc.kw - 1) // 2 if pad[1] is None else pad[1]
    t_out = T if t_out is None else t_out
    if hw_out is None:
        hw_out = (H << up, W << up) if stride == 1 else ((H + 1 - pc.kh) // stride + 1, (W + 1 - pc.kw) // stride + 1)
    ldo = pc.cout_store if ldo is None else ldo
    wn, bn = _natural(pc)
    xf = x.float()[..., : pc.cin]
    if cache is not None:
        cache = cache[..., : pc.cin]
    if pc.kt > 1:
        k = pc.kt - 1
        front = cache.float() if cache is not None else xf[:1].expand(k, -1, -1, -1)
        xf = torch.cat([front, xf], dim=0)
    else:
        xf = xf[[_frame_index(t, tmode) for t in range(t_out)]]
    if up:
        xf = xf.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)
    He, We = xf.shape[1], xf.shape[2]
    # explicit zero padding so that output (oh, ow) reads input oh*stride + dh - pad
    need_h = (hw_out[0] - 1) * stride + pc.kh - ph
    need_w = (hw_out[1] - 1) * stride + pc.kw - pw
    xp = F.pad(xf.permute(3, 0, 1, 2)[None], (pw, max(need_w - We, 0), ph, max(need_h - He, 0)))
    y = F.conv3d(xp, wn, None, stride=(1, stride, stride))[0]           # [cout, T, H, W]
    y = y[:, :t_out, : hw_out[0], : hw_out[1]].permute(1, 2, 3, 0)
    if bn is not None:
        y = y + bn
    if act == 1:
        y = F.gelu(y, approximate="tanh")
    y = F.pad(y, (0, pc.cout_store - pc.cout))                           # channels cout..cout_store of the output are zero
    if resid is not None:
        r = resid.float().reshape(t_out, hw_out[0], hw_out[1], -1)[..., : pc.cout_store]
        if gate is not None:
            npx = t_out * hw_out[0] * hw_out[1]
            cls = (torch.arange(npx) >= gate_split).long().view(t_out, hw_out[0], hw_out[1])
            g = gate.float()[:, : pc.cout_store][cls]
            y = r + g * y
        else:
            y = r + y
    odt = torch.float32 if out_f32 else BF
    if out is None:
        out = torch.zeros(t_out, hw_out[0], hw_out[1], ldo, dtype=odt)
    out[..., : pc.cout_store] = y.to(odt)
    return out


def conv_out_gather(p, Cc, bias, dtype, scale=1.0, shift=0.0, lo=-math.inf, hi=math.inf):
    """dove_conv_out_gather: y[c][t][oy][ox] = sum over the 9 taps of p[t][oy+dy-1][ox+dx-1][(dy*3+dx)*Cc + c] (zero padded) + bias,
    rounded to bf16 (the conv's output dtype), then the range map."""
    T, H, W, _ = p.shape
    pp = F.pad(p.float().permute(0, 3, 1, 2), (1, 1, 1, 1))               # [T, ld, H+2, W+2]
    acc = torch.zeros(T, Cc, H, W)
    for dy in range(3):
        for dx in range(3):
            k = (dy * 3 + dx) * Cc
            acc += pp[:, k:k + Cc, dy:dy + H, dx:dx + W]
    v = (acc + bias.float().view(1, Cc, 1, 1)).to(BF).float()
    return (v * scale + shift).clamp(lo, hi).permute(1, 0, 2, 3).contiguous().to(dtype)


def conv_out_gather_cl(p, Cc, bias, ld=8):
    """dove_conv_out_gather_cl: the same 9-tap shifted sum, channels-last bf16 [T,H,W,ld] (channels >= Cc zero), no range map."""
    v = conv_out_gather(p, Cc, bias, BF)                                   # [Cc, T, H, W]
    T, H, W, _ = p.shape
    y = torch.zeros(T, H, W, ld, dtype=BF)
    y[..., :Cc] = v.permute(1, 2, 3, 0)
    return y


def tile_gather(x, t0, nt, th, tw, origins, im2col_cin=0):
    """dove_tile_gather_bf16: tiles of a channels-last clip as one tile-major batch; for an im2col'ed clip the taps that reach outside a
    tile are zeroed at its border pixels."""
    tiles = []
    for oy, ox in origins:
        t = x[t0:t0 + nt, oy:oy + th, ox:ox + tw].clone()
        if im2col_cin:
            c = im2col_cin
            for dy in range(3):
                for dx in range(3):
                    k = (dy * 3 + dx) * c
                    if dy == 0:
                        t[:, 0, :, k:k + c] = 0
                    if dy == 2:
                        t[:, th - 1, :, k:k + c] = 0
                    if dx == 0:
                        t[:, :, 0, k:k + c] = 0
                    if dx == 2:
                        t[:, :, tw - 1, k:k + c] = 0
        tiles.append(t)
    return torch.cat(tiles, dim=0).contiguous()


def linear(x, pc, **kw):
    N = x.shape[0]
    out = kw.pop("out", None)
    resid = kw.pop("resid", None)
    y = conv(x.view(1, 1, N, x.shape[1]), pc, resid=None if resid is None else resid.view(1, 1, N, resid.shape[1]),
             out=None if out is None else out.view(1, 1, N, out.shape[1]), **kw)
    return y.view(N, y.shape[-1])


def groupnorm_stats(x, eps, nb=1):
    if nb > 1:
        return torch.stack([groupnorm_stats(xb, eps) for xb in x.view(nb, x.shape[0] // nb, *x.shape[1:])])
    Cc = x.shape[-1]
    xf = x.double().reshape(-1, 32, Cc // 32)
    mean = xf.mean(dim=(0, 2))
    var = (xf * xf).mean(dim=(0, 2)) - mean * mean
    return torch.stack([mean, 1.0 / torch.sqrt(var.clamp_min(0) + eps)], dim=1).float()


def groupnorm_sums(x):
    Cc = x.shape[-1]
    xf = x.double().reshape(-1, 32, Cc // 32)
    return torch.stack([xf.sum(dim=(0, 2)), (xf * xf).sum(dim=(0, 2))], dim=1)


def groupnorm_sums_of(x):
    return groupnorm_sums(x)


def groupnorm_from_sums(sums, count, eps):
    if count is None:                      # the 65-double pair message of dove_amd.dist: count rides behind the sums
        count, sums = sums.reshape(-1)[64], sums.reshape(-1)[:64].reshape(32, 2)
    mean = sums[:, 0] / count
    var = (sums[:, 1] / count - mean * mean).clamp_min(0)
    return torch.stack([mean, 1.0 / torch.sqrt(var + eps)], dim=1).float()


def groupnorm_stats_of(x, eps, nb=1):
    return groupnorm_stats(x, eps, nb)


def groupnorm_apply(x, stats, gamma, beta, *, silu=True, yb=None, sshift=0, tmap=None, out=None, nb=1):
    if nb > 1:
        assert out is None
        xs = x.view(nb, x.shape[0] // nb, *x.shape[1:])
        ys = None if yb is None else yb.view(nb, yb.shape[0] // nb, *yb.shape[1:])
        return torch.cat([groupnorm_apply(xs[b], stats[b], gamma, beta, silu=silu, yb=None if ys is None else ys[b], sshift=sshift, tmap=tmap)
                          for b in range(nb)], dim=0)
    T, H, W, Cc = x.shape
    cpg = Cc // 32
    mean = stats[:, 0].repeat_interleave(cpg)
    rstd = stats[:, 1].repeat_interleave(cpg)
    sc = rstd * gamma.float()
    sh = beta.float() - mean * sc
    y = x.float() * sc + sh
    if yb is not None:
        ti = torch.tensor(tmap, dtype=torch.long)
        hi = torch.arange(H) >> sshift
        wi = torch.arange(W) >> sshift
        g = yb.float()[ti][:, hi][:, :, wi]
        y = y * g[..., :Cc] + g[..., Cc:]
    if silu:
        y = F.silu(y)
    y = y.to(BF)
    if out is not None:
        out.copy_(y)
        return out
    return y


def layernorm_modulate(x, gamma, beta, eps, mod=None, split=0, out=None):
    y = F.layer_norm(x.float(), (x.shape[1],), gamma.float(), beta.float(), eps)
    if mod is not None:
        cls = (torch.arange(x.shape[0]) >= split).long()
        m = mod.float()[cls]          # [N, 2, D]
        y = y * (1 + m[:, 1]) + m[:, 0]
    y = y.to(BF)
    if out is not None:
        out.copy_(y)
        return out
    return y


def _quad_swap_index(npad):
    """position s of a V^T row holds key _quad_swap_index[s]: every 16 keys stored [0-3, 8-11, 4-7, 12-15] (an involution)."""
    idx = torch.arange(npad)
    q = (idx >> 2) & 3
    return idx + 4 * (q == 1).long() - 4 * (q == 2).long()


def vt_quad_swap(Vt):
    Vt.copy_(Vt[..., _quad_swap_index(Vt.shape[-1])])
    return Vt


def ulysses_place(rq, rk, rv, counts, hloc, N, Npad, Qh, Kh, Vt, norm2_out=None):
    """dove_ulysses_place_bf16: blocks [rank i][hloc][c_i][64] / [hloc][64][c_i] -> [hloc][Npad][64] / [hloc][64][Npad] (quad-swapped, zero pad).
    With ``norm2_out`` every block has one extra row / column per head and the K block's extra row starts with that rank's two fp32 norms."""
    off = b = 0
    x = 0 if norm2_out is None else 1
    Vt.zero_()
    if x:
        norm2_out.zero_()
    for c in counts:
        n = (c + x) * hloc * 64
        Qh[:, b:b + c] = rq[off:off + n].view(hloc, c + x, 64)[:, :c]
        kb = rk[off:off + n].view(hloc, c + x, 64)
        Kh[:, b:b + c] = kb[:, :c]
        Vt[:, :, b:b + c] = rv[off:off + n].view(hloc, 64, c + x)[:, :, :c]
        if x:
            norm2_out.copy_(torch.maximum(norm2_out, kb[:, c, :4].contiguous().view(torch.float32)))
        off += n
        b += c
    vt_quad_swap(Vt)


def qkv_post(qkv, N, Npad, heads, text_len, gq, bq, gk, bk, cos, sin, qscale, eps, Qh, Kh, Vt, v_order=1, norm2=None):
    D = heads * 64
    q, k, v = (qkv.float()[:, i * D:(i + 1) * D].reshape(N, heads, 64) for i in range(3))
    q = F.layer_norm(q, (64,), gq.float(), bq.float(), eps)
    k = F.layer_norm(k, (64,), gk.float(), bk.float(), eps)

    def rope(t):
        tv = t[text_len:]
        xr, xi = tv.reshape(tv.shape[0], heads, 32, 2).unbind(-1)
        rot = torch.stack([-xi, xr], dim=-1).flatten(2)
        return torch.cat([t[:text_len], tv * cos.float()[:, None] + rot * sin.float()[:, None]], dim=0)

    if cos is not None:
        q, k = rope(q), rope(k)
    Qh[:, :N] = (q * qscale).permute(1, 0, 2).to(BF)
    Kh[:, :N] = k.permute(1, 0, 2).to(BF)
    Vt[:, :, :N] = v.permute(1, 2, 0).to(BF)
    if norm2 is not None:                                               # max squared norms of the STORED rows (include/dove_hip.h)
        norm2[:, 0] = (Qh[:, :N].float() ** 2).sum(-1).amax(-1)
        norm2[:, 1] = (Kh[:, :N].float() ** 2).sum(-1).amax(-1)
    if v_order == 1:
        assert Vt.shape[-1] % 16 == 0
        Vt[:, :, N:] = 0                                                # the swap moves tail keys into [N, Npad): keep the pad defined
        vt_quad_swap(Vt)


def attention(Qh, Kh, Vt, N, Npad, heads, out, norm2=None):
    """Vt in the quad-swapped key order (dove_attention_fwd_bf16's contract).  ``norm2`` only selects HOW the kernel shifts its softmax;
    the exact softmax below is the reference for both ways."""
    q, k, v = Qh.float()[:, :N], Kh.float()[:, :N], Vt.float()[:, :, _quad_swap_index(Vt.shape[-1])][:, :, :N]
    s = torch.einsum("hqd,hkd->hqk", q, k) * math.log(2.0)       # Qh carries scale*log2(e)
    p = torch.softmax(s, dim=-1)
    o = torch.einsum("hqk,hdk->hqd", p, v)                        # [H, N, 64]
    out[:N, : heads * 64] = o.permute(1, 0, 2).reshape(N, heads * 64).to(BF)
    return out


def _v8_store_index(npad):
    """position p of a V8t row holds key _v8_store_index[p]: inside every 32-key block the quads are stored [0,2,4,6,1,3,5,7]."""
    idx = torch.arange(npad)
    qs = (idx >> 2) & 7
    return (idx & ~31) | (((qs & 3) * 2 + (qs >> 2)) << 2) | (idx & 3)


def v8_store_order(x):
    """natural key order -> the stored order of dove_qkv_post_mxfp8 (last dim = keys)."""
    return x[..., _v8_store_index(x.shape[-1])]


def qkv_post_mx(qkv, N, Npad, heads, text_len, gq, bq, gk, bk, cos, sin, qscale, eps, Q8, K8, V8t, Vs):
    """dove_qkv_post_mxfp8: same pre-processing, e4m3 outputs (uint8 views); V in MXFP8 along the keys (32-key blocks per d row)."""
    Qf = torch.zeros(heads, Npad, 64)
    Kf = torch.zeros(heads, Npad, 64)
    Vf = torch.zeros(heads, 64, Npad)
    D = heads * 64
    q, k, v = (qkv.float()[:, i * D:(i + 1) * D].reshape(N, heads, 64) for i in range(3))
    q = F.layer_norm(q, (64,), gq.float(), bq.float(), eps)
    k = F.layer_norm(k, (64,), gk.float(), bk.float(), eps)

    def rope(t):
        tv = t[text_len:]
        xr, xi = tv.reshape(tv.shape[0], heads, 32, 2).unbind(-1)
        rot = torch.stack([-xi, xr], dim=-1).flatten(2)
        return torch.cat([t[:text_len], tv * cos.float()[:, None] + rot * sin.float()[:, None]], dim=0)

    if cos is not None:
        q, k = rope(q), rope(k)
    Qf[:, :N] = (q * (qscale * 8.0)).permute(1, 0, 2)
    Kf[:, :N] = k.permute(1, 0, 2)
    Vf[:, :, :N] = v.permute(1, 2, 0)
    Q8.copy_(Qf.to(torch.float8_e4m3fn).view(torch.uint8))
    K8.copy_(Kf.to(torch.float8_e4m3fn).view(torch.uint8))
    vq, ve = mx_quant_ref(Vf.reshape(heads * 64, Npad))                  # blocks of 32 consecutive keys
    V8t.copy_(v8_store_order(vq.view(torch.uint8).reshape(heads, 64, Npad)))
    Vs.copy_(ve.reshape(heads, 64, Npad // 64, 2).permute(0, 2, 1, 3))


def attention_mx(Q8, K8, V8t, Vs, N, Npad, heads, out, thr=6.0):
    """dove_attention_fwd_mxfp8 restated tile by tile.  Running max m as in the kernel: set by the first tile, afterwards moved
    (for all 32 queries of a wave) only when one of them sees a score above m + thr.  The probabilities of each (query, 64-key
    tile) are e4m3 values of 2^(s - m - e), e = ceil(max_tile(s) - m) - 8; the row sum uses the unquantised values."""
    q = torch.zeros(heads, Npad, 64)
    q[:, :N] = Q8.view(torch.float8_e4m3fn).float()[:, :N] * 0.125
    k = K8.view(torch.float8_e4m3fn).float()
    ve = Vs.permute(0, 2, 1, 3).reshape(heads * 64, Npad // 32)
    inv = torch.empty(Npad, dtype=torch.long)
    inv[_v8_store_index(Npad)] = torch.arange(Npad)                      # natural key k sits at stored position inv[k]
    v8 = V8t[..., inv].contiguous()
    v = mx_dequant(v8.view(torch.float8_e4m3fn).reshape(heads * 64, Npad), ve).reshape(heads, 64, Npad)
    m = torch.zeros(heads, Npad)
    l = torch.zeros(heads, Npad)
    o = torch.zeros(heads, Npad, 64)
    for t0 in range(0, N, 64):
        t1 = min(t0 + 64, N)
        s = torch.einsum("hqd,hkd->hqk", q, k[:, t0:t1]) - m[..., None]
        mt = s.amax(dim=2)
        if t0 == 0:
            delta, alpha = mt, torch.ones_like(mt)
        else:
            fire = (mt > thr).reshape(heads, Npad // 32, 32).any(dim=2, keepdim=True).expand(-1, -1, 32).reshape(heads, Npad)
            delta = torch.where(fire, mt.clamp_min(0.0), torch.zeros_like(mt))
            alpha = torch.exp2(-delta)
        m = m + delta
        s = s - delta[..., None]
        mt = mt - delta
        e = torch.ceil(mt).clamp_min(-100.0) - 8.0
        p = torch.exp2(s)
        l = l * alpha + p.sum(dim=2)
        pq = (p * torch.exp2(-e)[..., None]).to(torch.float8_e4m3fn).float()
        o = o * alpha[..., None] + torch.einsum("hqk,hdk->hqd", pq, v[:, :, t0:t1]) * torch.exp2(e)[..., None]
    o = (o / l[..., None])[:, :N]
    out[:N, : heads * 64] = o.permute(1, 0, 2).reshape(N, heads * 64).to(BF)
    return out


def cl_from_ncthw(x, cp, scale=1.0, shift=0.0):
    Cc, T, H, W = x.shape
    y = torch.zeros(T, H, W, cp, dtype=BF)
    y[..., :Cc] = (x.float() * scale + shift).permute(1, 2, 3, 0).to(BF)
    return y


def cl_im2col3x3_from_ncthw(x, cp, scale=1.0, shift=0.0):
    Cc, T, H, W = x.shape
    xp = F.pad(x.float() * scale + shift, (1, 1, 1, 1))                  # [C, T, H+2, W+2], zeros outside the frame
    y = torch.zeros(T, H, W, cp, dtype=BF)
    for dy in range(3):
        for dx in range(3):
            k = (dy * 3 + dx) * Cc
            y[..., k:k + Cc] = xp[:, :, dy:dy + H, dx:dx + W].permute(1, 2, 3, 0).to(BF)
    return y


def ncthw_from_cl(x, Cc, dtype, scale=1.0, shift=0.0, lo=-math.inf, hi=math.inf):
    return (x.float()[..., :Cc] * scale + shift).clamp(lo, hi).permute(3, 0, 1, 2).contiguous().to(dtype)


def avgpool_time(x, nb=1):
    if nb > 1:
        return torch.cat([avgpool_time(xb) for xb in x.view(nb, x.shape[0] // nb, *x.shape[1:])], dim=0)
    T = x.shape[0]
    if T == 1:
        return x
    xf = x.float()
    if T % 2:
        return torch.cat([xf[:1], 0.5 * (xf[1::2] + xf[2::2])], dim=0).to(BF)
    return (0.5 * (xf[0::2] + xf[1::2])).to(BF)


def posterior_sample(moments_cl, latent_channels, noise, dtype):
    L = latent_channels
    m = moments_cl.float()
    mean, lv = m[..., :L].permute(3, 0, 1, 2), m[..., L:2 * L].permute(3, 0, 1, 2).clamp(-30.0, 20.0)
    return (mean + torch.exp(0.5 * lv) * noise.float()).to(dtype)


def axpby(x, y, a, b):
    return (a * x.float() + b * y.float()).to(x.dtype)


def patchify(x, pt, p, ld):
    T, Cc, H, W = x.shape
    t = x.float().reshape(T // pt, pt, Cc, H // p, p, W // p, p).permute(0, 3, 5, 2, 1, 4, 6).reshape(-1, Cc * pt * p * p)
    tok = torch.zeros(t.shape[0], ld, dtype=BF)
    tok[:, : t.shape[1]] = t.to(BF)
    return tok


def unpatchify(tok, T, Cc, H, W, pt, p, dtype):
    t = tok.float()[:, : Cc * pt * p * p].reshape(T // pt, H // p, W // p, Cc, pt, p, p)
    return t.permute(0, 4, 3, 1, 5, 2, 6).reshape(T, Cc, H, W).to(dtype)


def gemv(W, bias, x, act_in=0):
    xv = F.silu(x.float()) if act_in == 1 else x.float()
    y = W.float() @ xv
    return y + bias.float() if bias is not None else y


def blend_edge(a, b, extent, axis):
    af, bf = a.float(), b.float()
    for e in range(extent):
        wb = e / extent
        if axis == 0:
            bf[:, e] = af[:, a.shape[1] - extent + e] * (1 - wb) + bf[:, e] * wb
        else:
            bf[:, :, e] = af[:, :, a.shape[2] - extent + e] * (1 - wb) + bf[:, :, e] * wb
    b.copy_(bf.to(BF))
    return b


def preprocess_u8(frames, pad_f, pad_h, pad_w, upscale, dtype):
    """The reference's own torch code path (ref :220-232, :672-679) restated."""
    f = frames
    if pad_f:
        f = torch.cat([f, f[-1:].repeat(pad_f, 1, 1, 1)], dim=0)
    f = F.pad(f, (0, 0, 0, pad_w, 0, pad_h))
    v = f.float().permute(0, 3, 1, 2)
    v = F.interpolate(v, size=(v.shape[2] * upscale, v.shape[3] * upscale), mode="bilinear", align_corners=False)
    return (v / 255.0 * 2.0 - 1.0).permute(1, 0, 2, 3).contiguous().to(dtype)


def postprocess_u8(video, Fo, Ho, Wo):
    v = video[:, :Fo, :Ho, :Wo].permute(1, 2, 3, 0)
    return (v.float() * 255).clamp(0, 255).to(torch.uint8).contiguous()


@dataclass
class EmuMx:
    """Emulation-side MXFP8 operand: dequantised values (what the block-scaled MFMA multiplies) + optional bias."""
    deq: torch.Tensor
    rows: int
    K: int
    bias: torch.Tensor | None = None


def mx_quant(x):
    q, e = mx_quant_ref(x)
    return EmuMx(mx_dequant(q, e), x.shape[0], x.shape[1])


def pack_linear_mx(weight, bias, device="cpu"):
    pm = mx_quant(weight.detach().to(BF))
    pm.bias = None if bias is None else bias.detach().float()
    return pm


def linear_mx(x, w, *, resid=None, gate=None, gate_split=0, act=0, out=None):
    y = x.deq @ w.deq.t()
    if w.bias is not None:
        y = y + w.bias
    if act == 1:
        y = F.gelu(y, approximate="tanh")
    if resid is not None:
        if gate is not None:
            cls = (torch.arange(x.rows) >= gate_split).long()
            y = resid.float() + gate.float()[cls] * y
        else:
            y = resid.float() + y
    y = y.to(BF)
    if out is not None:
        out.copy_(y)
        return out
    return y


def rmsnorm(x, weight, eps):
    xf = x.float()
    return (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps) * weight.float()).to(BF)


def gated_gelu(x):
    Fh = x.shape[1] // 2
    return (F.gelu(x.float()[:, :Fh], approximate="tanh").to(BF).float() * x.float()[:, Fh:]).to(BF)


def attention_bias(qkv, bias, heads):
    N, D = qkv.shape[0], heads * 64
    q, k, v = (qkv.float()[:, i * D:(i + 1) * D].reshape(N, heads, 64).permute(1, 0, 2) for i in range(3))
    s = (torch.einsum("hqd,hkd->hqk", q, k).to(BF).float() + bias.to(BF).float()).to(BF).float()
    p = torch.softmax(s, dim=-1).to(BF).float()
    return torch.einsum("hqk,hkd->hqd", p, v).permute(1, 0, 2).reshape(N, D).to(BF)


ALL = ["rmsnorm", "gated_gelu", "attention_bias", "mx_quant", "pack_linear_mx", "linear_mx", "groupnorm_sums", "groupnorm_sums_of", "groupnorm_from_sums", "groupnorm_stats_of", "blend_edge", "preprocess_u8", "postprocess_u8", "conv", "conv_out_gather", "conv_out_gather_cl", "tile_gather", "linear", "groupnorm_stats", "groupnorm_apply", "layernorm_modulate", "qkv_post", "attention", "vt_quad_swap", "ulysses_place", "qkv_post_mx", "attention_mx",
       "cl_from_ncthw", "cl_im2col3x3_from_ncthw", "ncthw_from_cl", "avgpool_time", "posterior_sample", "axpby", "patchify", "unpatchify", "gemv"]


def install(monkeypatch):
    """Route dove_amd.ops through this emulation (CPU tests of the host graph only)."""
    import sys

    import dove_amd.ops as real

    me = sys.modules[__name__]
    for n in ALL:
        monkeypatch.setattr(real, n, getattr(me, n))


# ---- MXFP8 (OCP microscaling) restatement: per 32-element block one power-of-two scale, elements e4m3fn ---------------------
def mx_quant_ref(x):
    """x [rows, K] -> (q float8_e4m3fn [rows, K], e uint8 [rows, K/32] biased E8M0 exponents): scale = 2^ceil(log2(amax/448))."""
    rows, K = x.shape
    xb = x.float().reshape(rows, K // 32, 32)
    amax = xb.abs().amax(dim=2)
    r = amax / 448.0
    e = torch.where(amax > 0, torch.ceil(torch.log2(r.clamp_min(1e-45))).clamp(-127, 127) + 127, torch.zeros_like(r))
    # exact powers of two: log2 is exact for them in fp32, ceil leaves them alone
    scale = torch.pow(2.0, e - 127)
    q = (xb / scale[..., None]).to(torch.float8_e4m3fn)
    return q.reshape(rows, K), e.to(torch.uint8)


def mx_scale_words(e):
    """[rows, K/32] exponents -> the kernel's layout int32 [K/256][rows][2]: word (c, r, h), byte u = block 8c + 2u + h."""
    rows, nb = e.shape
    w = e.reshape(rows, nb // 8, 4, 2).permute(1, 0, 3, 2).contiguous().to(torch.int64)       # [c][r][h][u]
    return (w[..., 0] | (w[..., 1] << 8) | (w[..., 2] << 16) | (w[..., 3] << 24)).to(torch.int64).to(torch.int32)


def mx_dequant(q, e):
    rows, K = q.shape
    return (q.float().reshape(rows, K // 32, 32) * torch.pow(2.0, e.float() - 127)[..., None]).reshape(rows, K)


def linear_mx_ref(x, weight, bias, *, resid=None, gate=None, gate_split=0, act=0):
    """bf16 x [M,K], weight [N,K] (any float): quantise both to MXFP8, multiply the dequantised values in fp32."""
    xq, xe = mx_quant_ref(x)
    wq, we = mx_quant_ref(weight.to(BF))
    y = mx_dequant(xq, xe) @ mx_dequant(wq, we).t()
    if bias is not None:
        y = y + bias.float()
    if act == 1:
        y = F.gelu(y, approximate="tanh")
    if resid is not None:
        if gate is not None:
            cls = (torch.arange(x.shape[0]) >= gate_split).long()
            y = resid.float() + gate.float()[cls] * y
        else:
            y = resid.float() + y
    return y.to(BF)
