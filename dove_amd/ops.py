"""torch-tensor front end of the C-ABI operators (allocation + argument marshalling only; all arithmetic is
in libdove_hip.so).  Activations are channels-last bf16: [T,H,W,C] (VAE) or [N,C] (DiT tokens)."""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass

import torch

from . import lib as L


def _ru(x: int, m: int) -> int:
    return (x + m - 1) // m * m


@dataclass
class PackedConv:
    """Weights repacked for the implicit-GEMM kernel: [taps][cout_pad][cin_pad] bf16, bias fp32 [cout_pad]."""
    w: torch.Tensor
    bias: torch.Tensor | None
    kt: int
    kh: int
    kw: int
    cin: int
    cin_pad: int
    cout: int
    cout_pad: int
    # kt == 3: [2][kh*kw][cout_pad][cin_pad] temporal sums w0 + w1 and w0 + w1 + w2 of the bf16 weights (fp32 sums, one rounding): what a causal
    # conv without a cache needs for its first two output frames, whose early taps all read the replicated frame 0 (dove_conv_desc.w_first)
    w_first: torch.Tensor | None = None
    # 3x3, kt == 1: [4 phases][2x2 taps][cout_pad][cin_pad] - the sub-pixel form of an upsample-fused conv (dove_conv_desc.w_sub): per output
    # phase the 3x3 weights summed over the taps that read the same low-res pixel (fp32 sums of the bf16 weights, one rounding)
    w_sub: torch.Tensor | None = None
    # kt == 3, on request: [2][kh*kw][cout_pad][cin_pad] - w0 + w1 and w1 + w2 for a conv whose input frames come in bit-identical pairs
    # (dove_conv_desc.w_pair / tdup: the first causal conv behind Upsample3D's time doubling); fp32 sums of the bf16 weights, one rounding
    w_pair: torch.Tensor | None = None

    @property
    def cout_store(self) -> int:
        return _ru(self.cout, 4)


def pack_conv(weight: torch.Tensor, bias: torch.Tensor | None, device, *, sub: bool = True, pair: bool = False) -> PackedConv:
    """weight: Conv3d [Cout,Cin,kt,kh,kw], Conv2d [Cout,Cin,kh,kw] or Linear [Cout,Cin] (any float dtype).
    ``sub``: build the sub-pixel sums of a 3x3 Conv2d (only an upsample-fused use reads them: the VAE passes False for its down-samplers);
    ``pair``: build the pair sums of a 3x3x3 conv (``conv(..., tdup=)``)."""
    w = weight.detach()
    if w.dim() == 2:
        w = w[:, :, None, None, None]
    elif w.dim() == 4:
        w = w[:, :, None]
    cout, cin, kt, kh, kw = w.shape
    cin_pad = _ru(cin, 32) if cin <= 32 or cin % 64 else cin
    cout_pad = _ru(cout, 32)
    wp = torch.zeros(kt * kh * kw, cout_pad, cin_pad, dtype=torch.bfloat16, device=device)
    wp[:, :cout, :cin] = w.to(device=device, dtype=torch.float32).permute(2, 3, 4, 0, 1).reshape(kt * kh * kw, cout, cin).to(torch.bfloat16)
    bp = None
    if bias is not None:
        bp = torch.zeros(cout_pad, dtype=torch.float32, device=device)
        bp[:cout] = bias.detach().to(device=device, dtype=torch.float32)
    wp = wp.contiguous()
    w_first = None
    if kt == 3:
        t = wp.float().view(3, kh * kw, cout_pad, cin_pad)
        s01 = t[0] + t[1]
        w_first = torch.stack([s01, s01 + t[2]]).to(torch.bfloat16).contiguous()
    w_pair = None
    if kt == 3 and pair:
        t = wp.float().view(3, kh * kw, cout_pad, cin_pad)
        w_pair = torch.stack([t[0] + t[1], t[1] + t[2]]).to(torch.bfloat16).contiguous()
    w_sub = None
    if kt == 1 and kh == 3 and kw == 3 and sub:
        t = wp.float().view(3, 3, cout_pad, cin_pad)
        w_sub = torch.zeros(4, 4, cout_pad, cin_pad, dtype=torch.float32, device=device)
        for py in range(2):
            for px in range(2):
                for dh in range(3):                       # fixed summation order (csrc/graph.hip pack_sub_kernel: the same)
                    for dw in range(3):
                        a, b = (py + dh + 1) // 2 - py, (px + dw + 1) // 2 - px
                        w_sub[2 * py + px, 2 * a + b] += t[dh, dw]
        w_sub = w_sub.to(torch.bfloat16).contiguous()
    return PackedConv(wp, bp, kt, kh, kw, cin, cin_pad, cout, cout_pad, w_first, w_sub, w_pair)


def temporal_split(tl: int, tdup: int, cached: bool, first_sums: bool = True):
    """Host restatement of csrc/igemm.hip h4_split + h4_set_nxt: the temporal groups conv3x3_halo4x runs for output frame tl of an instance,
    as [(weights, source frame)] with weights in {"w0", "w1", "w2", "w01", "w12", "w012"} (tap blocks of w / their pack-time sums) and the
    source an instance frame index (negative: before the first frame = the conv cache, else frame 0 replicated).
    tdup 0: no declaration (only the cache-less first two frames split, when ``first_sums``); 1: frames are pairs (0,1), (2,3), ... and the
    cache too; 2: frame 0 single, then pairs (1,2), (3,4), ... (no cache)."""
    plain = [("w0", tl - 2), ("w1", tl - 1), ("w2", tl)]
    if tdup == 0:
        if cached or not first_sums or tl > 1:
            return plain
        return [("w012", 0)] if tl == 0 else [("w01", 0), ("w2", 1)]
    if tdup == 1:
        if not cached and tl < 2:
            return [("w012", tl)]
        return [("w0", tl - 2), ("w12", tl)] if tl & 1 else [("w01", tl - 1), ("w2", tl)]
    if tl == 0:
        return [("w012", 0)]
    return [("w01", tl - 1), ("w2", tl)] if tl & 1 else [("w0", tl - 2), ("w12", tl)]


def temporal_groups(tl: int, tdup: int, cached: bool) -> int:
    return len(temporal_split(tl, tdup, cached))


_profiler = None


def set_profiler(records: list | None):
    """bench.py hook: when a list is installed, every igemm launch appends (key, algorithmic_flops, ev_start, ev_end, kernel_name,
    issued_flops) with HIP events recorded on the launch stream (issued < algorithmic where a weight-summed form skips duplicate taps)."""
    global _profiler
    _profiler = records


def conv_kernel_name(x_shape, pc: PackedConv, *, stride=1, pad=(None, None), up=0, tmode=0, t_out=None, hw_out=None, act=0, gated=False,
                     resid=False, nb=1, partial=False):
    """Kernel a conv / linear of this shape dispatches to (dove_conv_kernel_name; pure function of the descriptor, no GPU needed).
    ``partial=True``: the partial-tile launch mask of the call instead (dove_conv_partial_launches: 1 = last tile column, 2 = last tile row);
    ``x_shape`` is per instance, ``nb`` instances."""
    T, H, W, Cx = x_shape
    d = L.ConvDesc()
    d.nb = nb
    ph = (pc.kh - 1) // 2 if pad[0] is None else pad[0]
    pw = (pc.kw - 1) // 2 if pad[1] is None else pad[1]
    t_out = T if t_out is None else t_out
    if hw_out is None:
        hw_out = (H << up, W << up) if stride == 1 else ((H + 1 - pc.kh) // stride + 1, (W + 1 - pc.kw) // stride + 1)
    d.t_in, d.h_in, d.w_in, d.cin = T, H, W, Cx
    d.t_out, d.h_out, d.w_out = t_out, hw_out[0], hw_out[1]
    d.cout_pad, d.cout_store = pc.cout_pad, pc.cout_store
    d.kt, d.kh, d.kw, d.stride, d.pad_h, d.pad_w = pc.kt, pc.kh, pc.kw, stride, ph, pw
    d.up, d.tmode, d.act = up, tmode, act
    d.ldo = pc.cout_store
    d.ldr = pc.cout_store if resid else 0
    d.resid = 1 if (resid or gated) else None                  # only tested for NULL-ness by the selection rule
    d.w_sub = 1 if (up == 1 and pc.kt == 1 and pc.kh == 3 and pc.kw == 3) else None
    d.gate = 1 if gated else None
    if partial:
        return int(L.load().dove_conv_partial_launches(C.byref(d)))
    return L.load().dove_conv_kernel_name(C.byref(d)).decode()


def conv(x: torch.Tensor, pc: PackedConv, *, cache: torch.Tensor | None = None, stride: int = 1, pad=(None, None),
         up: int = 0, tmode: int = 0, t_out: int | None = None, hw_out=None, resid: torch.Tensor | None = None,
         gate: torch.Tensor | None = None, gate_split: int = 0, act: int = 0, ldo: int | None = None,
         out: torch.Tensor | None = None, debug_buf: torch.Tensor | None = None,
         gn_eps: float | None = None, out_f32: bool = False, nb: int = 1, tdup: int = 0, weight_sums: bool = True) -> torch.Tensor:
    """Implicit-GEMM conv on channels-last x [T,H,W,cin_pad] -> [t_out,h_out,w_out,ldo].

    ``gn_eps``: the output feeds an nn.GroupNorm(32, C, eps) (resnet norm1/norm2, SpatialNorm's norm_layer).  When the
    dispatched kernel can fuse the statistics into its epilogue (``dove_conv_gn_partial_rows`` > 0), they are computed
    there and attached to the returned tensor as ``out.gn_stats`` ([32,2] mean / rstd, fp32) - `groupnorm_stats_of`
    then skips the separate pass over the tensor; otherwise nothing is attached and the caller's path is unchanged.

    ``nb`` > 1: x holds nb independent instances back to back along the frame axis ([nb*T, H, W, C]; ``t_out`` stays the per-instance
    count) - the same-shaped tiles of the tiled VAE in one launch (include/dove_hip.h dove_conv_desc.nb).  ``cache`` is then
    [nb, kt-1, H, W, C], possibly a strided view (dim 0) of the previous frame-batch's input; statistics come back as [nb, 32, 2].

    ``tdup`` (kt == 3, ``pc.w_pair`` packed): the caller declares that every instance's frames come in bit-identical pairs - 1: (0,1), (2,3), ...
    and the cache too; 2: frame 0 single, then (1,2), (3,4), ..., no cache (include/dove_hip.h dove_conv_desc.tdup): two temporal groups per
    frame instead of three.  ``weight_sums=False``: none of the pack-time weight sums (w_first / w_sub / w_pair) is handed over - the launch
    computes the reference's per-tap arithmetic."""
    L.require_cuda(x, resid, gate, out)
    assert x.dtype == torch.bfloat16 and x.dim() == 4, (x.dtype, x.shape)
    T, H, W, Cx = x.shape
    assert T % nb == 0, (T, nb)
    T //= nb
    if Cx != pc.cin_pad:
        raise RuntimeError(f"conv: input has {Cx} channels, packed weight expects {pc.cin_pad}")
    ph = (pc.kh - 1) // 2 if pad[0] is None else pad[0]
    pw = (pc.kw - 1) // 2 if pad[1] is None else pad[1]
    if t_out is None:
        t_out = T
    if hw_out is None:
        hw_out = (H << up, W << up) if stride == 1 else ((H + 1 - pc.kh) // stride + 1, (W + 1 - pc.kw) // stride + 1)
    if ldo is None:
        ldo = pc.cout_store
    odt = torch.float32 if out_f32 else torch.bfloat16        # out_f32: un-rounded accumulators (tap-split conv_out, conv_out_gather)
    if out is None:
        out = torch.empty(nb * t_out, hw_out[0], hw_out[1], ldo, dtype=odt, device=x.device)
    else:
        assert out.shape == (nb * t_out, hw_out[0], hw_out[1], ldo) and out.dtype == odt
    cache_stride = 0
    if cache is not None:
        if nb > 1:
            assert cache.shape == (nb, pc.kt - 1, H, W, Cx) and cache.dtype == torch.bfloat16 and cache.is_cuda, (cache.shape, x.shape)
            assert cache[0].is_contiguous() and cache.device == x.device
            cache_stride = cache.stride(0)
        else:
            L.require_cuda(cache)
            assert cache.shape == (pc.kt - 1, H, W, Cx) and cache.dtype == torch.bfloat16, (cache.shape, x.shape)
    d = L.ConvDesc()
    d.nb, d.cache_stride = nb, cache_stride
    if cache is None and pc.w_first is not None and weight_sums:
        d.w_first = pc.w_first.data_ptr()
    if up == 1 and pc.w_sub is not None and weight_sums:
        d.w_sub = pc.w_sub.data_ptr()
    if tdup and weight_sums and pc.w_pair is not None and pc.kt == 3 and not (tdup == 2 and cache is not None):
        d.tdup, d.w_pair = tdup, pc.w_pair.data_ptr()
    d.x, d.cache, d.w = x.data_ptr(), (cache.data_ptr() if cache is not None else None), pc.w.data_ptr()
    d.bias = pc.bias.data_ptr() if pc.bias is not None else None
    d.resid = resid.data_ptr() if resid is not None else None
    d.gate = gate.data_ptr() if gate is not None else None
    d.out = out.data_ptr()
    d.t_in, d.h_in, d.w_in, d.cin = T, H, W, Cx
    d.t_out, d.h_out, d.w_out = t_out, hw_out[0], hw_out[1]
    d.cout_pad, d.cout_store = pc.cout_pad, pc.cout_store
    d.kt, d.kh, d.kw, d.stride, d.pad_h, d.pad_w = pc.kt, pc.kh, pc.kw, stride, ph, pw
    d.up, d.tmode, d.act = up, tmode, act
    d.ldo = ldo
    d.ldr = resid.shape[-1] if resid is not None else 0
    d.gate_split = gate_split
    if debug_buf is not None:      # tools/*_timing.py only: exists in libdove_hip_timing.so, an AttributeError on the product library
        L.load().dove_timing_set_debug_buf(C.c_void_p(debug_buf.data_ptr()))
    d.out_f32 = int(out_f32)
    if resid is not None:
        assert resid.dtype == torch.bfloat16 and resid.numel() == nb * t_out * hw_out[0] * hw_out[1] * resid.shape[-1]
    if gate is not None:
        assert gate.dtype == torch.float32 and gate.shape == (2, pc.cout_pad)
    partial = None
    if gn_eps is not None and ldo == pc.cout_store:
        rows = int(L.load().dove_conv_gn_partial_rows(C.byref(d)))
        if rows > 0:
            partial = torch.empty(rows, 64, dtype=torch.float32, device=x.device)
            d.gn_partial = partial.data_ptr()
    if _profiler is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    L.check(L.load().dove_conv_igemm_bf16(C.byref(d), L.stream_ptr()), "dove_conv_igemm_bf16")
    if _profiler is not None:
        e1.record()
        flops = 2.0 * nb * t_out * hw_out[0] * hw_out[1] * pc.cout * pc.cin * pc.kt * pc.kh * pc.kw
        name = L.load().dove_conv_kernel_name(C.byref(d)).decode()
        # MFMA work actually issued: the ALGORITHMIC count above is the reference's formulation; the weight-summed forms skip taps whose
        # input is a duplicate (w_first: 2 + 1 of the first two frames' 3 + 3 temporal taps per instance; w_sub: 5 of 9 spatial taps)
        real = flops
        if name == "conv3x3_halo4x_kernel":
            if d.tdup and pc.kt == 3 and up == 0:
                real = flops * sum(temporal_groups(tl, d.tdup, cache is not None) for tl in range(t_out)) / (3.0 * t_out)
            elif d.w_first and pc.kt == 3 and up == 0:
                real = flops * (1.0 - (2.0 + (1.0 if t_out > 1 else 0.0)) / (3.0 * t_out))
            elif d.w_sub and up == 1 and H >= 16 and W >= 32:
                real = flops * 4.0 / 9.0
        _profiler.append(((pc.cin, pc.cout, pc.kt * pc.kh * pc.kw), flops, e0, e1, name, real))
    if partial is None:
        if getattr(out, "gn_stats", None) is not None:      # a re-used `out` tensor must not keep statistics of old contents
            out.gn_stats = None
            out.gn_rows = None
    else:
        count = float(t_out * hw_out[0] * hw_out[1]) * (pc.cout_store // 32)
        if nb > 1:
            stats = torch.empty(nb, 32, 2, dtype=torch.float32, device=x.device)
            ws = _ws(x.device, nb * 512)                       # nb x 256 rows of 64 doubles
            L.check(L.load().dove_groupnorm_finalize_partials_nb(partial.data_ptr(), partial.shape[0] // nb, nb, count, gn_eps, L.ptr(ws),
                                                                 ws.numel() * 4, L.ptr(stats), L.stream_ptr()), "dove_groupnorm_finalize_partials_nb")
        else:
            stats = torch.empty(32, 2, dtype=torch.float32, device=x.device)
            L.check(L.load().dove_groupnorm_finalize_partials(partial.data_ptr(), partial.shape[0], count, gn_eps, L.ptr(_ws(x.device)),
                                                              L.ptr(stats), L.stream_ptr()), "dove_groupnorm_finalize_partials")
        out.gn_stats = (stats, gn_eps, out._version)           # torch's in-place counter: a later torch write voids them
        out.gn_rows = partial                                  # raw per-tile sums: dove_amd.dist combines them across a rank pair
    return out


def groupnorm_stats_of(x: torch.Tensor, eps: float, nb: int = 1) -> torch.Tensor:
    """GroupNorm(32) statistics of x: the ones its producing conv already computed (``conv(..., gn_eps=eps)``), else a
    pass over x.  ``nb`` > 1: x is nb instances back to back, one statistics scope each ([nb, 32, 2])."""
    have = getattr(x, "gn_stats", None)
    if have is not None and have[1] == eps and have[2] == x._version and have[0].numel() == nb * 64:
        return have[0]
    return groupnorm_stats(x, eps, nb)


def linear(x: torch.Tensor, pc: PackedConv, **kw) -> torch.Tensor:
    """x [N, cin_pad] -> [N, ldo]."""
    N = x.shape[0]
    out = kw.pop("out", None)
    resid = kw.pop("resid", None)
    y = conv(x.view(1, 1, N, x.shape[1]), pc, resid=None if resid is None else resid.view(1, 1, N, resid.shape[1]),
             out=None if out is None else out.view(1, 1, N, out.shape[1]), **kw)
    return y.view(N, y.shape[-1])


_gn_ws: dict = {}
GN_WS_ROWS = 4096        # partial rows of scratch: 9 frames x 256 blocks per frame at most (csrc/norm.hip gn_partial_launch)


def _ws(device, rows: int = GN_WS_ROWS):
    # one scratch buffer per (device, stream): the two-stream VAE mode runs GroupNorm statistics concurrently.  It only ever grows
    # (launches that used the old buffer are ordered before anything that re-uses its memory on the same stream)
    key = (str(device), torch.cuda.current_stream(device).cuda_stream if torch.cuda.is_available() else 0)
    rows = max(rows, GN_WS_ROWS)
    if key not in _gn_ws or _gn_ws[key].numel() < rows * 64:
        _gn_ws[key] = torch.empty(rows * 64, dtype=torch.float32, device=device)
    return _gn_ws[key]


def _gn_rows_needed(x, frames: int) -> int:
    """Partial rows csrc/norm.hip gn_partial_launch writes for `frames` frames of x's frame size (at most 256 blocks per frame)."""
    fp = _frame_pix(x) or x.numel() // x.shape[-1]
    nsub = 256 // max(x.shape[-1] // 8, 1)
    return frames * min(256, -(-fp // (max(nsub, 1) * 32)))


def _frame_pix(x):
    """H*W of a [T,H,W,C] frame-batch (0 = "one frame" for anything else): the statistics kernels sum per frame share."""
    return x.shape[1] * x.shape[2] if x.dim() == 4 else 0


def groupnorm_stats(x: torch.Tensor, eps: float, nb: int = 1) -> torch.Tensor:
    """x [T,H,W,C] (one frame-batch) -> stats [32,2] (mean, rstd) fp32; ``nb`` > 1: [nb*T,H,W,C] -> [nb,32,2], one scope per instance."""
    L.require_cuda(x)
    assert x.dtype == torch.bfloat16
    Cc = x.shape[-1]
    frames = x.shape[0] if x.dim() == 4 else 1
    ws = _ws(x.device, _gn_rows_needed(x, frames))
    if nb > 1:
        assert x.dim() == 4 and x.shape[0] % nb == 0
        stats = torch.empty(nb, 32, 2, dtype=torch.float32, device=x.device)
        L.check(L.load().dove_groupnorm_stats_nb_bf16(L.ptr(x), nb, x.numel() // Cc // nb, _frame_pix(x), Cc, eps, L.ptr(ws), ws.numel() // 64,
                                                      L.ptr(stats), L.stream_ptr()), "dove_groupnorm_stats_nb_bf16")
        return stats
    stats = torch.empty(32, 2, dtype=torch.float32, device=x.device)
    L.check(L.load().dove_groupnorm_stats_bf16(L.ptr(x), x.numel() // Cc, _frame_pix(x), Cc, eps, L.ptr(ws), ws.numel() // 64, L.ptr(stats),
                                               L.stream_ptr()), "dove_groupnorm_stats_bf16")
    return stats


def groupnorm_sums(x: torch.Tensor) -> torch.Tensor:
    """Raw per-group (sum, sum of squares) of one piece of a frame-batch, fp64 [32,2] (dove_amd.dist adds the pieces)."""
    L.require_cuda(x)
    assert x.dtype == torch.bfloat16
    Cc = x.shape[-1]
    sums = torch.empty(32, 2, dtype=torch.float64, device=x.device)
    ws = _ws(x.device, _gn_rows_needed(x, x.shape[0] if x.dim() == 4 else 1))
    L.check(L.load().dove_groupnorm_sums_bf16(L.ptr(x), x.numel() // Cc, _frame_pix(x), Cc, L.ptr(ws), ws.numel() // 64, L.ptr(sums),
                                              L.stream_ptr()), "dove_groupnorm_sums_bf16")
    return sums


def groupnorm_sums_of(x: torch.Tensor) -> torch.Tensor:
    """``groupnorm_sums`` of x, from the partial rows its producing conv wrote when there are any (no pass over x)."""
    rows = getattr(x, "gn_rows", None)
    if rows is None:
        return groupnorm_sums(x)
    sums = torch.empty(32, 2, dtype=torch.float64, device=x.device)
    L.check(L.load().dove_groupnorm_sums_from_partials(L.ptr(rows), rows.shape[0], L.ptr(_ws(x.device)), L.ptr(sums), L.stream_ptr()),
            "dove_groupnorm_sums_from_partials")
    return sums


def groupnorm_from_sums(sums: torch.Tensor, count: float, eps: float) -> torch.Tensor:
    """stats [32,2] (mean, rstd) from summed fp64 (sum, sumsq) over ``count`` elements per group.  ``count=None``: sums is the
    65-double message of dove_amd.dist's pair exchange, whose last entry is the element count (read on the device)."""
    L.require_cuda(sums)
    assert sums.dtype == torch.float64 and (sums.shape == (32, 2) if count is not None else sums.numel() == 65)
    stats = torch.empty(32, 2, dtype=torch.float32, device=sums.device)
    L.check(L.load().dove_groupnorm_finalize_sums(L.ptr(sums.contiguous()), 0.0 if count is None else float(count), eps, L.ptr(stats), L.stream_ptr()),
            "dove_groupnorm_finalize_sums")
    return stats


def groupnorm_apply(x, stats, gamma, beta, *, silu=True, yb=None, sshift=0, tmap=None, out=None, nb=1):
    """y = silu?(GN(x) [* Y + B]) with the SpatialNorm3D table yb [Tz,hz,wz,2C] gathered by nearest resize.  ``nb`` > 1: x [nb*T,..],
    stats [nb,32,2], yb [nb*Tz,..], ``tmap`` the frame map of ONE instance."""
    L.require_cuda(x, stats, gamma, beta, yb, out)
    T, H, W, Cc = x.shape
    assert T % nb == 0 and stats.numel() == nb * 64
    T //= nb
    if out is None:
        out = torch.empty_like(x)
    Tz = hz = wz = 0
    tm = None
    if yb is not None:
        assert yb.dtype == torch.bfloat16 and yb.shape[-1] == 2 * Cc and tmap is not None and len(tmap) == T and yb.shape[0] % nb == 0
        Tz, hz, wz = yb.shape[0] // nb, yb.shape[1], yb.shape[2]
        tm = (C.c_int * T)(*tmap)
    L.check(L.load().dove_groupnorm_apply_nb_bf16(L.ptr(x), L.ptr(out), nb, T, H, W, Cc, L.ptr(stats), L.ptr(gamma), L.ptr(beta),
                                                  int(silu), L.ptr(yb), Tz, hz, wz, sshift, tm, L.stream_ptr()),
            "dove_groupnorm_apply_nb_bf16")
    return out


def layernorm_modulate(x, gamma, beta, eps, mod=None, split=0, out=None):
    L.require_cuda(x, gamma, beta, mod, out)
    assert x.dtype == torch.bfloat16 and x.dim() == 2
    if out is None:
        out = torch.empty_like(x)
    if mod is not None:
        assert mod.dtype == torch.float32 and mod.shape == (2, 2, x.shape[1])
    L.check(L.load().dove_layernorm_modulate_bf16(L.ptr(x), L.ptr(out), x.shape[0], x.shape[1], eps, L.ptr(gamma),
                                                  L.ptr(beta), L.ptr(mod), split, L.stream_ptr()),
            "dove_layernorm_modulate_bf16")
    return out


def qkv_post(qkv, N, Npad, heads, text_len, gq, bq, gk, bk, cos, sin, qscale, eps, Qh, Kh, Vt, v_order=1, norm2=None):
    """``v_order=1``: V^T rows in the quad-swapped key order ``attention`` reads (include/dove_hip.h); 0 = natural (pieces that
    are assembled later and converted with ``vt_quad_swap``).  ``norm2`` (fp32 [heads, 2], optional): receives the max squared row
    norms of the stored q / k rows per head - the score bound ``attention`` can use instead of a running maximum."""
    L.require_cuda(qkv, gq, bq, gk, bk, cos, sin, Qh, Kh, Vt, norm2)
    if norm2 is not None:
        assert norm2.dtype == torch.float32 and norm2.shape == (heads, 2) and norm2.is_contiguous()
    L.check(L.load().dove_qkv_post_bf16(L.ptr(qkv), N, Npad, heads, 64, text_len, L.ptr(gq), L.ptr(bq), L.ptr(gk), L.ptr(bk),
                                        L.ptr(cos), L.ptr(sin), qscale, eps, L.ptr(Qh), L.ptr(Kh), L.ptr(Vt), v_order, L.ptr(norm2),
                                        L.stream_ptr()), "dove_qkv_post_bf16")


def vt_quad_swap(Vt):
    """[..., 64, Npad] V^T rows: natural <-> quad-swapped key order, in place."""
    L.require_cuda(Vt)
    assert Vt.is_contiguous()
    L.check(L.load().dove_vt_quad_swap_bf16(L.ptr(Vt), Vt.numel() // Vt.shape[-1], Vt.shape[-1], L.stream_ptr()), "dove_vt_quad_swap_bf16")
    return Vt


def ulysses_place(rq, rk, rv, counts, hloc, N, Npad, Qh, Kh, Vt, norm2_out=None):
    """Receive side of the Ulysses all-to-all (include/dove_hip.h dove_ulysses_place_bf16): per-source-rank blocks of my heads ->
    the attention kernel's head-major operands (V^T quad-swapped, pad columns zero) in one launch.  ``norm2_out`` (fp32 [hloc, 2]): the
    blocks carry one extra row per head and the K blocks' extra rows the ranks' score-bound pairs; their maximum is written here."""
    L.require_cuda(rq, rk, rv, Qh, Kh, Vt, norm2_out)
    if norm2_out is not None:
        assert norm2_out.dtype == torch.float32 and norm2_out.shape == (hloc, 2)
    cnt = (C.c_longlong * len(counts))(*[int(c) for c in counts])
    L.check(L.load().dove_ulysses_place_bf16(L.ptr(rq), L.ptr(rk), L.ptr(rv), cnt, len(counts), hloc, N, Npad, L.ptr(Qh), L.ptr(Kh), L.ptr(Vt),
                                             L.ptr(norm2_out), L.stream_ptr()), "dove_ulysses_place_bf16")


def attention(Qh, Kh, Vt, N, Npad, heads, out, norm2=None):
    """``norm2`` (fp32 [heads, 2], IN/OUT; from the ``qkv_post`` call that produced Qh / Kh - or the element-wise maximum over the ranks
    sharing the rows): heads with finite entries run on the software-pipelined no-shift kernel; a head whose row sums leave that kernel's
    safe window is marked NaN in ``norm2[h, 0]`` and recomputed with the running maximum inside the same call (include/dove_hip.h).  None:
    the running maximum in every head.  ``attention_head_paths(norm2)`` afterwards names the kernel that produced each head."""
    L.require_cuda(Qh, Kh, Vt, out, norm2)
    if norm2 is not None:
        assert norm2.dtype == torch.float32 and norm2.shape == (heads, 2) and norm2.is_contiguous()
    L.check(L.load().dove_attention_fwd_bf16(L.ptr(Qh), L.ptr(Kh), L.ptr(Vt), L.ptr(out), N, Npad, heads, 64, out.shape[1], L.ptr(norm2),
                                             L.stream_ptr()), "dove_attention_fwd_bf16")
    return out


def attention_head_paths(norm2, heads=None):
    """Kernel names, one per head, of the ``attention`` call that was handed ``norm2`` (read AFTER the call; synchronises).  ``norm2=None``
    with ``heads``: a call without bounds."""
    lib = L.load()
    if norm2 is None:
        host, n = None, int(heads)
    else:
        host = norm2.detach().float().cpu().contiguous()
        n = host.shape[0]
    path = (C.c_int * n)()
    L.check(lib.dove_attention_head_paths(None if host is None else C.cast(host.data_ptr(), C.POINTER(C.c_float)), n, path),
            "dove_attention_head_paths")
    return [lib.dove_attention_path_name(int(v)).decode() for v in path]


def qkv_post_mx(qkv, N, Npad, heads, text_len, gq, bq, gk, bk, cos, sin, qscale, eps, Q8, K8, V8t, Vs):
    """dove_qkv_post_mxfp8: e4m3 attention operands (Q8, K8 [heads][Npad][64], V8t [heads][64][Npad] u8; Vs [heads][Npad/64][64][2] u8)."""
    L.require_cuda(qkv, gq, bq, gk, bk, cos, sin, Q8, K8, V8t, Vs)
    L.check(L.load().dove_qkv_post_mxfp8(L.ptr(qkv), N, Npad, heads, 64, text_len, L.ptr(gq), L.ptr(bq), L.ptr(gk), L.ptr(bk),
                                         L.ptr(cos), L.ptr(sin), qscale, eps, L.ptr(Q8), L.ptr(K8), L.ptr(V8t), L.ptr(Vs),
                                         L.stream_ptr()), "dove_qkv_post_mxfp8")


def attention_mx(Q8, K8, V8t, Vs, N, Npad, heads, out):
    L.require_cuda(Q8, K8, V8t, Vs, out)
    L.check(L.load().dove_attention_fwd_mxfp8(L.ptr(Q8), L.ptr(K8), L.ptr(V8t), L.ptr(Vs), L.ptr(out), N, Npad, heads, 64,
                                              out.shape[1], L.stream_ptr()), "dove_attention_fwd_mxfp8")
    return out


def cl_im2col3x3_from_ncthw(x: torch.Tensor, cp: int, scale=1.0, shift=0.0) -> torch.Tensor:
    """[C,T,H,W] -> [T,H,W,cp] bf16 with the 3x3 neighbourhood in the channels ((dy*3+dx)*C + c; zero outside the frame and beyond 9C)."""
    L.require_cuda(x)
    Cc, T, H, W = x.shape
    y = torch.empty(T, H, W, cp, dtype=torch.bfloat16, device=x.device)
    L.check(L.load().dove_cl_im2col3x3_from_ncthw(L.ptr(x), L.dt_code(x), Cc, T, H, W, cp, scale, shift, L.ptr(y), L.stream_ptr()),
            "dove_cl_im2col3x3_from_ncthw")
    return y


def conv_out_gather(p: torch.Tensor, Cc: int, bias, dtype, scale=1.0, shift=0.0, lo=-math.inf, hi=math.inf) -> torch.Tensor:
    """Second half of the tap-split decoder.conv_out (include/dove_hip.h dove_conv_out_gather): p [T,H,W,ld] fp32 partial planes ->
    [Cc,T,H,W] ``dtype`` with the range map of ``ncthw_from_cl``."""
    L.require_cuda(p, bias)
    assert p.dtype == torch.float32 and p.dim() == 4
    T, H, W, ld = p.shape
    y = torch.empty(Cc, T, H, W, dtype=dtype, device=p.device)
    L.check(L.load().dove_conv_out_gather(L.ptr(p), ld, T, H, W, Cc, L.ptr(bias), scale, shift, lo, hi, L.ptr(y), L.dt_code(y),
                                          L.stream_ptr()), "dove_conv_out_gather")
    return y


def conv_out_gather_cl(p: torch.Tensor, Cc: int, bias, ld: int = 8) -> torch.Tensor:
    """The same gather for a spatial TILE (dove_conv_out_gather_cl): p [T,H,W,ldp] fp32 partial planes -> [T,H,W,ld] bf16 channels-last
    (channels >= Cc zero, no range map): tiles are cross-faded channels-last before the clip changes layout.  T may be nb x frames."""
    L.require_cuda(p, bias)
    assert p.dtype == torch.float32 and p.dim() == 4
    T, H, W, ldp = p.shape
    y = torch.empty(T, H, W, ld, dtype=torch.bfloat16, device=p.device)
    L.check(L.load().dove_conv_out_gather_cl(L.ptr(p), ldp, T, H, W, Cc, L.ptr(bias), L.ptr(y), ld, L.stream_ptr()), "dove_conv_out_gather_cl")
    return y


def tile_gather(x: torch.Tensor, t0: int, nt: int, th: int, tw: int, origins, im2col_cin: int = 0) -> torch.Tensor:
    """Same-shaped spatial tiles of a channels-last clip x [T,H,W,C] -> one tile-major batch [nb*nt, th, tw, C] (dove_tile_gather_bf16):
    frames [t0, t0+nt), tile n at origins[n] = (oy, ox).  ``im2col_cin`` > 0: x is the im2col'ed clip (cl_im2col3x3_from_ncthw) and the
    channels of taps that reach outside a tile are zeroed at its border (diffusers' tiles see zero padding at their own border)."""
    import ctypes as C
    L.require_cuda(x)
    assert x.dtype == torch.bfloat16 and x.dim() == 4
    T, H, W, Cc = x.shape
    nb = len(origins)
    oy = (C.c_int * nb)(*[int(o[0]) for o in origins])
    ox = (C.c_int * nb)(*[int(o[1]) for o in origins])
    assert 0 <= t0 and t0 + nt <= T
    y = torch.empty(nb * nt, th, tw, Cc, dtype=torch.bfloat16, device=x.device)
    L.check(L.load().dove_tile_gather_bf16(L.ptr(x), H, W, Cc, t0, nt, th, tw, nb, oy, ox, im2col_cin, L.ptr(y), L.stream_ptr()),
            "dove_tile_gather_bf16")
    return y


def cl_from_ncthw(x: torch.Tensor, cp: int, scale=1.0, shift=0.0) -> torch.Tensor:
    """[C,T,H,W] fp32/bf16 -> [T,H,W,cp] bf16 (zero-padded channels)."""
    L.require_cuda(x)
    Cc, T, H, W = x.shape
    y = torch.empty(T, H, W, cp, dtype=torch.bfloat16, device=x.device)
    L.check(L.load().dove_cl_from_ncthw(L.ptr(x), L.dt_code(x), Cc, T * H * W, cp, scale, shift, L.ptr(y), L.stream_ptr()),
            "dove_cl_from_ncthw")
    return y


def ncthw_from_cl(x: torch.Tensor, Cc: int, dtype, scale=1.0, shift=0.0, lo=-math.inf, hi=math.inf) -> torch.Tensor:
    """[T,H,W,ld] bf16 -> [Cc,T,H,W] dtype, y = clamp(x*scale+shift, lo, hi)."""
    L.require_cuda(x)
    T, H, W, ld = x.shape
    y = torch.empty(Cc, T, H, W, dtype=dtype, device=x.device)
    L.check(L.load().dove_ncthw_from_cl(L.ptr(x), ld, Cc, T * H * W, scale, shift, lo, hi, L.ptr(y), L.dt_code(y),
                                        L.stream_ptr()), "dove_ncthw_from_cl")
    return y


def avgpool_time(x: torch.Tensor, nb: int = 1) -> torch.Tensor:
    """Downsample3D's temporal pool; ``nb`` > 1: x is nb instances [nb*T, H, W, C], pooled per instance."""
    L.require_cuda(x)
    T, H, W, Cc = x.shape
    assert T % nb == 0
    T //= nb
    if T == 1:
        return x
    To = 1 + (T - 1) // 2 if T % 2 else T // 2
    y = torch.empty(nb * To, H, W, Cc, dtype=torch.bfloat16, device=x.device)
    L.check(L.load().dove_avgpool_time_nb_bf16(L.ptr(x), nb, T, H * W * Cc, L.ptr(y), L.stream_ptr()), "dove_avgpool_time_nb_bf16")
    return y


def posterior_sample(moments_cl: torch.Tensor, latent_channels: int, noise: torch.Tensor, dtype) -> torch.Tensor:
    """moments [T,h,w,>=2L] bf16 + noise [L,T,h,w] -> sample [L,T,h,w] dtype."""
    L.require_cuda(moments_cl, noise)
    T, h, w, ld = moments_cl.shape
    out = torch.empty(latent_channels, T, h, w, dtype=dtype, device=noise.device)
    L.check(L.load().dove_posterior_sample(L.ptr(moments_cl), ld, latent_channels, T * h * w, L.ptr(noise), L.dt_code(noise),
                                           L.ptr(out), L.dt_code(out), L.stream_ptr()), "dove_posterior_sample")
    return out


def axpby(x, y, a: float, b: float):
    L.require_cuda(x, y)
    assert x.shape == y.shape and x.dtype == y.dtype
    out = torch.empty_like(x)
    L.check(L.load().dove_axpby(L.ptr(x), L.ptr(y), L.ptr(out), L.dt_code(x), x.numel(), a, b, L.stream_ptr()), "dove_axpby")
    return out


def patchify(x: torch.Tensor, pt: int, p: int, ld: int) -> torch.Tensor:
    """[T,C,h,w] -> tokens [Nv, ld] bf16 (features beyond C*pt*p*p zero)."""
    L.require_cuda(x)
    T, Cc, H, W = x.shape
    nv = (T // pt) * (H // p) * (W // p)
    feat = Cc * pt * p * p
    tok = (torch.zeros if ld > feat else torch.empty)(nv, ld, dtype=torch.bfloat16, device=x.device)
    L.check(L.load().dove_patchify(L.ptr(x), L.dt_code(x), T, Cc, H, W, pt, p, L.ptr(tok), ld, L.stream_ptr()), "dove_patchify")
    return tok


def unpatchify(tok: torch.Tensor, T, Cc, H, W, pt, p, dtype) -> torch.Tensor:
    L.require_cuda(tok)
    y = torch.empty(T, Cc, H, W, dtype=dtype, device=tok.device)
    L.check(L.load().dove_unpatchify(L.ptr(tok), tok.shape[1], T, Cc, H, W, pt, p, L.ptr(y), L.dt_code(y), L.stream_ptr()),
            "dove_unpatchify")
    return y


def gemv(W: torch.Tensor, bias, x: torch.Tensor, act_in: int = 0) -> torch.Tensor:
    """y = W @ act(x) + b for one vector; W [out,in] bf16, x fp32 [in] -> fp32 [out]."""
    L.require_cuda(W, bias, x)
    assert W.dtype == torch.bfloat16 and x.dtype == torch.float32 and x.numel() == W.shape[1]
    y = torch.empty(W.shape[0], dtype=torch.float32, device=x.device)
    L.check(L.load().dove_gemv_bf16(L.ptr(W), L.ptr(bias), L.ptr(x), W.shape[1], W.shape[0], act_in, L.ptr(y), L.stream_ptr()),
            "dove_gemv_bf16")
    return y


def blend_edge(a: torch.Tensor, b: torch.Tensor, extent: int, axis: int) -> torch.Tensor:
    """diffusers blend_v (axis 0) / blend_h (axis 1) on channels-last tiles [T,H,W,ld], in place on b."""
    L.require_cuda(a, b)
    assert a.dtype == b.dtype == torch.bfloat16 and a.shape[0] == b.shape[0] and a.shape[3] == b.shape[3]
    L.check(L.load().dove_blend_edge_bf16(L.ptr(a), L.ptr(b), b.shape[0], a.shape[1], a.shape[2], b.shape[1], b.shape[2],
                                          b.shape[3], extent, axis, L.stream_ptr()), "dove_blend_edge_bf16")
    if getattr(b, "gn_stats", None) is not None:             # b changed in place: statistics its producing conv attached are stale
        b.gn_stats = None
        b.gn_rows = None
    return b


def preprocess_u8(frames: torch.Tensor, pad_f: int, pad_h: int, pad_w: int, upscale: int, dtype) -> torch.Tensor:
    """frames [F,H,W,3] uint8 -> [3, F+pad_f, (H+pad_h)*up, (W+pad_w)*up] dtype in [-1,1]."""
    L.require_cuda(frames)
    assert frames.dtype == torch.uint8 and frames.dim() == 4 and frames.shape[3] == 3
    F0, H0, W0, _ = frames.shape
    out = torch.empty(3, F0 + pad_f, (H0 + pad_h) * upscale, (W0 + pad_w) * upscale, dtype=dtype, device=frames.device)
    L.check(L.load().dove_preprocess_u8(L.ptr(frames), F0, H0, W0, pad_f, pad_h, pad_w, upscale, L.ptr(out), L.dt_code(out),
                                        L.stream_ptr()), "dove_preprocess_u8")
    return out


def postprocess_u8(video: torch.Tensor, Fo: int, Ho: int, Wo: int) -> torch.Tensor:
    """video [3,F,H,W] in [0,1] -> [Fo,Ho,Wo,3] uint8 (crop + x255 + truncate)."""
    L.require_cuda(video)
    _, F, H, W = video.shape
    out = torch.empty(Fo, Ho, Wo, 3, dtype=torch.uint8, device=video.device)
    L.check(L.load().dove_postprocess_u8(L.ptr(video), L.dt_code(video), F, H, W, Fo, Ho, Wo, L.ptr(out), L.stream_ptr()),
            "dove_postprocess_u8")
    return out


# ---- MXFP8 linears (BASELINE configs[4]; csrc/mxfp8.hip) ------------------------------------------------------------------
@dataclass
class PackedMx:
    """One operand in OCP MXFP8: q [rows][K] e4m3fn bytes, s [K/256][rows][2] words of four E8M0 block scales each."""
    q: torch.Tensor
    s: torch.Tensor
    rows: int
    K: int
    bias: torch.Tensor | None = None


def mx_quant(x: torch.Tensor) -> PackedMx:
    """x [rows, K] bf16 (K % 256 == 0) -> MXFP8 (per 32-element block: scale 2^ceil(log2(amax/448)), RNE to e4m3fn)."""
    L.require_cuda(x)
    assert x.dtype == torch.bfloat16 and x.dim() == 2
    rows, K = x.shape
    q = torch.empty(rows, K, dtype=torch.uint8, device=x.device)
    s = torch.empty(K // 256, rows, 2, dtype=torch.int32, device=x.device)
    L.check(L.load().dove_mx_quant_bf16(L.ptr(x), rows, K, L.ptr(q), L.ptr(s), L.stream_ptr()), "dove_mx_quant_bf16")
    return PackedMx(q, s, rows, K)


def pack_linear_mx(weight: torch.Tensor, bias: torch.Tensor | None, device) -> PackedMx:
    """nn.Linear weight [N, K] -> MXFP8 operand (quantised on the GPU from its bf16 rounding, like the bf16 path's weights)."""
    w = weight.detach().to(device=device, dtype=torch.bfloat16).contiguous()
    pm = mx_quant(w)
    if bias is not None:
        pm.bias = bias.detach().to(device=device, dtype=torch.float32).contiguous()
    return pm


def linear_mx(x: PackedMx, w: PackedMx, *, resid: torch.Tensor | None = None, gate: torch.Tensor | None = None, gate_split: int = 0,
              act: int = 0, out: torch.Tensor | None = None) -> torch.Tensor:
    """out [M, N] bf16 = epilogue(dequant(x) @ dequant(w)^T + bias); same epilogue contract as ``linear``."""
    M, N, K = x.rows, w.rows, w.K
    assert x.K == K
    L.require_cuda(x.q, x.s, w.q, w.s, resid, gate, out)
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=x.q.device)
    assert out.dtype == torch.bfloat16 and out.shape[0] == M
    if gate is not None:
        assert gate.dtype == torch.float32 and gate.shape == (2, N)
    L.check(L.load().dove_linear_mxfp8(L.ptr(x.q), L.ptr(x.s), L.ptr(w.q), L.ptr(w.s), L.ptr(w.bias), L.ptr(resid), L.ptr(gate), L.ptr(out),
                                       M, N, K, out.shape[1], resid.shape[1] if resid is not None else 0, gate_split, act,
                                       L.stream_ptr()), "dove_linear_mxfp8")
    return out


# ---- T5 text-encoder operators (csrc/t5.hip) --------------------------------------------------------------------------------
def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    L.require_cuda(x, weight)
    assert x.dtype == torch.bfloat16 and x.dim() == 2 and weight.dtype == torch.float32
    y = torch.empty_like(x)
    L.check(L.load().dove_rmsnorm_bf16(L.ptr(x), L.ptr(y), x.shape[0], x.shape[1], eps, L.ptr(weight), L.stream_ptr()), "dove_rmsnorm_bf16")
    return y


def gated_gelu(x: torch.Tensor) -> torch.Tensor:
    """x [M, 2F] (wi_0 x || wi_1 x) -> gelu_new(x[:, :F]) * x[:, F:]."""
    L.require_cuda(x)
    assert x.dtype == torch.bfloat16 and x.dim() == 2 and x.shape[1] % 16 == 0
    y = torch.empty(x.shape[0], x.shape[1] // 2, dtype=torch.bfloat16, device=x.device)
    L.check(L.load().dove_gated_gelu_bf16(L.ptr(x), L.ptr(y), x.shape[0], x.shape[1] // 2, L.stream_ptr()), "dove_gated_gelu_bf16")
    return y


def attention_bias(qkv: torch.Tensor, bias: torch.Tensor, heads: int) -> torch.Tensor:
    """T5 self-attention on the fused projection qkv [N, 3*heads*64] with additive fp32 bias [heads, N, N] -> [N, heads*64]."""
    L.require_cuda(qkv, bias)
    N, D = qkv.shape[0], heads * 64
    assert qkv.dtype == torch.bfloat16 and qkv.shape[1] == 3 * D and bias.dtype == torch.float32 and bias.shape == (heads, N, N)
    out = torch.empty(N, D, dtype=torch.bfloat16, device=qkv.device)
    base = qkv.data_ptr()
    L.check(L.load().dove_attention_bias_bf16(C.c_void_p(base), C.c_void_p(base + 2 * D), C.c_void_p(base + 4 * D), 3 * D, L.ptr(bias),
                                              L.ptr(out), D, N, heads, 64, L.stream_ptr()), "dove_attention_bias_bf16")
    return out
