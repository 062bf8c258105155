"""CPU: the algebra behind the two pack-time weight sums of dove_amd.ops.pack_conv (dove_conv_desc.w_first / .w_sub, include/dove_hip.h), checked
in fp64 against plain torch on the formulations the reference computes - independent of any kernel:

  * CogVideoXCausalConv3d without a conv cache pads the front with the replicated first frame: output frame 0 = (w0 + w1 + w2) * x0 and
    output frame 1 = (w0 + w1) * x0 + w2 * x1 (one / two temporal taps instead of three);
  * CogVideoXUpsample3D = F.interpolate(scale_factor=2, mode="nearest") followed by a 3x3 Conv2d(padding=1): per output phase (oy & 1, ox & 1) a
    2x2 conv on the LOW-RES input with the weights summed over the taps that read the same low-res pixel (4 / 9 of the MACs).

  * the first causal conv behind CogVideoXUpsample3D's TIME doubling reads frames that come in equal pairs: two of every frame's three taps
    see the same frame, so (w0 + w1) x[t-1] + w2 x[t] or w0 x[t-2] + (w1 + w2) x[t] by frame parity (dove_conv_desc.tdup / w_pair).

The packed tensors hold bf16-rounded sums; here the same index mapping is applied to exact (fp64) weights, so any mismatch is an indexing error."""
import torch
import torch.nn.functional as F

from dove_amd import ops


def _unpacked(pc, t):
    """[taps][cout_pad][cin_pad] bf16 -> [cout][cin][taps] fp64"""
    return t.double().permute(1, 2, 0)[: pc.cout, : pc.cin]


def test_first_frame_temporal_sums_identity():
    g = torch.Generator().manual_seed(1)
    cout, cin, T, H, W = 32, 32, 4, 6, 7
    # weights that are exact in bf16, so the packed sums are exact too and the identity can be checked to fp64 rounding
    w = torch.randint(-8, 9, (cout, cin, 3, 3, 3), generator=g).double() / 16
    x = torch.randn(1, cin, T, H, W, generator=g, dtype=torch.float64)
    pc = ops.pack_conv(w.float(), None, "cpu")
    assert pc.w_first is not None and pc.w_first.shape == (2, 9, 32, 32)
    s01 = _unpacked(pc, pc.w_first[0]).reshape(cout, cin, 3, 3)          # w0 + w1
    s012 = _unpacked(pc, pc.w_first[1]).reshape(cout, cin, 3, 3)         # w0 + w1 + w2
    assert torch.equal(s01, w[:, :, 0] + w[:, :, 1]) and torch.equal(s012, w.sum(2))
    # the reference: replicate frame 0 twice in front, 3x3x3 conv, zero spatial padding
    xp = torch.cat([x[:, :, :1]] * 2 + [x], dim=2)
    ref = F.conv3d(F.pad(xp, (1, 1, 1, 1)), w)
    f0 = F.conv2d(x[:, :, 0], s012, padding=1)
    f1 = F.conv2d(x[:, :, 0], s01, padding=1) + F.conv2d(x[:, :, 1], w[:, :, 2], padding=1)
    assert torch.allclose(ref[:, :, 0], f0, atol=1e-12) and torch.allclose(ref[:, :, 1], f1, atol=1e-12)


def test_subpixel_upsample_conv_identity():
    g = torch.Generator().manual_seed(2)
    cout, cin, H, W = 32, 32, 5, 9
    w = torch.randint(-8, 9, (cout, cin, 3, 3), generator=g).double() / 16
    x = torch.randn(1, cin, H, W, generator=g, dtype=torch.float64)
    pc = ops.pack_conv(w.float(), None, "cpu")
    assert pc.w_sub is not None and pc.w_sub.shape == (4, 4, 32, 32)
    ref = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, padding=1)          # what Upsample3D computes per frame
    out = torch.zeros_like(ref)
    xp = F.pad(x, (1, 1, 1, 1))                                                              # low-res halo: origin (y - 1, x - 1)
    for py in range(2):
        for px in range(2):
            acc = torch.zeros(1, cout, H, W, dtype=torch.float64)
            for a in range(2):
                for b in range(2):
                    wt = pc.w_sub[2 * py + px, 2 * a + b].double()[:cout, :cin]              # [cout][cin]
                    acc += torch.einsum("oc,nchw->nohw", wt, xp[:, :, py + a: py + a + H, px + b: px + b + W])
            out[:, :, py::2, px::2] = acc
    assert torch.allclose(out, ref, atol=1e-12), float((out - ref).abs().max())


def test_sums_are_rounded_once():
    """With generic weights the packed sums are bf16(fp32 sum of the bf16 weights): one rounding, not a sum of rounded partial sums."""
    g = torch.Generator().manual_seed(3)
    w = torch.randn(32, 32, 3, 3, 3, generator=g)
    pc = ops.pack_conv(w, None, "cpu")
    wb = w.to(torch.bfloat16).float()
    want = ((wb[:, :, 0] + wb[:, :, 1]) + wb[:, :, 2]).to(torch.bfloat16)                    # [cout][cin][3][3]
    got = pc.w_first[1].permute(1, 2, 0).reshape(32, 32, 3, 3)
    assert torch.equal(got, want)


def _doubled(T_src, first_single, g, cin, H, W):
    """Upsample3D's time doubling of T_src random frames: pairs, the first frame single when ``first_single``."""
    src = torch.randn(1, cin, T_src, H, W, generator=g, dtype=torch.float64)
    idx = ([0] + [1 + i // 2 for i in range(2 * (T_src - 1))]) if first_single else [i // 2 for i in range(2 * T_src)]
    return src[:, :, idx]


def test_frame_pair_sums_identity():
    """ops.temporal_split (the host restatement of the kernel's group rule, csrc/igemm.hip h4_split) applied with EXACT weights reproduces
    F.conv3d on time-doubled input for every frame of: an odd batch without a cache (tdup 2: frame 0 single), an even batch with a cache
    that is a pair (tdup 1), an even batch without a cache (tdup 1, replicated front), and the packed w_pair holds w0 + w1 / w1 + w2."""
    g = torch.Generator().manual_seed(4)
    cout, cin, H, W = 32, 32, 5, 6
    w = torch.randint(-8, 9, (cout, cin, 3, 3, 3), generator=g).double() / 16
    pc = ops.pack_conv(w.float(), None, "cpu", pair=True)
    assert pc.w_pair is not None and pc.w_pair.shape == (2, 9, 32, 32)
    assert torch.equal(_unpacked(pc, pc.w_pair[0]).reshape(cout, cin, 3, 3), w[:, :, 0] + w[:, :, 1])
    assert torch.equal(_unpacked(pc, pc.w_pair[1]).reshape(cout, cin, 3, 3), w[:, :, 1] + w[:, :, 2])
    assert ops.pack_conv(w.float(), None, "cpu").w_pair is None
    W_ = {"w0": w[:, :, 0], "w1": w[:, :, 1], "w2": w[:, :, 2], "w01": w[:, :, 0] + w[:, :, 1], "w12": w[:, :, 1] + w[:, :, 2], "w012": w.sum(2)}
    for tdup, cached, x in ((2, False, _doubled(3, True, g, cin, H, W)),          # 1 + 4 frames, the head of a clip
                            (1, True, _doubled(2, False, g, cin, H, W)),           # 4 frames behind a cache
                            (1, False, _doubled(2, False, g, cin, H, W))):         # an even first batch
        T = x.shape[2]
        cache = _doubled(1, False, g, cin, H, W) if cached else None              # an equal pair
        front = cache if cached else torch.cat([x[:, :, :1]] * 2, dim=2)
        ref = F.conv3d(F.pad(torch.cat([front, x], dim=2), (1, 1, 1, 1)), w)

        def frame(i):
            return x[:, :, i] if i >= 0 else (cache[:, :, 2 + i] if cached else x[:, :, 0])
        n_groups = 0
        for t in range(T):
            groups = ops.temporal_split(t, tdup, cached)
            n_groups += len(groups)
            got = sum(F.conv2d(frame(src), W_[wk], padding=1) for wk, src in groups)
            assert torch.allclose(got, ref[:, :, t], atol=1e-12), (tdup, cached, t, groups)
        assert n_groups == sum(ops.temporal_groups(t, tdup, cached) for t in range(T)) <= 2 * T
