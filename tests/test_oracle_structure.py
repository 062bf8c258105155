"""Structural pins of the clean-room oracle (the reference ships no tests): published MAC and
parameter counts, frame-batch rules, conv-cache semantics, shape conservation."""
import torch

from dove_amd import config, flops, weights
from oracle import dit as odit
from oracle.vae import OracleVAE, frame_batches


def test_published_mac_count():
    """/root/reference/assets/Quantitative-2.png: DOVE = 504.81 T MACs.  Our count for a 33x720x1280 pipeline tensor
    with the SDPA matmuls excluded (how profilers count) is 504.60 T (-0.04 %)."""
    v, t, _ = config.default_configs()
    m = flops.clip_macs(v, t, 33, 720, 1280)
    no_attn = m["total"] - m["attention"]
    assert abs(no_attn / 1e12 - 504.60) < 0.01
    assert abs(no_attn / 504.81e12 - 1) < 1e-3
    assert abs(m["flop"] / 1.180641e15 - 1) < 1e-6          # 35.777 TFLOP per output frame
    assert m["tokens"] == 18226
    assert abs(flops.clip_macs(v, t, 33, 768, 1280)["flop"] / 1.271096e15 - 1) < 1e-5
    assert abs(flops.clip_macs(v, t, 9, 256, 256)["flop"] / 0.0239e15 - 1) < 2e-2


def test_published_param_count():
    """Same image: 5 787.19 M parameters (T5 excluded).  VAE 215.58 M + DiT 5 570.68 M = 5 786.26 M; the remaining
    0.93 M equals the [226,4096] empty-prompt embedding the DOVE trainer keeps alongside the model."""
    v, t, _ = config.default_configs()
    nv = weights.count_params(weights.vae_param_shapes(v))
    nt = weights.count_params(weights.dit_param_shapes(t))
    assert abs((nv + nt) / 1e6 - 5786.26) < 0.01
    assert abs((nv + nt + 226 * 4096) / 1e6 - 5787.19) < 0.01


def test_frame_batch_rules():
    assert frame_batches(33, 8) == [(0, 9), (9, 17), (17, 25), (25, 33)]
    assert frame_batches(9, 2) == [(0, 3), (3, 5), (5, 7), (7, 9)]
    assert frame_batches(24, 8) == [(0, 8), (8, 16), (16, 24)]
    assert frame_batches(129, 8)[0] == (0, 9) and len(frame_batches(129, 8)) == 16
    assert frame_batches(1, 8) == [(0, 1)] and frame_batches(5, 8) == [(0, 5)]
    from dove_amd.vae import frame_batches as fb2
    for n in (1, 5, 9, 17, 24, 33, 54, 129):
        assert fb2(n, 8) == frame_batches(n, 8) and fb2(n, 2) == frame_batches(n, 2)


def test_conv_cache_equals_single_shot_causal_conv():
    """A CausalConv3d run batch-by-batch with conv_cache equals one causal conv over the whole clip (App. A.1)."""
    v, _, _ = config.tiny_configs()
    w = weights.random_state_dict(weights.vae_param_shapes(v), 3)
    vae = OracleVAE(v, w)
    x = torch.randn(1, 3, 17, 8, 8)
    whole = vae.causal_conv(x, "encoder.conv_in", {})
    cache, parts = {}, []
    for s, e in frame_batches(17, 8):
        parts.append(vae.causal_conv(x[:, :, s:e], "encoder.conv_in", cache))
    assert torch.allclose(torch.cat(parts, 2), whole, atol=1e-5)


def test_shape_conservation():
    v, t, s = config.tiny_configs()
    vae = OracleVAE(v, weights.random_state_dict(weights.vae_param_shapes(v), 3))
    for F, T in ((1, 1), (9, 3), (17, 5), (33, 9)):
        p = vae.encode(torch.zeros(1, 3, F, 16, 16))
        assert p.shape == (1, 32, T, 2, 2)
        assert vae.decode(torch.zeros(1, 16, T, 2, 2)).shape == (1, 3, F, 16, 16)


def test_rope_and_unpatchify_are_inverse_consistent():
    c, s = odit.rope_3d(64, 5, 45, 80)
    assert c.shape == (18000, 64) and torch.all(c[:, 0::2] == c[:, 1::2]) and torch.allclose(c * c + s * s, torch.ones_like(c), atol=1e-5)
    # position (t=0,h=0,w=0) is the identity rotation
    assert torch.all(c[0] == 1) and torch.all(s[0] == 0)
