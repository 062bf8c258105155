"""T5 text encoder (ref :429-444): the host graph of dove_amd.t5 against the REAL reference implementation
of this component - transformers' T5EncoderModel, importable here - on random-init weights.  This row's oracle is therefore
PINNED (unlike the diffusers arithmetic): relative-position buckets, un-scaled attention, shared position bias, gated GELU,
T5LayerNorm and the residual order are all checked against transformers itself.  Operators are the torch emulation here; the
same graph runs on the HIP operators in tests/test_t5_gpu.py."""
import pytest
import torch

import emu_ops
from dove_amd import t5 as dt5

transformers = pytest.importorskip("transformers")


def hf_model(cfg):
    from transformers import T5Config, T5EncoderModel
    c = T5Config(vocab_size=cfg["vocab_size"], d_model=cfg["d_model"], d_kv=cfg["d_kv"], d_ff=cfg["d_ff"], num_layers=cfg["num_layers"],
                 num_heads=cfg["num_heads"], relative_attention_num_buckets=cfg["relative_attention_num_buckets"],
                 relative_attention_max_distance=cfg["relative_attention_max_distance"], feed_forward_proj="gated-gelu",
                 dense_act_fn="gelu_new", layer_norm_epsilon=cfg["layer_norm_epsilon"], dropout_rate=0.0)
    torch.manual_seed(0)
    m = T5EncoderModel(c).eval()
    with torch.no_grad():                      # HF inits the norms to 1 and the bias table small: make every parameter matter
        for n, p in m.named_parameters():
            if "layer_norm" in n:
                p.add_(0.1 * torch.randn_like(p))
            if "relative_attention_bias" in n:
                p.mul_(20.0)
    return m


SMALL = dict(dt5.T5_XXL_CONFIG, vocab_size=200, d_model=256, d_ff=512, num_layers=3, num_heads=4)


def test_relative_position_buckets_match_transformers():
    from transformers.models.t5.modeling_t5 import T5Attention
    for n in (5, 226, 300):
        rel = torch.arange(n)[None, :] - torch.arange(n)[:, None]
        want = T5Attention._relative_position_bucket(rel, bidirectional=True, num_buckets=32, max_distance=128)
        assert torch.equal(dt5.relative_position_buckets(n), want)


def test_t5_host_graph_vs_transformers(monkeypatch):
    emu_ops.install(monkeypatch)
    m = hf_model(SMALL)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    assert set(dt5.t5_param_shapes(SMALL)) <= set(sd), "state-dict names differ from transformers' T5EncoderModel"
    enc = dt5.T5EncoderModel(SMALL, sd, "cpu")
    ids = torch.randint(0, SMALL["vocab_size"], (2, 226), generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        want = m(input_ids=ids)[0]
    got = enc(ids)[0].float()
    err = float((got - want).abs().max() / want.abs().max())
    rms = float((got - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt())
    print(f"T5 host graph vs transformers fp32: rel-max {err:.4f} rms-rel {rms:.2e}")
    assert got.shape == want.shape == (2, 226, 256)
    assert err < 0.04 and rms < 1.2e-2
    with pytest.raises(NotImplementedError):
        enc(ids, attention_mask=torch.ones_like(ids))
