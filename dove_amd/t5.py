"""T5 text encoder facade (CogVideoX's ``text_encoder``: transformers' T5EncoderModel, T5-v1.1-XXL) over the HIP operators.

The reference touches it only for a non-empty prompt (/root/reference/inference_script.py:429-444):

    ids = pipe.tokenizer(prompt, padding="max_length", max_length=226, truncation=True, add_special_tokens=True,
                         return_tensors="pt").input_ids
    prompt_embedding = pipe.text_encoder(ids.to(device))[0]            # [B, 226, 4096], NO attention mask

so the surface is ``__call__(input_ids) -> (last_hidden_state,)``.  Arithmetic (transformers T5Stack, encoder only):
embedding lookup; per block  h += o(softmax(q k^T + position_bias) v)  on T5LayerNorm(h) [no 1/sqrt(d) scaling, no biases,
relative-position bias from block 0 shared by all blocks],  h += wo(gelu_new(wi_0 n) * (wi_1 n))  on T5LayerNorm(h);
final T5LayerNorm.  Weights: ``text_encoder/`` of the checkpoint (sharded safetensors, state-dict names of
transformers' T5EncoderModel).  The tokenizer is host-side text processing and stays transformers' own (the reference
uses the same class); see dove_amd.pipeline.
"""
from __future__ import annotations

import json
import math
import os

import torch

from . import ops
from .config import AttrDict

T5_XXL_CONFIG = {      # google/t5-v1_1-xxl encoder, as shipped in CogVideoX1.5-5B's text_encoder/config.json
    "vocab_size": 32128, "d_model": 4096, "d_kv": 64, "d_ff": 10240, "num_layers": 24, "num_heads": 64,
    "relative_attention_num_buckets": 32, "relative_attention_max_distance": 128, "layer_norm_epsilon": 1e-6,
    "feed_forward_proj": "gated-gelu", "dense_act_fn": "gelu_new",
}


def t5_param_shapes(cfg: dict):
    from collections import OrderedDict
    d, dk, ff, H = cfg["d_model"], cfg["d_kv"], cfg["d_ff"], cfg["num_heads"]
    s = OrderedDict()
    s["shared.weight"] = (cfg["vocab_size"], d)
    for i in range(cfg["num_layers"]):
        b = f"encoder.block.{i}."
        for n in ("q", "k", "v"):
            s[b + f"layer.0.SelfAttention.{n}.weight"] = (H * dk, d)
        s[b + "layer.0.SelfAttention.o.weight"] = (d, H * dk)
        if i == 0:
            s[b + "layer.0.SelfAttention.relative_attention_bias.weight"] = (cfg["relative_attention_num_buckets"], H)
        s[b + "layer.0.layer_norm.weight"] = (d,)
        s[b + "layer.1.DenseReluDense.wi_0.weight"] = (ff, d)
        s[b + "layer.1.DenseReluDense.wi_1.weight"] = (ff, d)
        s[b + "layer.1.DenseReluDense.wo.weight"] = (d, ff)
        s[b + "layer.1.layer_norm.weight"] = (d,)
    s["encoder.final_layer_norm.weight"] = (d,)
    return s


def relative_position_buckets(n: int, num_buckets: int = 32, max_distance: int = 128) -> torch.Tensor:
    """Bucket index [n, n] of (key position - query position) for a bidirectional T5 encoder: half the buckets per sign,
    of each half the first ``max_exact`` are exact offsets and the rest logarithmic up to ``max_distance``."""
    rel = torch.arange(n)[None, :] - torch.arange(n)[:, None]           # memory - context
    half = num_buckets // 2
    out = (rel > 0).long() * half
    dist = rel.abs()
    max_exact = half // 2
    large = max_exact + (torch.log(dist.float().clamp_min(1) / max_exact) / math.log(max_distance / max_exact) * (half - max_exact)).long()
    large = large.clamp_max(half - 1)
    return out + torch.where(dist < max_exact, dist, large)


class T5EncoderModel:
    def __init__(self, config: dict, state_dict, device="cuda", dtype=torch.bfloat16):
        self.config = AttrDict(config)
        self.device, self.dtype = torch.device(device), dtype
        c = self.config
        if c["d_kv"] != 64:
            raise NotImplementedError("HIP attention kernels are built for head_dim 64")
        if c.get("feed_forward_proj", "gated-gelu") != "gated-gelu":
            raise NotImplementedError("only the gated-GELU T5 v1.1 feed-forward is implemented")
        self.H, self.d, self.eps = c["num_heads"], c["d_model"], c.get("layer_norm_epsilon", 1e-6)
        self._bias_cache = {}
        self._pack(state_dict)

    def _pack(self, sd):
        dev = self.device
        f32 = lambda k: sd[k].to(dev, torch.float32).contiguous()   # noqa: E731
        emb = sd["shared.weight"] if "shared.weight" in sd else sd["encoder.embed_tokens.weight"]
        self.embed = emb.to(dev, torch.bfloat16).contiguous()
        self.blocks = []
        for i in range(self.config["num_layers"]):
            b = f"encoder.block.{i}."
            wqkv = torch.cat([sd[b + f"layer.0.SelfAttention.{n}.weight"] for n in ("q", "k", "v")], dim=0)
            wi = torch.cat([sd[b + "layer.1.DenseReluDense.wi_0.weight"], sd[b + "layer.1.DenseReluDense.wi_1.weight"]], dim=0)
            self.blocks.append(dict(
                ln0=f32(b + "layer.0.layer_norm.weight"), qkv=ops.pack_conv(wqkv, None, dev),
                o=ops.pack_conv(sd[b + "layer.0.SelfAttention.o.weight"], None, dev),
                ln1=f32(b + "layer.1.layer_norm.weight"), wi=ops.pack_conv(wi, None, dev),
                wo=ops.pack_conv(sd[b + "layer.1.DenseReluDense.wo.weight"], None, dev)))
        self.rel_bias = f32("encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight")     # [buckets, H]
        self.final_ln = f32("encoder.final_layer_norm.weight")

    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    def _position_bias(self, n: int) -> torch.Tensor:
        if n not in self._bias_cache:
            c = self.config
            bk = relative_position_buckets(n, c.get("relative_attention_num_buckets", 32), c.get("relative_attention_max_distance", 128))
            self._bias_cache = {n: self.rel_bias[bk.to(self.device)].permute(2, 0, 1).contiguous()}      # [H, n, n] fp32
        return self._bias_cache[n]

    @torch.no_grad()
    def __call__(self, input_ids, attention_mask=None, **kw):
        if attention_mask is not None:
            raise NotImplementedError("the reference calls the text encoder without an attention mask (ref :438)")
        ids = input_ids.to(self.device)
        outs = []
        for b in range(ids.shape[0]):
            h = self.embed[ids[b]].contiguous()                                  # [N, d] bf16
            bias = self._position_bias(h.shape[0])
            for blk in self.blocks:
                n0 = ops.rmsnorm(h, blk["ln0"], self.eps)
                att = ops.attention_bias(ops.linear(n0, blk["qkv"]), bias, self.H)
                ops.linear(att, blk["o"], resid=h, out=h)
                n1 = ops.rmsnorm(h, blk["ln1"], self.eps)
                ops.linear(ops.gated_gelu(ops.linear(n1, blk["wi"])), blk["wo"], resid=h, out=h)
            outs.append(ops.rmsnorm(h, self.final_ln, self.eps))
        return (torch.stack(outs),)

    forward = __call__

    @classmethod
    def from_pretrained(cls, dirname: str, device="cuda", dtype=torch.bfloat16):
        """``text_encoder/`` of a CogVideoX / DOVE checkpoint: config.json + (sharded) model*.safetensors."""
        from safetensors.torch import load_file
        with open(os.path.join(dirname, "config.json")) as f:
            cfg = json.load(f)
        idx = os.path.join(dirname, "model.safetensors.index.json")
        if os.path.exists(idx):
            with open(idx) as f:
                files = sorted(set(json.load(f)["weight_map"].values()))
        else:
            files = [fn for fn in sorted(os.listdir(dirname)) if fn.endswith(".safetensors")]
        if not files:
            raise FileNotFoundError(f"no safetensors weights under {dirname}")
        state = {}
        for fn in files:
            state.update(load_file(os.path.join(dirname, fn)))
        spec = t5_param_shapes(cfg)
        missing = [k for k in spec if k not in state and not (k == "shared.weight" and "encoder.embed_tokens.weight" in state)]
        if missing:
            raise RuntimeError(f"{dirname}: state dict mismatch; missing={missing[:8]}")
        for k, shp in spec.items():
            if k in state and tuple(state[k].shape) != tuple(shp):
                raise RuntimeError(f"{dirname}: {k} has shape {tuple(state[k].shape)}, expected {tuple(shp)}")
        return cls(cfg, state, device, dtype)
