"""State-dict specification, deterministic random init and checkpoint loading.

Names/shapes follow the diffusers CogVideoX state dict the reference loads through
``CogVideoXPipeline.from_pretrained`` (/root/reference/inference_script.py:613) and that
/root/reference/finetune/scripts/prepare_sft_ckpt.py:17-69 writes (fp32 transformer shards +
index json); SURVEY.md App. E.  The loader is key-driven and fails loudly on any mismatch.
"""
from __future__ import annotations

import hashlib
import json
import os
from collections import OrderedDict

import torch


# ---- specification -----------------------------------------------------------------------------
def vae_param_shapes(cfg: dict) -> "OrderedDict[str, tuple]":
    boc = list(cfg["block_out_channels"])
    L = cfg.get("layers_per_block", 3)
    lat = cfg["latent_channels"]
    cin, cout = cfg["in_channels"], cfg["out_channels"]
    s: "OrderedDict[str, tuple]" = OrderedDict()

    def conv3(name, ci, co, k=3):
        s[name + ".conv.weight"] = (co, ci, k, k, k)
        s[name + ".conv.bias"] = (co,)

    def gn(name, c):
        s[name + ".weight"] = (c,)
        s[name + ".bias"] = (c,)

    def sn(name, c):
        gn(name + ".norm_layer", c)
        conv3(name + ".conv_y", lat, c, 1)
        conv3(name + ".conv_b", lat, c, 1)

    def resnet(name, ci, co, spatial):
        (sn if spatial else gn)(name + ".norm1", ci)
        conv3(name + ".conv1", ci, co)
        (sn if spatial else gn)(name + ".norm2", co)
        conv3(name + ".conv2", co, co)
        if ci != co:
            s[name + ".conv_shortcut.weight"] = (co, ci, 1, 1, 1)
            s[name + ".conv_shortcut.bias"] = (co,)

    def conv2(name, c):
        s[name + ".conv.weight"] = (c, c, 3, 3)
        s[name + ".conv.bias"] = (c,)

    # encoder
    conv3("encoder.conv_in", cin, boc[0])
    ch = boc[0]
    for i, co in enumerate(boc):
        for j in range(L):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}", ch, co, False)
            ch = co
        if i < len(boc) - 1:
            conv2(f"encoder.down_blocks.{i}.downsamplers.0", ch)
    for j in range(2):
        resnet(f"encoder.mid_block.resnets.{j}", ch, ch, False)
    gn("encoder.norm_out", ch)
    conv3("encoder.conv_out", ch, 2 * lat)
    # decoder
    rev = boc[::-1]
    conv3("decoder.conv_in", lat, rev[0])
    ch = rev[0]
    for j in range(2):
        resnet(f"decoder.mid_block.resnets.{j}", ch, ch, True)
    for i, co in enumerate(rev):
        for j in range(L + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", ch, co, True)
            ch = co
        if i < len(rev) - 1:
            conv2(f"decoder.up_blocks.{i}.upsamplers.0", ch)
    sn("decoder.norm_out", ch)
    conv3("decoder.conv_out", ch, cout)
    return s


def dit_param_shapes(cfg: dict) -> "OrderedDict[str, tuple]":
    D = cfg["num_attention_heads"] * cfg["attention_head_dim"]
    hd = cfg["attention_head_dim"]
    te = cfg["time_embed_dim"]
    p, pt = cfg["patch_size"], cfg["patch_size_t"] or 1
    s: "OrderedDict[str, tuple]" = OrderedDict()

    def lin(name, ci, co):
        s[name + ".weight"] = (co, ci)
        s[name + ".bias"] = (co,)

    def ln(name, c):
        s[name + ".weight"] = (c,)
        s[name + ".bias"] = (c,)

    lin("patch_embed.proj", cfg["in_channels"] * p * p * pt, D)
    lin("patch_embed.text_proj", cfg["text_embed_dim"], D)
    lin("time_embedding.linear_1", D, te)
    lin("time_embedding.linear_2", te, te)
    for i in range(cfg["num_layers"]):
        b = f"transformer_blocks.{i}."
        lin(b + "norm1.linear", te, 6 * D)
        ln(b + "norm1.norm", D)
        for n in ("to_q", "to_k", "to_v"):
            lin(b + "attn1." + n, D, D)
        ln(b + "attn1.norm_q", hd)
        ln(b + "attn1.norm_k", hd)
        lin(b + "attn1.to_out.0", D, D)
        lin(b + "norm2.linear", te, 6 * D)
        ln(b + "norm2.norm", D)
        lin(b + "ff.net.0.proj", D, 4 * D)
        lin(b + "ff.net.2", 4 * D, D)
    ln("norm_final", D)
    lin("norm_out.linear", te, 2 * D)
    ln("norm_out.norm", D)
    lin("proj_out", D, cfg["out_channels"] * p * p * pt)
    return s


def count_params(shapes) -> int:
    n = 0
    for shp in shapes.values():
        k = 1
        for d in shp:
            k *= d
        n += k
    return n


# ---- random init ---------------------------------------------------------------------------------
def _seed_for(name: str, seed: int) -> int:
    h = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    return int.from_bytes(h[:7], "little")


def random_state_dict(shapes, seed: int = 1234, device="cpu", dtype=torch.float32):
    """Deterministic synthetic weights (one generator per tensor, keyed by name): linears/convs
    N(0, 1/fan_in) so activations keep unit scale through the deep stacks, biases N(0, 0.02^2),
    norm gammas 1 + N(0, 0.02^2), norm betas N(0, 0.02^2) (non-trivial so affine bugs show)."""
    out = OrderedDict()
    dev = torch.device(device)
    for name, shp in shapes.items():
        g = torch.Generator(device=dev)
        g.manual_seed(_seed_for(name, seed))
        if len(shp) >= 2:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            t = torch.randn(shp, generator=g, device=dev, dtype=torch.float32) * (fan_in ** -0.5)
        elif name.endswith(".weight"):
            t = 1.0 + 0.02 * torch.randn(shp, generator=g, device=dev, dtype=torch.float32)
        else:
            t = 0.02 * torch.randn(shp, generator=g, device=dev, dtype=torch.float32)
        out[name] = t.to(dtype)
    return out


class LazyStateDict:
    """Read-only mapping with the tensors of ``random_state_dict(shapes, seed, device)`` generated one at a time on
    access (never all resident: the full 42-layer DiT is 22 GB in fp32).  Same values as ``random_state_dict`` for the
    same (seed, device type); ``to`` moves every generated tensor (e.g. a CUDA-generated set read by the CPU oracle)."""

    def __init__(self, shapes, seed: int = 1234, device="cpu", dtype=torch.float32, to=None, scale: dict | None = None):
        self._shapes, self._seed, self._device, self._dtype, self._to = shapes, seed, torch.device(device), dtype, to
        self._scale = scale or {}

    def __getitem__(self, name):
        t = random_state_dict({name: self._shapes[name]}, self._seed, self._device, self._dtype)[name]
        if name in self._scale:
            sc = self._scale[name]                        # a number, or a tensor broadcast over the weight (per-channel factors)
            t = t * (sc.to(t.device, t.dtype) if torch.is_tensor(sc) else sc)
        return t if self._to is None else t.to(self._to)

    def get(self, name, default=None):
        return self[name] if name in self._shapes else default

    def __contains__(self, name):
        return name in self._shapes

    def __iter__(self):
        return iter(self._shapes)

    def __len__(self):
        return len(self._shapes)

    def keys(self):
        return self._shapes.keys()

    def items(self):
        return ((k, self[k]) for k in self._shapes)

    def moved(self, to):
        return LazyStateDict(self._shapes, self._seed, self._device, self._dtype, to, self._scale)


# ---- checkpoint loading --------------------------------------------------------------------------
def _load_component_state(dirname: str) -> dict:
    from safetensors.torch import load_file

    idx = os.path.join(dirname, "diffusion_pytorch_model.safetensors.index.json")
    single = os.path.join(dirname, "diffusion_pytorch_model.safetensors")
    state = {}
    if os.path.exists(idx):
        with open(idx) as f:
            files = sorted(set(json.load(f)["weight_map"].values()))
        for fn in files:
            state.update(load_file(os.path.join(dirname, fn)))
    elif os.path.exists(single):
        state.update(load_file(single))
    else:
        raise FileNotFoundError(f"no safetensors weights under {dirname}")
    return state


def load_component(dirname: str, shapes_fn):
    """Read ``config.json`` + sharded safetensors of one component (``vae`` / ``transformer``), validate
    every key and shape against the spec."""
    with open(os.path.join(dirname, "config.json")) as f:
        cfg = json.load(f)
    state = _load_component_state(dirname)
    spec = shapes_fn(cfg)
    missing = [k for k in spec if k not in state]
    extra = [k for k in state if k not in spec]
    if missing or extra:
        raise RuntimeError(f"{dirname}: state dict mismatch; missing={missing[:8]} unexpected={extra[:8]}")
    for k, shp in spec.items():
        if tuple(state[k].shape) != tuple(shp):
            raise RuntimeError(f"{dirname}: {k} has shape {tuple(state[k].shape)}, expected {tuple(shp)}")
    return cfg, state
