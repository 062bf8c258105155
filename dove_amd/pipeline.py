"""CogVideoXPipeline-shaped facade: the duck-typed ``pipe`` object /root/reference/inference_script.py drives
(:613 from_pretrained, :629 scheduler swap, :641 .to("cuda"), :644-645 VAE toggles, :500 decode_latents).
``DiffusionPipeline.__call__`` is never invoked by the reference and is not provided."""
from __future__ import annotations

import json
import os

import torch

from . import weights as W
from .config import default_configs
from .scheduler import CogVideoXDPMScheduler
from .transformer import CogVideoXTransformer3DModel
from .vae import AutoencoderKLCogVideoX


class _Unavailable:
    def __init__(self, what):
        self._what = what

    def __call__(self, *a, **k):
        raise NotImplementedError(f"{self._what} is not part of the accelerated path: DOVE runs with the cached empty-prompt "
                                  "embedding (ref :423-428); non-empty prompts are a SURVEY 8(f) follow-up")

    __getattr__ = lambda self, n: self.__call__()   # noqa: E731


class CogVideoXPipeline:
    def __init__(self, vae, transformer, scheduler):
        self.vae, self.transformer, self.scheduler = vae, transformer, scheduler
        self.tokenizer = _Unavailable("tokenizer")
        self.text_encoder = _Unavailable("text_encoder (T5)")

    # ---- constructors ---------------------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, model_path, torch_dtype=torch.bfloat16, device="cuda", **kw):
        """Read a CogVideoX1.5 / DOVE checkpoint directory (vae/, transformer/, scheduler/ with config.json +
        safetensors shards; /root/reference/finetune/scripts/prepare_sft_ckpt.py:43-69)."""
        cls._check_dtype(torch_dtype)
        vcfg, vsd = W.load_component(os.path.join(model_path, "vae"), W.vae_param_shapes)
        tcfg, tsd = W.load_component(os.path.join(model_path, "transformer"), W.dit_param_shapes)
        with open(os.path.join(model_path, "scheduler", "scheduler_config.json")) as f:
            scfg = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
        return cls(AutoencoderKLCogVideoX(vcfg, vsd, device, torch_dtype),
                   CogVideoXTransformer3DModel(tcfg, tsd, device, torch_dtype), CogVideoXDPMScheduler(**scfg))

    @classmethod
    def from_config(cls, vae_config=None, transformer_config=None, scheduler_config=None, seed: int = 1234,
                    torch_dtype=torch.bfloat16, device="cuda", init_device="cpu"):
        """Deterministic random-init pipeline for synthetic benchmarks / parity tests (no weights offline)."""
        cls._check_dtype(torch_dtype)
        dv, dt, ds = default_configs()
        vcfg, tcfg, scfg = vae_config or dv, transformer_config or dt, scheduler_config or ds
        vsd = W.random_state_dict(W.vae_param_shapes(vcfg), seed, init_device)
        vae = AutoencoderKLCogVideoX(vcfg, vsd, device, torch_dtype)
        del vsd
        tsd = W.random_state_dict(W.dit_param_shapes(tcfg), seed, init_device)
        tr = CogVideoXTransformer3DModel(tcfg, tsd, device, torch_dtype)
        del tsd
        return cls(vae, tr, CogVideoXDPMScheduler(**scfg))

    @staticmethod
    def _check_dtype(dt):
        if dt != torch.bfloat16:
            raise NotImplementedError("the HIP path computes in bf16 (fp32 accumulation); pass torch_dtype=torch.bfloat16 "
                                      "(the reference's default, ref :525)")

    # ---- reference surface ------------------------------------------------------------------------
    @property
    def device(self):
        return self.vae.device

    def to(self, *a, **k):
        return self

    def enable_sequential_cpu_offload(self, *a, **k):
        return self  # 288 GB HBM: nothing to offload

    enable_model_cpu_offload = enable_sequential_cpu_offload

    def load_lora_weights(self, *a, **k):
        raise NotImplementedError("LoRA fuse is a SURVEY 8(f) follow-up (DOVE's released checkpoint is full-SFT)")

    fuse_lora = load_lora_weights

    @torch.no_grad()
    def decode_latents(self, latents: torch.Tensor, _range01: bool = False) -> torch.Tensor:
        """[B,T,C,h,w] -> [B,3,F,H,W]: permute, x 1/scaling_factor, vae.decode(...).sample (diffusers
        CogVideoXPipeline.decode_latents).  ``_range01`` fuses the caller's (x*0.5+0.5).clamp(0,1) (ref :501)."""
        z = latents.permute(0, 2, 1, 3, 4).contiguous()
        sf = float(self.vae.config["scaling_factor"])
        # the 1/scaling_factor multiply is folded into the layout kernel (same single bf16 rounding)
        return self.vae.decode(z, _range01=_range01, _prescale=1.0 / sf).sample
