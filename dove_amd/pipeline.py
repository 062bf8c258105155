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
        raise NotImplementedError(f"this pipeline was built without a {self._what} (the checkpoint has no such directory, or it "
                                  "is a random-init pipeline): only the empty prompt with the cached embedding (ref :423-428) "
                                  "can run; load a checkpoint with text_encoder/ and tokenizer/ for non-empty prompts")

    __getattr__ = lambda self, n: self.__call__()   # noqa: E731


class _Lazy:
    """``pipe.text_encoder`` / ``pipe.tokenizer`` of a loaded checkpoint: the object is built by ``loader()`` on first use (call or
    attribute access) and kept."""

    def __init__(self, what, loader):
        self.__dict__.update(_what=what, _loader=loader, _obj=None)

    def _get(self):
        if self._obj is None:
            self.__dict__["_obj"] = self._loader()
        return self._obj

    @property
    def loaded(self):
        return self._obj is not None

    def __call__(self, *a, **k):
        return self._get()(*a, **k)

    def __getattr__(self, n):
        return getattr(self._get(), n)


class CogVideoXPipeline:
    def __init__(self, vae, transformer, scheduler, text_encoder=None, tokenizer=None):
        self.vae, self.transformer, self.scheduler = vae, transformer, scheduler
        # only a non-empty prompt needs them (ref :429-444); a checkpoint without text_encoder/ or tokenizer/ still runs the
        # documented empty-prompt path
        self.tokenizer = tokenizer if tokenizer is not None else _Unavailable("tokenizer")
        self.text_encoder = text_encoder if text_encoder is not None else _Unavailable("text_encoder (T5)")

    # ---- constructors ---------------------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, model_path, torch_dtype=torch.bfloat16, device="cuda", dit_linear_precision="bf16", dit_attention_precision="bf16", **kw):
        """Read a CogVideoX1.5 / DOVE checkpoint directory (vae/, transformer/, scheduler/ with config.json +
        safetensors shards; /root/reference/finetune/scripts/prepare_sft_ckpt.py:43-69)."""
        cls._check_dtype(torch_dtype)
        vcfg, vsd = W.load_component(os.path.join(model_path, "vae"), W.vae_param_shapes)
        tcfg, tsd = W.load_component(os.path.join(model_path, "transformer"), W.dit_param_shapes)
        with open(os.path.join(model_path, "scheduler", "scheduler_config.json")) as f:
            scfg = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
        # The documented runs use the cached empty-prompt embedding (ref :423-428): the T5-XXL encoder (~9.5 GB packed) and
        # transformers' tokenizer are constructed on the first non-empty prompt, not here
        text_encoder = tokenizer = None
        te_dir, tok_dir = os.path.join(model_path, "text_encoder"), os.path.join(model_path, "tokenizer")
        if os.path.isdir(te_dir):
            def _load_te(te_dir=te_dir):
                from .t5 import T5EncoderModel
                return T5EncoderModel.from_pretrained(te_dir, device, torch_dtype)
            text_encoder = _Lazy("text_encoder (T5)", _load_te)
        if os.path.isdir(tok_dir):
            def _load_tok(tok_dir=tok_dir):
                try:      # host-side text processing: transformers' own T5 tokenizer, exactly what the reference's pipeline holds
                    from transformers import AutoTokenizer
                except ImportError as e:
                    raise NotImplementedError("a non-empty prompt needs the `transformers` package for the T5 tokenizer "
                                              "(the empty prompt uses the shipped embedding and needs neither)") from e
                return AutoTokenizer.from_pretrained(tok_dir)
            tokenizer = _Lazy("tokenizer", _load_tok)
        return cls(AutoencoderKLCogVideoX(vcfg, vsd, device, torch_dtype),
                   CogVideoXTransformer3DModel(tcfg, tsd, device, torch_dtype, dit_linear_precision, dit_attention_precision), CogVideoXDPMScheduler(**scfg),
                   text_encoder, tokenizer)

    @classmethod
    def from_config(cls, vae_config=None, transformer_config=None, scheduler_config=None, seed: int = 1234,
                    torch_dtype=torch.bfloat16, device="cuda", init_device="cpu", dit_linear_precision="bf16",
                    dit_attention_precision="bf16"):
        """Deterministic random-init pipeline for synthetic benchmarks / parity tests (no weights offline)."""
        cls._check_dtype(torch_dtype)
        dv, dt, ds = default_configs()
        vcfg, tcfg, scfg = vae_config or dv, transformer_config or dt, scheduler_config or ds
        # tensors are generated one at a time while they are packed (the 42-layer DiT is 22 GB in fp32)
        vae = AutoencoderKLCogVideoX(vcfg, W.LazyStateDict(W.vae_param_shapes(vcfg), seed, init_device), device, torch_dtype)
        tr = CogVideoXTransformer3DModel(tcfg, W.LazyStateDict(W.dit_param_shapes(tcfg), seed, init_device), device, torch_dtype,
                                         dit_linear_precision, dit_attention_precision)
        return cls(vae, tr, CogVideoXDPMScheduler(**scfg))

    @staticmethod
    def _check_dtype(dt):
        if dt != torch.bfloat16:
            raise NotImplementedError(f"torch_dtype={dt}: the HIP path computes in bf16 (fp32 accumulation); pass torch_dtype=torch.bfloat16, "
                                      "the reference's default (ref :525); float16 / float32 are not implemented (INTEGRATION.md, 'dtype')")

    # ---- reference surface ------------------------------------------------------------------------
    @property
    def device(self):
        return self.vae.device

    def to(self, *a, **k):
        return self

    def enable_sequential_cpu_offload(self, *a, **k):
        return self  # 288 GB HBM: nothing to offload

    enable_model_cpu_offload = enable_sequential_cpu_offload

    def load_lora_weights(self, path, weight_name="pytorch_lora_weights.safetensors", adapter_name=None, **kw):
        """Read a diffusers/peft LoRA file (ref :616-620): keys ``transformer.<module>.lora_A.weight`` [r, in] and
        ``.lora_B.weight`` [out, r] on the attention projections (DOVE's targets: to_q, to_k, to_v, to_out.0;
        /root/reference/finetune/schemas/args.py:71-73).  The adapter is kept on the host until ``fuse_lora``."""
        import json as _json

        from safetensors import safe_open
        fn = os.path.join(path, weight_name) if os.path.isdir(path) else path
        sd, meta = {}, {}
        with safe_open(fn, framework="pt") as f:
            meta = f.metadata() or {}
            for k in f.keys():
                sd[k] = f.get_tensor(k)
        alpha = rank = None
        if "lora_adapter_metadata" in meta:      # newer diffusers store the peft config next to the tensors
            cfg = _json.loads(meta["lora_adapter_metadata"])
            cfg = cfg.get("transformer", cfg)
            alpha, rank = cfg.get("lora_alpha"), cfg.get("r")
        self._lora = dict(state=sd, alpha=alpha, rank=rank, name=adapter_name)

    def fuse_lora(self, components=("transformer",), lora_scale: float = 1.0, **kw):
        """W += lora_scale * (alpha / r) * B @ A on every adapted Linear of the transformer (ref :621), then drop the
        adapter.  Without stored alpha the peft default alpha == r (scale 1) is assumed, like diffusers."""
        if getattr(self, "_lora", None) is None:
            raise RuntimeError("fuse_lora() called before load_lora_weights()")
        if "transformer" not in components:
            return
        lo = self._lora
        pairs = {}
        for k, v in lo["state"].items():
            k2 = k[len("transformer."):] if k.startswith("transformer.") else k
            for tag in (".lora_A.weight", ".lora_B.weight", ".lora.down.weight", ".lora.up.weight"):
                if k2.endswith(tag):
                    which = "A" if ("lora_A" in tag or "down" in tag) else "B"
                    pairs.setdefault(k2[: -len(tag)], {})[which] = v
        if not pairs:
            raise RuntimeError("no lora_A / lora_B tensors found in the adapter file")
        for mod, ab in pairs.items():
            if "A" not in ab or "B" not in ab:
                raise RuntimeError(f"incomplete LoRA pair for {mod}")
            r = ab["A"].shape[0]
            scale = lora_scale * ((lo["alpha"] / (lo["rank"] or r)) if lo["alpha"] is not None else 1.0)
            self.transformer.add_weight_delta(mod, ab["B"].float() @ ab["A"].float(), scale)
        self.transformer._mod_cache = {}
        self._lora = None

    @torch.no_grad()
    def decode_latents(self, latents: torch.Tensor, _range01: bool = False) -> torch.Tensor:
        """[B,T,C,h,w] -> [B,3,F,H,W]: permute, x 1/scaling_factor, vae.decode(...).sample (diffusers
        CogVideoXPipeline.decode_latents).  ``_range01`` fuses the caller's (x*0.5+0.5).clamp(0,1) (ref :501)."""
        z = latents.permute(0, 2, 1, 3, 4).contiguous()
        sf = float(self.vae.config["scaling_factor"])
        # the 1/scaling_factor multiply is folded into the layout kernel (same single bf16 rounding)
        return self.vae.decode(z, _range01=_range01, _prescale=1.0 / sf).sample
