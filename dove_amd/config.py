"""Model hyper-parameters for the DOVE hot path.

DOVE = CogVideoX1.5-5B with the transformer fine-tuned (/root/reference/README.md:165-166,181;
/root/reference/finetune/train_ddp_one_s1.sh:8).  The reference reads every value below from the
checkpoint's ``config.json`` files through diffusers (/root/reference/inference_script.py:613);
``Pipeline.from_pretrained`` does the same here and these dicts are only the defaults used for
random-init synthetic benchmarks (no weights are available offline).
"""
from __future__ import annotations

import copy

VAE_CONFIG = {
    "in_channels": 3,
    "out_channels": 3,
    "latent_channels": 16,
    "block_out_channels": [128, 256, 256, 512],
    "layers_per_block": 3,
    "act_fn": "silu",
    "norm_eps": 1e-6,
    "norm_num_groups": 32,
    "temporal_compression_ratio": 4,
    "scaling_factor": 0.7,
    "sample_height": 480,
    "sample_width": 720,
    "use_quant_conv": False,
    "use_post_quant_conv": False,
    "num_sample_frames_batch_size": 8,
    "num_latent_frames_batch_size": 2,
}

TRANSFORMER_CONFIG = {
    "num_attention_heads": 48,
    "attention_head_dim": 64,
    "in_channels": 16,
    "out_channels": 16,
    "num_layers": 42,
    "patch_size": 2,
    "patch_size_t": 2,
    "text_embed_dim": 4096,
    "time_embed_dim": 512,
    "flip_sin_to_cos": True,
    "freq_shift": 0,
    "timestep_activation_fn": "silu",
    "activation_fn": "gelu-approximate",
    "attention_bias": True,
    "norm_elementwise_affine": True,
    "norm_eps": 1e-5,
    "use_rotary_positional_embeddings": True,
    "use_learned_positional_embeddings": False,
    "max_text_seq_length": 226,
    "patch_bias": True,
}

SCHEDULER_CONFIG = {
    "num_train_timesteps": 1000,
    "beta_start": 0.00085,
    "beta_end": 0.012,
    "beta_schedule": "scaled_linear",
    "prediction_type": "v_prediction",
    "rescale_betas_zero_snr": True,
    "snr_shift_scale": 1.0,
    "timestep_spacing": "trailing",
}


def default_configs():
    return (copy.deepcopy(VAE_CONFIG), copy.deepcopy(TRANSFORMER_CONFIG), copy.deepcopy(SCHEDULER_CONFIG))


def small_configs(num_layers: int = 2):
    """Same architecture and widths as CogVideoX1.5-5B with fewer DiT layers -- used by parity tests so the
    fp32 CPU oracle finishes in seconds.  Every kernel shape class of the full model is exercised."""
    v, t, s = default_configs()
    t["num_layers"] = num_layers
    return v, t, s


def tiny_configs():
    """Narrow model (same topology) for CPU-only plumbing tests of the host logic."""
    v, t, s = default_configs()
    v["block_out_channels"] = [32, 64, 64, 128]
    v["layers_per_block"] = 1
    t.update(num_attention_heads=4, attention_head_dim=64, num_layers=2, text_embed_dim=128, time_embed_dim=64)
    return v, t, s


class AttrDict(dict):
    """diffusers' FrozenDict-style config: attribute AND item access
    (/root/reference/inference_script.py:411,467 use ``pipe.*.config.<name>``)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e
