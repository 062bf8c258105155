"""CogVideoXDPMScheduler facade: only what the one-step path uses (`from_config`, `.config`, `get_velocity`,
`add_noise`; /root/reference/inference_script.py:457,491-493,629-631).  SURVEY.md App. A.6."""
from __future__ import annotations

import torch

from . import ops
from .config import AttrDict


class CogVideoXDPMScheduler:
    def __init__(self, **config):
        self.config = AttrDict(config)
        c = self.config
        n = c.get("num_train_timesteps", 1000)
        b0, b1 = c.get("beta_start", 0.00085), c.get("beta_end", 0.012)
        sched = c.get("beta_schedule", "scaled_linear")
        if sched == "scaled_linear":
            betas = torch.linspace(b0 ** 0.5, b1 ** 0.5, n, dtype=torch.float64) ** 2
        elif sched == "linear":
            betas = torch.linspace(b0, b1, n, dtype=torch.float64)
        else:
            raise NotImplementedError(f"beta_schedule {sched}")
        ac = torch.cumprod(1.0 - betas, dim=0)
        # absent keys take diffusers' CogVideoXDPMScheduler class defaults (snr_shift_scale 3.0, rescale_betas_zero_snr False);
        # the CogVideoX1.5 / DOVE scheduler_config.json states both explicitly (1.0 / true; dove_amd.config.SCHEDULER_CONFIG)
        s = c.get("snr_shift_scale", 3.0)
        ac = ac / (s + (1 - s) * ac)
        if c.get("rescale_betas_zero_snr", False):
            r = ac.sqrt()
            r0, rT = r[0].clone(), r[-1].clone()
            r = (r - rT) * r0 / (r0 - rT)
            ac = r ** 2
        self.alphas_cumprod = ac.to(torch.float32)

    @classmethod
    def from_config(cls, config, **overrides):
        cfg = dict(config)
        cfg.update(overrides)
        return cls(**cfg)

    def _coeffs(self, timesteps, dtype):
        """sqrt(a), sqrt(1-a) with alpha cast to the SAMPLE dtype before the sqrt, like diffusers (bf16: 0.625 /
        0.78125 at t=399 instead of 0.62733 / 0.77875)."""
        ts = set(int(v) for v in timesteps.reshape(-1).tolist())
        if len(ts) != 1:
            raise NotImplementedError("per-sample timesteps differ; the one-step path uses one timestep")
        a = self.alphas_cumprod.to(dtype)[ts.pop()]
        return float(a ** 0.5), float((1 - a) ** 0.5)

    # a*x + b*y is ONE fp32 FMA chain rounded once to the sample dtype; diffusers rounds each bf16 product and the sum
    # separately (<= 1 bf16 ulp apart; covered by the stage tolerances of tests/test_parity_gpu.py)
    def get_velocity(self, sample, noise, timesteps):
        sa, s1 = self._coeffs(timesteps, sample.dtype)
        return ops.axpby(noise.contiguous(), sample.contiguous(), sa, -s1)

    def add_noise(self, original_samples, noise, timesteps):
        sa, s1 = self._coeffs(timesteps, original_samples.dtype)
        return ops.axpby(original_samples.contiguous(), noise.contiguous(), sa, s1)
