"""The one-step SR operator and the script-level driver around it.

``process_video`` mirrors /root/reference/inference_script.py:394-503 call for call (same argument names and
meaning) on the facade objects of dove_amd.pipeline; ``run_clip`` mirrors the chunk x tile loop, stitching and
un-padding of the reference's ``__main__`` (:670-731).  Pre/post-processing that touches files (decord, mp4)
stays outside the accelerated path.
"""
from __future__ import annotations

import torch

from . import tiling
from .rope import prepare_rotary_positional_embeddings


@torch.no_grad()
def process_video(pipe, video: torch.Tensor, prompt: str = "", noise_step: int = 0, sr_noise_step: int = 399,
                  empty_prompt_embedding: torch.Tensor = None, posterior_noise: torch.Tensor = None,
                  generator: torch.Generator = None, _stages: dict = None) -> torch.Tensor:
    """video [B,3,F,H,W] in [-1,1] -> SR video [B,3,F,H,W] in [0,1].

    ``posterior_noise`` ([B,L,T,h,w]) / ``generator`` make the VAE posterior sample reproducible across devices
    (the reference draws it from the global RNG, ref :409).  ``_stages`` (parity checks: bench.py): a dict that receives the
    intermediate tensors under the oracle's trace names - moments, latent, v, x0."""
    video = video.to(pipe.vae.device, dtype=pipe.vae.dtype)
    latent_dist = pipe.vae.encode(video).latent_dist
    if _stages is not None:
        _stages["moments"] = latent_dist.parameters
    latent = latent_dist.sample(generator=generator, noise=posterior_noise) * pipe.vae.config.scaling_factor

    patch_size_t = pipe.transformer.config.patch_size_t
    ncopy = 0
    if patch_size_t is not None:
        ncopy = latent.shape[2] % patch_size_t
        first = latent[:, :, :1]
        latent = torch.cat([first.repeat(1, 1, ncopy, 1, 1), latent], dim=2)
        assert latent.shape[2] % patch_size_t == 0
    batch_size, num_channels, num_frames, height, width = latent.shape

    if prompt == "" and empty_prompt_embedding is not None:
        prompt_embedding = empty_prompt_embedding.to(latent.device, dtype=latent.dtype)
        if prompt_embedding.dim() == 2:
            prompt_embedding = prompt_embedding[None]
        if prompt_embedding.shape[0] != batch_size:
            prompt_embedding = prompt_embedding.repeat(batch_size, 1, 1)
    else:
        ids = pipe.tokenizer(prompt, padding="max_length", max_length=pipe.transformer.config.max_text_seq_length,
                             truncation=True, add_special_tokens=True, return_tensors="pt").input_ids
        prompt_embedding = pipe.text_encoder(ids.to(latent.device))[0]
        prompt_embedding = prompt_embedding.view(batch_size, prompt_embedding.shape[1], -1).to(dtype=latent.dtype)

    latent = latent.permute(0, 2, 1, 3, 4).contiguous()
    if noise_step != 0:
        noise = torch.randn_like(latent)
        add_t = torch.full((batch_size,), noise_step, dtype=torch.long, device=latent.device)
        latent = pipe.scheduler.add_noise(latent, noise, add_t)
    timesteps = torch.full((batch_size,), sr_noise_step, dtype=torch.long, device=latent.device)

    vsf = 2 ** (len(pipe.vae.config.block_out_channels) - 1)
    rotary_emb = (prepare_rotary_positional_embeddings(height=height * vsf, width=width * vsf, num_frames=num_frames,
                                                       transformer_config=pipe.transformer.config,
                                                       vae_scale_factor_spatial=vsf, device=latent.device)
                  if pipe.transformer.config.use_rotary_positional_embeddings else None)

    predicted = pipe.transformer(hidden_states=latent, encoder_hidden_states=prompt_embedding, timestep=timesteps,
                                 image_rotary_emb=rotary_emb, return_dict=False)[0]
    latent_generate = pipe.scheduler.get_velocity(predicted, latent, timesteps)
    if patch_size_t is not None and ncopy > 0:
        latent_generate = latent_generate[:, ncopy:]
    if _stages is not None:
        _stages.update(latent=latent, v=predicted, x0=latent_generate)
    # decode + (x*0.5+0.5).clamp(0,1) (ref :500-501), the range map fused into the decoder's last layout kernel
    return pipe.decode_latents(latent_generate, _range01=True)


@torch.no_grad()
def run_clip(pipe, video: torch.Tensor, *, chunk_len: int = 0, overlap_t: int = 8, tile_size_hw=(0, 0), overlap_hw=(32, 32),
             empty_prompt_embedding=None, sr_noise_step: int = 399, noise_step: int = 0, work_filter=None,
             generator=None) -> tuple:
    """Chunk x tile loop + stitch of ref :682-729 on a pre-processed [1,3,F,H,W] clip in [-1,1].
    ``work_filter(i, n)`` selects the work items this rank owns (multi-GPU chunk farm, dove_amd.dist).
    Returns (output_video, write_count) on the host like the reference."""
    items = tiling.plan(video.shape, chunk_len, overlap_t, tile_size_hw, overlap_hw)
    out = torch.zeros(video.shape, dtype=torch.float32)
    wc = torch.zeros(video.shape, dtype=torch.int32)
    for i, ((t0, t1, h0, h1, w0, w1), region) in enumerate(items):
        if work_filter is not None and not work_filter(i, len(items)):
            continue
        piece = process_video(pipe, video[:, :, t0:t1, h0:h1, w0:w1], sr_noise_step=sr_noise_step, noise_step=noise_step,
                              empty_prompt_embedding=empty_prompt_embedding, generator=generator)
        tiling.stitch(out, wc, piece.float().cpu(), region)
    return out, wc
