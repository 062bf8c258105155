"""Escape hatch for "parity unpinned" (DESIGN.md section 2): run on a machine that HAS diffusers + the DOVE /
CogVideoX1.5-5B checkpoint to dump per-stage tensors of the reference path for a seed-fixed synthetic clip, in the
fixture format tests/ can consume (safetensors: inputs + expected outputs).  Not runnable in the build container
(no diffusers, no weights); it imports nothing from /root/reference.

    python tools/capture_goldens.py --model_path pretrained_models/DOVE --out tests/golden/diffusers_stages.safetensors
"""
import argparse

import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model_path", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--frames", type=int, default=9)
    ap.add_argument("--height", type=int, default=64)
    ap.add_argument("--width", type=int, default=64)
    ap.add_argument("--dtype", default="float32")
    args = ap.parse_args()
    from diffusers import CogVideoXDPMScheduler, CogVideoXPipeline  # noqa: E402  (only on the capture machine)
    from diffusers.models.embeddings import get_3d_rotary_pos_embed
    from safetensors.torch import load_file, save_file

    dt = getattr(torch, args.dtype)
    pipe = CogVideoXPipeline.from_pretrained(args.model_path, torch_dtype=dt)
    pipe.scheduler = CogVideoXDPMScheduler.from_config(pipe.scheduler.config, timestep_spacing="trailing")
    g = torch.Generator().manual_seed(42)
    video = (torch.rand(1, 3, args.frames, args.height, args.width, generator=g) * 2 - 1).to(dt)
    text = load_file("tests/golden/empty_prompt_embedding.safetensors")["prompt_embedding"][None].to(dt)
    out = {"video": video, "text": text}
    with torch.no_grad():
        dist = pipe.vae.encode(video).latent_dist
        out["moments"] = dist.parameters
        noise = torch.randn(dist.mean.shape, generator=g).to(dt)
        out["noise"] = noise
        latent = (dist.mean + dist.std * noise) * pipe.vae.config.scaling_factor
        pt = pipe.transformer.config.patch_size_t
        ncopy = latent.shape[2] % pt
        latent = torch.cat([latent[:, :, :1].repeat(1, 1, ncopy, 1, 1), latent], dim=2).permute(0, 2, 1, 3, 4)
        B, T, C, h, w = latent.shape
        p = pipe.transformer.config.patch_size
        cos, sin = get_3d_rotary_pos_embed(pipe.transformer.config.attention_head_dim, None, (h // p, w // p), (T + pt - 1) // pt,
                                           grid_type="slice", max_size=(h // p, w // p))
        ts = torch.full((1,), 399, dtype=torch.long)
        v = pipe.transformer(hidden_states=latent, encoder_hidden_states=text, timestep=ts, image_rotary_emb=(cos, sin),
                             return_dict=False)[0]
        out.update(latent=latent, rope_cos=cos, rope_sin=sin, velocity=v)
        x0 = pipe.scheduler.get_velocity(v, latent, ts)[:, ncopy:]
        out["x0"] = x0
        vid = pipe.decode_latents(x0)
        out["sr"] = (vid * 0.5 + 0.5).clamp(0, 1)
    save_file({k: t.contiguous().float() for k, t in out.items()}, args.out)
    print("wrote", args.out, {k: tuple(t.shape) for k, t in out.items()})


if __name__ == "__main__":
    main()
