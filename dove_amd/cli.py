"""Command-line driver mirroring the reference's ``python inference_script.py ...`` (ref :506-778) on the HIP path.

Every flag of the reference's parser (ref :507-554) is accepted: ``--input_dir --input_json --gt_dir --eval_metrics --model_path
--lora_path --output_path --fps --dtype --seed --upscale_mode --upscale --noise_step --sr_noise_step --is_cpu_offload --is_vae_st
--png_save --save_format --tile_size_hw --overlap_hw --chunk_len --overlap_t``.  ``--fps`` / ``--save_format`` describe the mp4
container the reference writes with imageio - frame files carry neither, so they are accepted and reported; ``--is_cpu_offload``
calls ``pipe.enable_sequential_cpu_offload()`` like the reference (a no-op with 288 GB of HBM); ``--eval_metrics`` knows ``psnr``
(the pyiqa metrics are out of scope); ``--dtype`` other than bfloat16 is refused.  Inputs are PNG folders
or ``.npy`` clips (uint8 [F,H,W,3]) because H.264 decoding (decord) is outside the accelerated path; outputs are PNG
folders or ``.npy``.  ``--random_init`` builds the CogVideoX1.5-5B architecture with synthetic weights (no checkpoint is
available offline).  Metrics (pyiqa) are not provided; ``--eval_psnr_dir`` computes plain PSNR (10*log10(1/MSE), per-frame
mean) against ground-truth folders."""
from __future__ import annotations

import argparse
import os

import torch


def main(argv=None):
    ap = argparse.ArgumentParser(description="VSR using DOVE on MI355X (dove_amd)")
    ap.add_argument("--input_dir", type=str, required=True)
    ap.add_argument("--input_json", type=str, default=None, help="{clip name: prompt}; clips without an entry use the empty prompt (ref :590-594, :676)")
    ap.add_argument("--model_path", type=str, default=None)
    ap.add_argument("--lora_path", type=str, default=None, help="LoRA weights to fuse into the transformer (ref :613-621)")
    ap.add_argument("--random_init", action="store_true")
    ap.add_argument("--output_path", type=str, default="./results")
    ap.add_argument("--gt_dir", type=str, default=None, help="ground-truth folders / .npy clips for --eval_metrics (ref :511)")
    ap.add_argument("--eval_metrics", type=str, default="", help="'psnr' (ref :513; the pyiqa metrics ssim,lpips,... are out of scope)")
    ap.add_argument("--fps", type=int, default=16, help="accepted like the reference (ref :521): frame files (PNG / .npy) carry no frame rate")
    ap.add_argument("--dtype", type=str, default="bfloat16")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--upscale_mode", type=str, default="bilinear", help="ref :527, :672; 'bilinear' is the fused HIP kernel, others run torch's interpolate")
    ap.add_argument("--upscale", type=int, default=4)
    ap.add_argument("--is_cpu_offload", action="store_true", help="ref :533, :637-641: enable_sequential_cpu_offload() (a no-op here)")
    ap.add_argument("--save_format", type=str, default="yuv444p", help="accepted like the reference (ref :541): the pixel format of ITS mp4 writer")
    ap.add_argument("--noise_step", type=int, default=0)
    ap.add_argument("--sr_noise_step", type=int, default=399)
    ap.add_argument("--is_vae_st", action="store_true")
    ap.add_argument("--png_save", action="store_true")
    ap.add_argument("--tile_size_hw", type=int, nargs=2, default=(0, 0))
    ap.add_argument("--overlap_hw", type=int, nargs=2, default=(32, 32))
    ap.add_argument("--chunk_len", type=int, default=0)
    ap.add_argument("--overlap_t", type=int, default=8)
    ap.add_argument("--eval_psnr_dir", type=str, default=None)
    ap.add_argument("--num_layers", type=int, default=None, help="debug (with --random_init): fewer DiT layers than the 42 of CogVideoX1.5-5B")
    ap.add_argument("--prompt_embedding", type=str,
                    default="pretrained_models/prompt_embeddings/e3b0c44298fc1c149afbf4c8996fb92427ae41e4649b934ca495991b7852b855.safetensors")
    args = ap.parse_args(argv)
    if args.dtype != "bfloat16":
        raise ValueError(f"--dtype {args.dtype}: the HIP path computes in bfloat16 (the reference's default, ref :525); float16 / float32 "
                         "are not implemented (INTEGRATION.md, 'dtype')")
    metrics = [m for m in args.eval_metrics.split(",") if m]
    if any(m != "psnr" for m in metrics):
        raise NotImplementedError(f"--eval_metrics {args.eval_metrics}: only 'psnr' is computed here; the pyiqa metrics are outside the path")
    if metrics and not (args.gt_dir or args.eval_psnr_dir):
        raise ValueError("--eval_metrics psnr needs --gt_dir")
    if args.gt_dir and not args.eval_psnr_dir and metrics:
        args.eval_psnr_dir = args.gt_dir

    from safetensors.torch import load_file

    from . import prepost, tiling
    from .inference import process_video
    from .pipeline import CogVideoXPipeline
    from .scheduler import CogVideoXDPMScheduler

    torch.manual_seed(args.seed)
    emb = load_file(args.prompt_embedding)["prompt_embedding"] if os.path.exists(args.prompt_embedding) else None
    if emb is None:
        raise FileNotFoundError(f"empty-prompt embedding not found at {args.prompt_embedding} (the reference ships it; ref :668-676)")
    if args.random_init or not args.model_path:
        from . import config
        v, t, s = config.default_configs()
        if args.num_layers:
            t["num_layers"] = args.num_layers
        pipe = CogVideoXPipeline.from_config(v, t, s, device="cuda", init_device="cuda")
    else:
        pipe = CogVideoXPipeline.from_pretrained(args.model_path, torch_dtype=torch.bfloat16)
    if args.lora_path:
        print(f"Loading LoRA weights from {args.lora_path}")
        pipe.load_lora_weights(args.lora_path, weight_name="pytorch_lora_weights.safetensors", adapter_name="test_1")
        pipe.fuse_lora(components=["transformer"], lora_scale=1.0)
    pipe.scheduler = CogVideoXDPMScheduler.from_config(pipe.scheduler.config, timestep_spacing="trailing")
    if args.is_cpu_offload:
        pipe.enable_sequential_cpu_offload()
    else:
        pipe.to("cuda")
    if not args.png_save:
        print(f"[dove_amd] clips are written as .npy frame arrays (uint8 [F,H,W,3]); --fps {args.fps} / --save_format {args.save_format} "
              "apply to the reference's mp4 writer only")
    if args.is_vae_st:
        pipe.vae.enable_slicing()
        pipe.vae.enable_tiling()
    overlap_t = args.overlap_t if args.chunk_len > 0 else 0
    os.makedirs(args.output_path, exist_ok=True)
    names = sorted(n for n in os.listdir(args.input_dir)
                   if n.lower().endswith(".npy") or os.path.isdir(os.path.join(args.input_dir, n)))
    if not names:
        raise ValueError(f"No clips (.npy or PNG folders) found in {args.input_dir}")
    prompts = {}
    if args.input_json is not None:
        import json
        with open(args.input_json) as f:
            prompts = json.load(f)
    psnrs = {}
    for name in names:
        prompt = prompts.get(name, "")
        frames = prepost.load_frames(os.path.join(args.input_dir, name))
        video, pad_f, pad_h, pad_w, orig = prepost.preprocess_frames(frames, args.upscale, upscale_mode=args.upscale_mode)
        items = tiling.plan(video.shape, args.chunk_len, overlap_t, tuple(args.tile_size_hw), tuple(args.overlap_hw))
        out = torch.zeros(video.shape, dtype=torch.bfloat16, device=video.device)
        wc = torch.zeros(video.shape, dtype=torch.int32, device=video.device)
        print(f"Process video: {name} | Prompt: {prompt} | Frame: {video.shape[2]} (ori: {orig[0]}; pad: {pad_f}) | Target Resolution: "
              f"{video.shape[3]}, {video.shape[4]} | Chunk Num: {len(items)}")
        for (t0, t1, h0, h1, w0, w1), region in items:
            piece = process_video(pipe, video[:, :, t0:t1, h0:h1, w0:w1], prompt=prompt, noise_step=args.noise_step,
                                  sr_noise_step=args.sr_noise_step, empty_prompt_embedding=emb)
            tiling.stitch(out, wc, piece, region)
        tiling.check_coverage(wc)
        frames_out = prepost.postprocess_frames(out, pad_f, pad_h, pad_w)       # the reference crops pad*4 (ref :731)
        stem = name[:-4] if name.lower().endswith(".npy") else name
        if args.png_save:
            prepost.save_frames_as_png(frames_out, os.path.join(args.output_path, stem))
        else:
            import numpy as np
            np.save(os.path.join(args.output_path, stem + ".npy"), frames_out.cpu().numpy())
        if args.eval_psnr_dir:
            gt = prepost.load_frames(os.path.join(args.eval_psnr_dir, name)).float() / 255
            pr = frames_out.cpu().float() / 255
            mse = ((gt - pr) ** 2).flatten(1).mean(1)
            psnrs[name] = float((10 * torch.log10(1.0 / (mse + 1e-8))).mean())
            print(f"[{name}] PSNR={psnrs[name]:.4f}")
    if psnrs:
        print(f"=== Overall Average PSNR: {sum(psnrs.values()) / len(psnrs):.4f} ===")
    print("All videos processed.")


if __name__ == "__main__":
    main()
