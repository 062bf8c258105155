"""What does the headline clip cost when nothing toggles?  The whole operator (33x720x1280, 42 layers) once with the bench's random-init
weights and synthetic clip, once with EVERY weight, the clip and the posterior noise set to zero: same launches, same addresses, same
instruction streams - but the matrix pipe switches no bits, so the shader clock stays at its maximum.  The second time is what the
code's schedule allows; the ratio is the clock the chip takes back on real operands (see profiles/r04_zero_operands.log for the same
A/B per kernel)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dove_amd import config, ops, weights as W  # noqa: E402
from dove_amd.inference import process_video  # noqa: E402
from dove_amd.pipeline import CogVideoXPipeline  # noqa: E402
from dove_amd.scheduler import CogVideoXDPMScheduler  # noqa: E402
from dove_amd.transformer import CogVideoXTransformer3DModel  # noqa: E402
from dove_amd.vae import AutoencoderKLCogVideoX  # noqa: E402


def build(zero, dev):
    v, t, s = config.default_configs()
    vs, ts = W.vae_param_shapes(v), W.dit_param_shapes(t)
    sc_v = {k: 0.0 for k in vs} if zero else None
    sc_t = {k: 0.0 for k in ts} if zero else None
    vae = AutoencoderKLCogVideoX(v, W.LazyStateDict(vs, 1234, dev, scale=sc_v), dev, torch.bfloat16)
    tr = CogVideoXTransformer3DModel(t, W.LazyStateDict(ts, 1234, dev, scale=sc_t), dev, torch.bfloat16, "bf16", "bf16")
    return CogVideoXPipeline(vae, tr, CogVideoXDPMScheduler(**s))


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    dev = torch.device("cuda", 0)
    from safetensors.torch import load_file
    text = load_file(os.path.join(ROOT, "tests", "golden", "empty_prompt_embedding.safetensors"))["prompt_embedding"]
    F, H, Wd = 33, 720, 1280
    res = {}
    for zero in (False, True, False, True):
        pipe = build(zero, dev)
        video = bench.prepare_clip(bench.synth_lr_clip(F, H // 4, Wd // 4, seed=42, device=dev), 4)
        noise = torch.randn(1, 16, 9, H // 8, Wd // 8, device=dev, generator=torch.Generator(device=dev).manual_seed(7))
        txt = text
        if zero:
            video, noise, txt = torch.zeros_like(video), torch.zeros_like(noise), torch.zeros_like(text)
        for _ in range(2):
            out = process_video(pipe, video, empty_prompt_embedding=txt, posterior_noise=noise)
        rec = []
        ops.set_profiler(rec)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = process_video(pipe, video, empty_prompt_embedding=txt, posterior_noise=noise)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        ops.set_profiler(None)
        assert torch.isfinite(out.float()).all()
        per = {}
        for key, fl_alg, e0, e1, name, fl in rec:
            a = per.setdefault(name, [0.0, 0.0])
            a[0] += e0.elapsed_time(e1) / steps
            a[1] += fl / steps
        tag = "zeros" if zero else "N(0, s) weights, synthetic clip"
        print(f"# {tag}: {dt * 1e3:8.1f} ms per clip = {F / dt:6.2f} frames/s", flush=True)
        for name, (ms, fl) in sorted(per.items(), key=lambda kv: -kv[1][0]):
            print(f"     {name:28s} {ms:8.1f} ms   {fl / ms / 1e9:8.1f} TFLOP/s issued")
        res.setdefault(zero, []).append(dt)
        del pipe, out
        torch.cuda.empty_cache()
    a, z = min(res[False]), min(res[True])
    print(f"# schedule-limited clip time {z * 1e3:.1f} ms ({F / z:.2f} frames/s) vs {a * 1e3:.1f} ms on real operands: x {a / z:.3f}")


if __name__ == "__main__":
    main()
