import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from dove_amd import config
from dove_amd.inference import process_video
from dove_amd.pipeline import CogVideoXPipeline
from safetensors.torch import load_file
dev = torch.device("cuda", 0)
text = load_file(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "empty_prompt_embedding.safetensors"))["prompt_embedding"]
v, t, s = config.default_configs()
pipe = CogVideoXPipeline.from_config(v, t, s, seed=1234, device=dev, init_device=dev)
video = bench.prepare_clip(bench.synth_lr_clip(33, 180, 320, seed=42, device=dev), 4).to(torch.bfloat16)
def run():
    torch.cuda.synchronize(); t0 = time.time()
    o = process_video(pipe, video, sr_noise_step=399, empty_prompt_embedding=text, generator=torch.Generator(device=dev).manual_seed(7))
    torch.cuda.synchronize(); return o, time.time() - t0
a, ta = run(); a, ta = run()
pipe.vae.enable_slicing(); pipe.vae.enable_tiling()
b, tb = run(); b, tb = run()
mse = ((a.float() - b.float()) ** 2).mean()
print("untiled %.3f s, tiled (--is_vae_st) %.3f s, finite %s, PSNR tiled vs untiled %.2f dB" % (ta, tb, bool(torch.isfinite(b.float()).all()), float(10 * torch.log10(1.0 / (mse + 1e-8)))))
