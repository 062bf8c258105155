"""One VAE mode (tiled | untiled), ONE stream, for a rocprofv3 --kernel-trace --stats pass: per-kernel time of the reference's tiled
configuration next to the untiled clip (where does the tiled mode lose against its FLOP ratio?).  Also prints the per-class table of the
implicit-GEMM launches (ms, launches, issued TFLOP/s) from the library's own launch profiler.
    python tools/vae_mode_profile.py --mode tiled|untiled [--reps 2]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dove_amd import config, ops, weights
from dove_amd.vae import AutoencoderKLCogVideoX

ap = argparse.ArgumentParser()
ap.add_argument("--mode", default="tiled")
ap.add_argument("--reps", type=int, default=2)
args = ap.parse_args()
dev = torch.device("cuda", 0)
v, t, s = config.default_configs()
vae = AutoencoderKLCogVideoX(v, weights.LazyStateDict(weights.vae_param_shapes(v), 1234, dev), dev, torch.bfloat16)
video = bench.prepare_clip(bench.synth_lr_clip(33, 180, 320, seed=42, device=dev), 4).to(torch.bfloat16)
z = torch.randn(1, 16, 9, 90, 160, device=dev, generator=torch.Generator(device=dev).manual_seed(7)).to(torch.bfloat16)
vae.n_streams = 1
if args.mode == "tiled":
    vae.enable_slicing(); vae.enable_tiling()
    vae.tile_streams = 1
    vae.tile_batch_streams = 1
else:
    vae.disable_tiling()


def run():
    vae.encode(video).latent_dist.parameters
    vae.decode(z, _range01=True).sample


run(); torch.cuda.synchronize()
recs = []
ops.set_profiler(recs)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(args.reps):
    run()
e1.record(); torch.cuda.synchronize()
ops.set_profiler(None)
print(f"[{args.mode}] VAE {e0.elapsed_time(e1) / args.reps:.1f} ms per clip (one stream, per-launch events on)")
by = {}
for key, fa, a0, a1, name, fr in recs:
    k = f"{name}:cin{key[0]}_cout{key[1]}_taps{key[2]}"
    a = by.setdefault(k, [0.0, 0.0, 0])
    a[0] += fr; a[1] += a0.elapsed_time(a1); a[2] += 1
tot = 0.0
for k, a in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:64s} {a[1] / args.reps:8.2f} ms  {a[0] / (a[1] * 1e-3) / 1e12:8.1f} TF issued  {a[0] / args.reps / 1e12:8.2f} TFLOP  {a[2] // args.reps:4d} launches")
    tot += a[1] / args.reps
print("igemm sum", round(tot, 1), "ms")
