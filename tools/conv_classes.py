"""Every implicit-GEMM class of one headline clip (kernel, Cin, Cout, taps): ms per clip, TFLOP/s on the issued work, launches - the full table
behind bench.py's `top_classes`."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dove_amd import config, ops
from dove_amd.inference import process_video
from dove_amd.pipeline import CogVideoXPipeline
from safetensors.torch import load_file
dev = torch.device("cuda", 0)
text = load_file(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "empty_prompt_embedding.safetensors"))["prompt_embedding"]
v, t, s = config.default_configs()
pipe = CogVideoXPipeline.from_config(v, t, s, seed=1234, device=dev, init_device=dev)
video = bench.prepare_clip(bench.synth_lr_clip(33, 180, 320, seed=42, device=dev), 4)
noise = torch.randn(1, 16, 9, 90, 160, device=dev, generator=torch.Generator(device=dev).manual_seed(7))
process_video(pipe, video, empty_prompt_embedding=text, posterior_noise=noise)
recs = []
ops.set_profiler(recs)
for _ in range(2):
    process_video(pipe, video, empty_prompt_embedding=text, posterior_noise=noise)
torch.cuda.synchronize()
ops.set_profiler(None)
by = {}
for key, fa, e0, e1, name, fr in recs:
    k = f"{name}:cin{key[0]}_cout{key[1]}_taps{key[2]}"
    a = by.setdefault(k, [0.0, 0.0, 0])
    a[0] += fr; a[1] += e0.elapsed_time(e1); a[2] += 1
tot = sum(a[1] for a in by.values()) / 2
for k, a in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:60s} {a[1] / 2:8.2f} ms  {a[0] / (a[1] * 1e-3) / 1e12:8.1f} TF  {a[2] // 2:4d} launches")
print("sum", tot)
