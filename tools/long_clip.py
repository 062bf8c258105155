"""BASELINE.json configs[3]-style long clip through the reference's chunk loop (ref inference_script.py:682-729) on ONE
MI355X: synthetic LR 129x270x480 -> script rules (pad 272, x4) -> 129x1088x1920, `--chunk_len 33 --overlap_t 8` ->
chunks [(0,33),(25,58),(50,83),(75,129)] (the last one is 54 frames), each a full process_video call, stitched with the
reference's keep-region rule.  Prints per-chunk and total time (device-resident clip, output stitched on the host like
the reference) as one JSON line.  Not the headline metric; a record that the path runs at this size."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dove_amd import config, tiling  # noqa: E402
from dove_amd.inference import process_video  # noqa: E402
from dove_amd.pipeline import CogVideoXPipeline  # noqa: E402

F, H, W = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (129, 272, 480)))
dev = torch.device("cuda", 0)
from safetensors.torch import load_file  # noqa: E402
text = load_file(os.path.join(ROOT, "tests", "golden", "empty_prompt_embedding.safetensors"))["prompt_embedding"]
v, t, s = config.default_configs()
pipe = CogVideoXPipeline.from_config(v, t, s, seed=1234, device=dev, init_device=dev)
video = bench.prepare_clip(bench.synth_lr_clip(F, H, W, seed=42, device=dev), 4).to(torch.bfloat16)
items = tiling.plan(video.shape, 33, 8, (0, 0), (32, 32))
out = torch.zeros(video.shape, dtype=torch.bfloat16, device=dev)
wc = torch.zeros(video.shape, dtype=torch.int32, device=dev)
times = []
torch.cuda.synchronize()
t_all = time.time()
for (t0, t1, h0, h1, w0, w1), region in items:
    torch.cuda.synchronize()
    tc = time.time()
    piece = process_video(pipe, video[:, :, t0:t1, h0:h1, w0:w1], sr_noise_step=399, empty_prompt_embedding=text,
                          generator=torch.Generator(device=dev).manual_seed(7))
    tiling.stitch(out, wc, piece.to(out.dtype), region)
    torch.cuda.synchronize()
    times.append({"frames": [t0, t1], "s": round(time.time() - tc, 3)})
tiling.check_coverage(wc)
total = time.time() - t_all
print(json.dumps({"clip": [F, H * 4, W * 4], "chunks": times, "total_s": round(total, 3), "frames_per_s": round(F / total, 3),
                  "finite": bool(torch.isfinite(out.float()).all()), "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2**30, 1)}))
