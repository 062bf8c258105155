"""Whole-operator A/B of a TIMING-build switch on the headline clip: process_video alternately with <VAR> = A and = B (variables the
timing library reads per call).  Usage: python tools/e2e_env_ab.py <VAR> <A> <B> [rounds]
  DOVE_IGEMM_ABLATE 16 0   gemm8p nontemporal output stores forced off vs the product rule
  DOVE_GEMM8P 0 1          gemm4x (round 2's GEMM) vs gemm8p
  DOVE_ATTN_BOUND 0 1      attention with the running maximum vs the per-head score bound
  DOVE_IGEMM_ABLATE 64 0   conv3x3_halo4x epilogue without / with the early slice write"""
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dove_amd import lib as _L  # noqa: E402

_L.use_timing_build()
import bench  # noqa: E402
from dove_amd import config  # noqa: E402
from dove_amd.inference import process_video  # noqa: E402
from dove_amd.pipeline import CogVideoXPipeline  # noqa: E402
from safetensors.torch import load_file  # noqa: E402

VAR, A, B = sys.argv[1], sys.argv[2], sys.argv[3]
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 4
dev = torch.device("cuda", 0)
text = load_file(os.path.join(ROOT, "tests", "golden", "empty_prompt_embedding.safetensors"))["prompt_embedding"]
v, t, s = config.default_configs()
pipe = CogVideoXPipeline.from_config(v, t, s, seed=1234, device=dev, init_device=dev)
video = bench.prepare_clip(bench.synth_lr_clip(33, 180, 320, seed=42, device=dev), 4)
noise = torch.randn(1, 16, 9, 90, 160, device=dev, generator=torch.Generator(device=dev).manual_seed(7))
ts, outs = {A: [], B: []}, {}
for rnd in range(rounds + 1):
    for k in (A, B):
        os.environ[VAR] = k
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = process_video(pipe, video, empty_prompt_embedding=text, posterior_noise=noise)
        torch.cuda.synchronize()
        if rnd:
            ts[k].append(time.perf_counter() - t0)
        outs[k] = out
ma, mb = statistics.median(ts[A]), statistics.median(ts[B])
print(f"{VAR}={A}: {ma * 1e3:.1f} ms per clip | ={B}: {mb * 1e3:.1f} ms per clip ({(ma / mb - 1) * 100:+.2f} %)   outputs equal: {bool(torch.equal(outs[A], outs[B]))}", flush=True)
