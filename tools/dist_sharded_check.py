"""Launched with `python -m torch.distributed.run --nproc-per-node R ...` on a GPU box (backend nccl = RCCL): the fully
sharded single-clip path (halo-exact VAE + Ulysses DiT, dove_amd.dist.process_video_sharded) must be bit-identical to
the single-GPU process_video on every rank.  With R = 1 it still drives every collective of the path through RCCL."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dove_amd import config, dist as ddist  # noqa: E402
from dove_amd.inference import process_video  # noqa: E402
from dove_amd.pipeline import CogVideoXPipeline  # noqa: E402

local = int(os.environ.get("LOCAL_RANK", "0"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
dist.init_process_group("nccl", device_id=dev)
v, t, s = config.small_configs(num_layers=2)
pipe = CogVideoXPipeline.from_config(v, t, s, seed=21, device=dev, init_device=dev)
g = torch.Generator().manual_seed(5)
video = (torch.rand(1, 3, 17, 64, 96, generator=g) * 2 - 1).to(torch.bfloat16).to(dev)
noise = torch.randn(1, 16, 5, 8, 12, generator=g).to(dev)
text = torch.randn(226, t["text_embed_dim"], generator=g).to(torch.bfloat16).to(dev)
ref = process_video(pipe, video, empty_prompt_embedding=text, posterior_noise=noise)
out = ddist.process_video_sharded(pipe, video, empty_prompt_embedding=text, posterior_noise=noise)
torch.cuda.synchronize()
ok = bool(torch.equal(ref, out))
print(f"[rank {dist.get_rank()}/{dist.get_world_size()}] process_video_sharded == process_video: {ok}  shape {tuple(out.shape)}", flush=True)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
