"""BASELINE configs[2] from the C graph calls at its REAL size: 33x720x1280, 8 contexts = 8 ranks as threads on one GPU (buffering mailbox
transport), full-width VAE + 2-layer DiT.  dove_sr_clip on every rank; the ranks' frames must equal the single-context clip bit for bit."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from dove_amd import config, weights
from dove_amd.graph import GraphContext
from dove_amd.pipeline import CogVideoXPipeline
import test_graph_gpu as tg

R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
BF = torch.bfloat16
v, t, s = config.small_configs(num_layers=2)
dev = torch.device("cuda", 0)
wv = weights.LazyStateDict(weights.vae_param_shapes(v), 31, dev)
wt = weights.LazyStateDict(weights.dit_param_shapes(t), 31, dev)
pipe = CogVideoXPipeline.from_config(v, t, s, seed=31, device=dev, init_device=dev)
g = torch.Generator().manual_seed(208)
F, H, W = 33, 720, 1280
video = (torch.rand(3, F, H, W, generator=g) * 2 - 1).to(BF).cuda()
T = 1 + (F - 1) // 4
noise = torch.randn(16, T, H // 8, W // 8, generator=g).cuda()
text = (torch.randn(226, 4096, generator=g) * 0.15).to(BF).cuda()
rope = tg.rope_for(pipe, T + T % 2, H // 8, W // 8)
tproj = pipe.transformer.timestep_projection(399)
sa, s1 = pipe.scheduler._coeffs(torch.tensor([399]), BF)
ctx0 = GraphContext(v, t, wv, wt, dev)
ref = ctx0.sr_clip(video, noise, text, 399, sa, s1, rope=rope, timestep_proj=tproj)
torch.cuda.synchronize()
del ctx0, pipe
torch.cuda.empty_cache()
box = tg._Mailbox()
ctxs = []
WS = 24 << 30       # per-rank arena LENT by the caller: dove_workspace_bytes sizes for retaining every conv's previous input of a whole
                    # 9-frame batch (47 GB) - 8 ranks on ONE GPU would not fit; a rank's 4-5 frame piece runs in 24 GB (caches are then copied)
for r in range(R):
    c = GraphContext(v, t, wv, wt, dev)
    c.set_workspace(WS, buffer=torch.empty(WS, dtype=torch.uint8, device=dev))
    c.comm_init_custom(r, R, *box.fns(r))
    ctxs.append(c)
t0 = time.time()
clips = tg._run_ranks(ctxs, lambda r, c: (c.shard_frames(1, T), c.sr_clip(video, noise, text, 399, sa, s1, rope=rope, timestep_proj=tproj)))
torch.cuda.synchronize()
dt = time.time() - t0
clip = torch.full_like(ref, float("nan"))
for (first, count), o in clips:
    clip[:, first:first + count] = o[:, first:first + count]
halo = sorted({n for _, _, n in box.log}, reverse=True)[:4]
print(f"C level, {R} ranks as threads, 33x720x1280: frames per rank {[cnt for (_, cnt), _ in clips]}, bit-identical to one context: {bool(torch.equal(clip, ref))}, "
      f"{len(box.log)} messages ({sum(n for _, _, n in box.log) / 1e9:.1f} GB through the mailbox; largest {halo}), {dt:.1f} s, "
      f"arena high water per rank {[round(c.workspace_high_water() / 1e9, 1) for c in ctxs]} GB of {WS >> 30} lent")
assert torch.equal(clip, ref)
