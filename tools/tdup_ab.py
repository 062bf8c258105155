"""A/B of the frame-pair temporal weight sums (dove_conv_desc.tdup / w_pair: the first causal conv behind Upsample3D's time doubling runs two
temporal groups per frame), within one process, alternating, on the full-size decoder (9 x 90 x 160 latents -> 33 x 720 x 1280): decode time with
the pair sums handed over and without (w_pair = None: the declaration is dropped, three groups per frame), the difference it makes to the decoded
clip, and the two launches' own durations.
    python tools/tdup_ab.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dove_amd import config, ops, weights
from dove_amd.vae import AutoencoderKLCogVideoX

dev = torch.device("cuda", 0)
v, t, s = config.default_configs()
vae = AutoencoderKLCogVideoX(v, weights.LazyStateDict(weights.vae_param_shapes(v), 1234, dev), dev, torch.bfloat16)
z = torch.randn(1, 16, 9, 90, 160, device=dev, generator=torch.Generator(device=dev).manual_seed(7)).to(torch.bfloat16)
saved = {k: pc.w_pair for k, pc in vae.pc.items()}
print("convs with pair sums:", [k for k, w in saved.items() if w is not None])

def mode(on):
    for k, pc in vae.pc.items():
        pc.w_pair = saved[k] if on else None

def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); o = fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2], o

res = {}
for rnd in range(3):
    for on in (False, True):
        mode(on)
        d, o = timed(lambda: vae.decode(z, _range01=False).sample)
        res.setdefault(on, []).append((d, o))
for on in (False, True):
    recs = []
    mode(on)
    ops.set_profiler(recs)
    vae.decode(z)
    torch.cuda.synchronize()
    ops.set_profiler(None)
    sel = [r for r in recs if r[0] in ((512, 256, 27), (256, 256, 27)) and r[4] == "conv3x3_halo4x_kernel"]
    by = {}
    for key, fl_alg, e0, e1, name, fl in recs:
        if key in ((512, 256, 27),):
            by.setdefault(key, []).append((e0.elapsed_time(e1), fl / max(fl_alg, 1)))
    print(f"pair sums {'ON ' if on else 'off'}: decode {min(r[0] for r in res[on]):7.2f} ms; 512->256 3x3x3 launches (ms, issued / algorithmic): "
          + ", ".join(f"{a:.3f}/{b:.2f}" for a, b in by.get((512, 256, 27), [])))
def rr(a, b): return float(((a.float() - b.float()) ** 2).mean().sqrt() / (b.float() ** 2).mean().sqrt())
o0, o1 = res[False][0][1], res[True][0][1]
print(f"pair sums vs per-tap weights, decoded clip rms-rel {rr(o1, o0):.3e}; finite {bool(torch.isfinite(o1.float()).all())}")
print(f"saving {min(r[0] for r in res[False]) - min(r[0] for r in res[True]):.2f} ms per clip")
