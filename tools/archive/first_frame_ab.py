"""A/B of the two MAC-saving weight transforms of the VAE convs, within one process, alternating: the first-frame temporal-tap reduction
(dove_conv_desc.w_first: cache-less first frames with pre-summed temporal weights) and the sub-pixel form of the upsample-fused convs
(dove_conv_desc.w_sub: per output phase a 2x2 conv on the low-res input).  Also the difference each makes against the un-summed form.
    python tools/first_frame_ab.py [first|sub]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from dove_amd import config, ops, weights
from dove_amd.vae import AutoencoderKLCogVideoX

dev = torch.device("cuda", 0)
v, t, s = config.default_configs()
vae = AutoencoderKLCogVideoX(v, weights.LazyStateDict(weights.vae_param_shapes(v), 1234, dev), dev, torch.bfloat16)
video = bench.prepare_clip(bench.synth_lr_clip(33, 180, 320, seed=42, device=dev), 4).to(torch.bfloat16)
z = torch.randn(1, 16, 9, 90, 160, device=dev, generator=torch.Generator(device=dev).manual_seed(7)).to(torch.bfloat16)
FIELD = "w_sub" if (len(sys.argv) > 1 and sys.argv[1] == "sub") else "w_first"
saved = {k: getattr(pc, FIELD) for k, pc in vae.pc.items()}

def mode(on):
    for k, pc in vae.pc.items():
        setattr(pc, FIELD, saved[k] if on else None)

def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); o = fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2], o

res = {}
for rnd in range(2):
    for on in (False, True):
        mode(on)
        e, m = timed(lambda: vae.encode(video).latent_dist.parameters)
        d, o = timed(lambda: vae.decode(z, _range01=False).sample)
        res.setdefault(on, []).append((e, d, m, o))
for on in (False, True):
    print(f"{FIELD} {'ON ' if on else 'off'}: encode {min(r[0] for r in res[on]):7.2f} ms  decode {min(r[1] for r in res[on]):7.2f} ms  "
          f"VAE {min(r[0] + r[1] for r in res[on]):7.2f} ms")
def rr(a, b): return float(((a.float() - b.float()) ** 2).mean().sqrt() / (b.float() ** 2).mean().sqrt())
m0, o0, m1, o1 = res[False][0][2], res[False][0][3], res[True][0][2], res[True][0][3]
print(f"summed vs un-summed weights ({FIELD}), rms-rel: moments {rr(m1, m0):.3e} (first latent frame {rr(m1[:, :, :1], m0[:, :, :1]):.3e}, frames 3.. {rr(m1[:, :, 3:], m0[:, :, 3:]):.3e}), "
      f"decoded {rr(o1, o0):.3e} (first 2 frames {rr(o1[:, :, :2], o0[:, :, :2]):.3e}, frames 9.. {rr(o1[:, :, 9:], o0[:, :, 9:]):.3e})")
print(f"saving {min(r[0] + r[1] for r in res[False]) - min(r[0] + r[1] for r in res[True]):.2f} ms per clip")
