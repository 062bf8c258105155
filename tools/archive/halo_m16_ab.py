"""The dominant conv kernel on the other MFMA shape (conv3x3_halo4x_kernel<..., kM16>: v_mfma_f32_16x16x32_bf16), timing
library only: DOVE_HALO_M16=1 selects it per call.  (1) every form of the kernel against the product form and against a torch fp32
restatement of the operator (tests/emu_ops.py) on small cases - cache / no cache (w_first), residual, fused GroupNorm statistics,
batched instances, the sub-pixel upsample form, the upsample-in-addressing form; (2) headline-shape timings back to back; (3) the full-size
VAE encode + decode, alternating, with the rms-relative difference of the results.
    python tools/halo_m16_ab.py [check|time|vae]..."""
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from dove_amd import lib  # noqa: E402
lib.use_timing_build()
from dove_amd import ops  # noqa: E402
import emu_ops  # noqa: E402

dev = torch.device("cuda", 0)
BF = torch.bfloat16
what = sys.argv[1:] or ["check", "time", "vae"]


def m16(on):
    os.environ["DOVE_HALO_M16"] = "1" if on else "0"


def rr(a, b):
    return float(((a.float() - b.float()) ** 2).mean().sqrt() / ((b.float() ** 2).mean().sqrt() + 1e-30))


if "check" in what:
    g = torch.Generator(device="cpu").manual_seed(5)
    cases = [  # name, cin, cout, k, T, H, W, kwargs-flags
        ("3x3x3 128->128 cache+resid+gn", 128, 128, (3, 3, 3), 3, 48, 64, dict(cache=True, resid=True, gn=True)),
        ("3x3x3 128->128 no cache (w_first)", 128, 128, (3, 3, 3), 4, 32, 96, dict(gn=True)),
        ("3x3x3 256->256 ragged 37x70", 256, 256, (3, 3, 3), 2, 37, 70, dict(cache=True, gn=True)),
        ("3x3x3 512->512", 512, 512, (3, 3, 3), 2, 24, 40, dict(cache=True, resid=True, gn=True)),
        ("3x3 256->128 kt=1", 256, 128, (3, 3), 3, 32, 64, dict()),
        ("3x3x3 128->128 nb=3", 128, 128, (3, 3, 3), 2, 32, 32, dict(nb=3, gn=True)),
        ("up 3x3 256->256 sub-pixel 20x40 -> 40x80", 256, 256, (3, 3), 3, 20, 40, dict(up=True, gn=True)),
        ("up 3x3 128->128 in-addressing 8x16 -> 16x32", 128, 128, (3, 3), 2, 8, 16, dict(up=True)),
    ]
    worst = 0.0
    for name, cin, cout, k, T, H, W, fl in cases:
        w = torch.randn(cout, cin, *k, generator=g) * (cin * math.prod(k)) ** -0.5
        b = torch.randn(cout, generator=g) * 0.1
        pc, pe = ops.pack_conv(w, b, dev), emu_ops.pack_conv(w, b)
        nb = fl.get("nb", 1)
        x = torch.randn(nb * T, H, W, cin, generator=g).to(BF)
        kw, kwe = {}, {}
        if fl.get("up"):
            kw = kwe = dict(up=1, pad=(1, 1))
            kw, kwe = dict(kw), dict(kwe)
        Ho, Wo = (2 * H, 2 * W) if fl.get("up") else (H, W)
        if fl.get("cache"):
            c = torch.randn(len(k) == 3 and k[0] - 1 or 0, H, W, cin, generator=g).to(BF)
            kw["cache"], kwe["cache"] = c.to(dev), c
        if fl.get("resid"):
            r = torch.randn(nb * T, Ho, Wo, cout, generator=g).to(BF)
            kw["resid"], kwe["resid"] = r.to(dev), r
        if fl.get("gn"):
            kw["gn_eps"] = kwe["gn_eps"] = 1e-6
        if nb > 1:
            kw["nb"] = kwe["nb"] = nb
        ref = emu_ops.conv(x, pe, **kwe)
        outs = {}
        for on in (False, True):
            m16(on)
            y = ops.conv(x.to(dev), pc, **kw)
            torch.cuda.synchronize()
            outs[on] = (y.cpu(), getattr(y, "gn_stats", None))
        e0, e1, d = rr(outs[False][0], ref), rr(outs[True][0], ref), rr(outs[True][0], outs[False][0])
        st = ""
        if outs[True][1] is not None:
            s0, s1 = outs[False][1][0].cpu().float(), outs[True][1][0].cpu().float()
            rs = emu_ops.groupnorm_stats(ref, 1e-6, nb=nb).float()          # (mean, rstd) of the stored bf16 values
            st = f"  GN stats max |d|: between the two {float((s1 - s0).abs().max()):.2e}, 16x16x32 vs a pass over its own output {float((s1 - emu_ops.groupnorm_stats(outs[True][0], 1e-6, nb=nb).float()).abs().max()):.2e}"
        worst = max(worst, e1 / max(e0, 1e-9))
        print(f"{name:46s} rms-rel vs fp32 emu: 32x32x16 {e0:.3e}  16x16x32 {e1:.3e}   between the two {d:.3e}{st}", flush=True)
        assert e1 < 1.5 * e0 + 1e-4, name
    print(f"check passed: worst error ratio 16x16x32 / 32x32x16 = {worst:.3f}", flush=True)

if "time" in what:
    def timeit(fn, iters=10):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / iters

    shapes = [("conv3d 128->128 9x720x1280", 128, 128, (3, 3, 3), 9, 720, 1280, {}),
              ("conv3d 256->256 9x360x640", 256, 256, (3, 3, 3), 9, 360, 640, {}),
              ("conv3d 512->512 5x180x320", 512, 512, (3, 3, 3), 5, 180, 320, {}),
              ("conv2d up 256->256 -> 8x720x1280", 256, 256, (3, 3), 8, 360, 640, dict(up=1, pad=(1, 1)))]
    for name, cin, cout, k, T, H, W, kw in shapes:
        w = torch.randn(cout, cin, *k, device=dev) * (cin * math.prod(k)) ** -0.5
        pc = ops.pack_conv(w, torch.zeros(cout, device=dev), dev)
        x = torch.randn(T, H, W, cin, device=dev).to(BF)
        y = ops.conv(x, pc, **kw)
        fl = 2.0 * y.shape[0] * y.shape[1] * y.shape[2] * cout * cin * math.prod(k)
        res = {}
        for rnd in range(3):
            for on in (False, True):
                m16(on)
                res.setdefault(on, []).append(timeit(lambda: ops.conv(x, pc, out=y, **kw)))
        t0, t1 = sorted(res[False])[1], sorted(res[True])[1]
        print(f"{name:36s} 32x32x16 {t0:7.3f} ms ({fl / t0 / 1e9:7.1f} TFLOP/s)   16x16x32 {t1:7.3f} ms ({fl / t1 / 1e9:7.1f} TFLOP/s)   x{t1 / t0:.3f}", flush=True)

if "vae" in what:
    import bench
    from dove_amd import config, weights
    from dove_amd.vae import AutoencoderKLCogVideoX
    v, t, s = config.default_configs()
    vae = AutoencoderKLCogVideoX(v, weights.LazyStateDict(weights.vae_param_shapes(v), 1234, dev), dev, BF)
    video = bench.prepare_clip(bench.synth_lr_clip(33, 180, 320, seed=42, device=dev), 4).to(BF)
    z = torch.randn(1, 16, 9, 90, 160, device=dev, generator=torch.Generator(device=dev).manual_seed(7)).to(BF)

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            o = fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return sorted(ts)[len(ts) // 2], o

    res = {}
    for rnd in range(2):
        for on in (False, True):
            m16(on)
            e, m = timed(lambda: vae.encode(video).latent_dist.parameters)
            d, o = timed(lambda: vae.decode(z, _range01=False).sample)
            res.setdefault(on, []).append((e, d, m, o))
    for on in (False, True):
        print(f"{'16x16x32' if on else '32x32x16'}: encode {min(r[0] for r in res[on]):7.2f} ms  decode {min(r[1] for r in res[on]):7.2f} ms  "
              f"VAE {min(r[0] + r[1] for r in res[on]):7.2f} ms", flush=True)
    m0, o0, m1, o1 = res[False][0][2], res[False][0][3], res[True][0][2], res[True][0][3]
    print(f"16x16x32 vs 32x32x16, rms-rel: moments {rr(m1, m0):.3e}, decoded {rr(o1, o0):.3e}; finite: {bool(torch.isfinite(m1.float()).all() and torch.isfinite(o1.float()).all())}")
    print(f"saving {min(r[0] + r[1] for r in res[False]) - min(r[0] + r[1] for r in res[True]):.2f} ms per clip")
