"""The COST SIDE of a consumer-side GroupNorm + SiLU fusion, measured inside the product conv walk: the timing library's
conv3x3_halo4x_kernel<..., kFill> is the product kernel plus, in steps 2..7 of every (frame tap, channel chunk) group, the fusion's
instruction stream on the halo rounds that have just landed - 2 ds_read_b128, the normalise + SiLU + pack mix on their 16 elements per thread
(124 VALU: unpack, fma with per-channel scale / shift in registers, mul, exp, add, rcp, mul, cvt; eight independent chains), 2 ds_write_b128
of the ORIGINAL bytes, so the result is unchanged (asserted).  What it leaves out only makes the real thing dearer: the per-round reads of
the scale / shift table (a lane's channel chunk changes every round), the zero-padding mask (silu(shift) != 0 outside the image), the
per-frame tables of the cached frames, SpatialNorm's two gathers.  Against it: the time of the gn_apply (+ SiLU) pass the fusion would retire.
    python tools/gn_fusion_cost.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dove_amd import lib  # noqa: E402
lib.use_timing_build()
from dove_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
BF = torch.bfloat16


def fill(on):
    os.environ["DOVE_HALO_FILL"] = "1" if on else "0"


def timed(fn, reps=9):
    for _ in range(2):
        fn()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


for name, cin, cout, T, H, W, launches in (("128 -> 128 x27 at 8 x 720 x 1280 (52 launches per clip, 246 ms: the class the verdict names)", 128, 128, 8, 720, 1280, 52),
                                           ("256 -> 256 x27 at 8 x 360 x 640 (104 launches per clip, 206 ms)", 256, 256, 8, 360, 640, 104)):
    g = torch.Generator(device=dev).manual_seed(3)
    pc = ops.pack_conv(torch.randn(cout, cin, 3, 3, 3) * (cin * 27) ** -0.5, torch.randn(cout) * 0.1, dev)
    x = torch.nn.functional.silu(torch.randn(T, H, W, cin, device=dev, generator=g)).to(BF)       # what the conv really reads: SiLU outputs
    cache = torch.nn.functional.silu(torch.randn(2, H, W, cin, device=dev, generator=g)).to(BF)
    resid = torch.randn(T, H, W, cout, device=dev, generator=g).to(BF)
    fill(False); y0 = ops.conv(x, pc, cache=cache, resid=resid, gn_eps=1e-6)
    fill(True); y1 = ops.conv(x, pc, cache=cache, resid=resid, gn_eps=1e-6)
    torch.cuda.synchronize()
    same = bool(torch.equal(y0, y1))
    t = {False: [], True: []}
    for _ in range(3):                                              # alternate: the box's clock drifts with temperature
        for on in (False, True):
            fill(on)
            t[on].append(timed(lambda: ops.conv(x, pc, cache=cache, resid=resid, gn_eps=1e-6)))
    fill(False)
    t0, t1 = sorted(t[False])[1], sorted(t[True])[1]
    # the pass the fusion retires: GroupNorm apply + SiLU over the conv's INPUT tensor (statistics are already fused into the producer)
    raw = torch.randn(T, H, W, cin, device=dev, generator=g).to(BF)
    st = ops.groupnorm_stats(raw, 1e-6)
    gam, bet = torch.ones(cin, device=dev), torch.zeros(cin, device=dev)
    tg = timed(lambda: ops.groupnorm_apply(raw, st, gam, bet, silu=True))
    print(f"{name}\n   product walk {t0:.3f} ms, with the fusion's instruction stream {t1:.3f} ms (+{100 * (t1 / t0 - 1):.1f} %, output bit-identical: {same}); "
          f"gn_apply + SiLU on its input {tg:.3f} ms\n   per clip: +{(t1 - t0) * launches:.1f} ms of conv time against {tg * launches:.1f} ms of gn_apply retired "
          f"-> net {(t1 - t0 - tg) * launches:+.1f} ms")
