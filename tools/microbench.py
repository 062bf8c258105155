"""Per-kernel timing at the headline (33x720x1280) shapes: TFLOP/s or GB/s per operator, HIP-event timed.
Used to pick what to optimise; results are printed and dumped as JSON."""
import json
import math
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dove_amd import ops  # noqa: E402

BF = torch.bfloat16


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


DATA = "normal"        # --data zeros: all-zero operands draw no switching power in the matrix pipe - the clock stays at its maximum, so the
                       # time on zeros is the kernel's SCHEDULE-limited time and the ratio to the time on N(0,1) data its power give-back


def rnd(*shape):
    return torch.zeros(*shape, device="cuda") if DATA == "zeros" else torch.randn(*shape, device="cuda")


def pack(cout, cin, k):
    w = rnd(cout, cin, *k) * (cin * math.prod(k)) ** -0.5
    return ops.pack_conv(w, torch.zeros(cout, device="cuda"), "cuda")


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="", help="comma list of substrings; run only matching cases")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--data", default="normal", choices=["normal", "zeros"])
    args = ap.parse_args()
    global DATA
    DATA = args.data
    only = [o for o in args.only.split(",") if o]
    want = lambda name: (not only) or any(o in name for o in only)   # noqa: E731
    res = {}
    dev = "cuda"
    global timeit
    _t = timeit
    timeit = lambda fn, iters=args.iters, warm=2: _t(fn, iters=min(iters, args.iters), warm=min(warm, 2))   # noqa: E731

    def conv_case(name, cin, cout, k, T, H, W, direct=False, **kw):
        if not want(name):
            return
        pc = pack(cout, cin, k)
        if direct:                 # the forms without the pack-time weight sums (dove_conv_desc.w_first / w_sub = NULL)
            pc.w_first = pc.w_sub = None
        x = rnd(T, H, W, pc.cin_pad).to(BF)
        y = ops.conv(x, pc, **kw)
        dt = timeit(lambda: ops.conv(x, pc, out=y, **kw))
        flops = 2.0 * y.shape[0] * y.shape[1] * y.shape[2] * cout * cin * math.prod(k)
        res[name] = dict(ms=dt * 1e3, tflops=flops / dt / 1e12)
        print(f"{name:34s} {dt*1e3:9.3f} ms  {flops/dt/1e12:8.1f} TFLOP/s", flush=True)

    conv_case("conv3d 128->128 9x720x1280", 128, 128, (3, 3, 3), 9, 720, 1280)
    conv_case("conv3d 256->256 9x360x640", 256, 256, (3, 3, 3), 9, 360, 640)
    conv_case("conv3d 512->512 5x180x320", 512, 512, (3, 3, 3), 5, 180, 320)
    conv_case("conv3d 256->128 8x720x1280", 256, 128, (3, 3, 3), 8, 720, 1280)
    conv_case("conv3d 3->128 9x720x1280", 3, 128, (3, 3, 3), 9, 720, 1280)
    conv_case("conv3d 128->3 9x720x1280", 128, 3, (3, 3, 3), 9, 720, 1280)
    conv_case("conv2d up 256->256 ->8x720x1280", 256, 256, (3, 3), 8, 360, 640, up=1, pad=(1, 1))
    conv_case("conv2d updirect 256->256 ->8x720x1280", 256, 256, (3, 3), 8, 360, 640, direct=True, up=1, pad=(1, 1))
    conv_case("conv2d down 128 s2 9x720x1280", 128, 128, (3, 3), 9, 720, 1280, stride=2, pad=(0, 0))

    N = 18226
    for name, cin, cout, act in (("linear qkv 3072->9216", 3072, 9216, 0), ("linear out 3072->3072", 3072, 3072, 0),
                                 ("linear ff1 3072->12288 gelu", 3072, 12288, 1), ("linear ff2 12288->3072", 12288, 3072, 0)):
        if not want(name):
            continue
        pc = pack(cout, cin, ())
        x = rnd(N, cin).to(BF)
        y = ops.linear(x, pc, act=act)
        dt = timeit(lambda: ops.linear(x, pc, act=act, out=y))
        fl = 2.0 * N * cin * cout
        res[name] = dict(ms=dt * 1e3, tflops=fl / dt / 1e12)
        print(f"{name:34s} {dt*1e3:9.3f} ms  {fl/dt/1e12:8.1f} TFLOP/s", flush=True)

    heads = 48
    npad = (N + 127) // 128 * 128
    if only and not (want("attention") or want("gn_") or want("ln_mod") or want("qkv_post")):
        return
    Qh = (rnd(heads, npad, 64) * 0.3).to(BF)
    Kh = (rnd(heads, npad, 64) * 0.3).to(BF)
    Vt = rnd(heads, 64, npad).to(BF)
    O = torch.empty(N, heads * 64, device=dev, dtype=BF)
    dt = timeit(lambda: ops.attention(Qh, Kh, Vt, N, npad, heads, O), iters=3, warm=1)
    fl = 4.0 * heads * N * N * 64
    res["attention N=18226 h=48"] = dict(ms=dt * 1e3, tflops=fl / dt / 1e12)
    print(f"{'attention N=18226 h=48':34s} {dt*1e3:9.3f} ms  {fl/dt/1e12:8.1f} TFLOP/s", flush=True)

    # HBM-bound ops
    x = torch.randn(9, 720, 1280, 128, device=dev).to(BF)
    g = torch.ones(128, device=dev)
    st = ops.groupnorm_stats(x, 1e-6)
    dt = timeit(lambda: ops.groupnorm_stats(x, 1e-6))
    res["gn_stats 128ch 9x720x1280"] = dict(ms=dt * 1e3, gbs=x.numel() * 2 / dt / 1e9)
    print(f"{'gn_stats 128ch 9x720x1280':34s} {dt*1e3:9.3f} ms  {x.numel()*2/dt/1e9:8.1f} GB/s", flush=True)
    y = torch.empty_like(x)
    dt = timeit(lambda: ops.groupnorm_apply(x, st, g, g, silu=True, out=y))
    res["gn_apply 128ch 9x720x1280"] = dict(ms=dt * 1e3, gbs=x.numel() * 4 / dt / 1e9)
    print(f"{'gn_apply 128ch 9x720x1280':34s} {dt*1e3:9.3f} ms  {x.numel()*4/dt/1e9:8.1f} GB/s", flush=True)
    yb = torch.randn(3, 90, 160, 256, device=dev).to(BF)
    tm = [0, 1, 1, 1, 1, 2, 2, 2, 2]
    dt = timeit(lambda: ops.groupnorm_apply(x, st, g, g, silu=True, yb=yb, sshift=3, tmap=tm, out=y))
    res["sn_apply 128ch 9x720x1280"] = dict(ms=dt * 1e3, gbs=x.numel() * 4 / dt / 1e9)
    print(f"{'sn_apply 128ch 9x720x1280':34s} {dt*1e3:9.3f} ms  {x.numel()*4/dt/1e9:8.1f} GB/s", flush=True)
    h = torch.randn(N, 3072, device=dev).to(BF)
    g3 = torch.ones(3072, device=dev)
    mod = torch.zeros(2, 2, 3072, device=dev)
    o = torch.empty_like(h)
    dt = timeit(lambda: ops.layernorm_modulate(h, g3, g3, 1e-5, mod, 226, out=o))
    res["ln_mod 18226x3072"] = dict(ms=dt * 1e3, gbs=h.numel() * 4 / dt / 1e9)
    print(f"{'ln_mod 18226x3072':34s} {dt*1e3:9.3f} ms  {h.numel()*4/dt/1e9:8.1f} GB/s", flush=True)
    qkv = torch.randn(N, 9216, device=dev).to(BF)
    g64 = torch.ones(64, device=dev)
    cs = torch.ones(N - 226, 64, device=dev)
    dt = timeit(lambda: ops.qkv_post(qkv, N, npad, heads, 226, g64, g64, g64, g64, cs, cs, 0.18, 1e-6, Qh, Kh, Vt))
    res["qkv_post 18226"] = dict(ms=dt * 1e3, gbs=qkv.numel() * 4 / dt / 1e9)
    print(f"{'qkv_post 18226':34s} {dt*1e3:9.3f} ms  {qkv.numel()*4/dt/1e9:8.1f} GB/s", flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/microbench.json", "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
