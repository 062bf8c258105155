"""gemm8p on the 16x16x32 MFMA shape (gemm8p_kernel<..., kM16>, timing library only: DOVE_GEMM_M16=1 per call) against the product's 32x32x16
phases: bit-identity on the DiT's linear forms (plain, GELU, gated residual; M with and without a row tail) and back-to-back timings at
N = 18 226 tokens.      python tools/gemm_m16_ab.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dove_amd import lib  # noqa: E402
lib.use_timing_build()
from dove_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
BF = torch.bfloat16


def m16(on):
    os.environ["DOVE_GEMM_M16"] = "1" if on else "0"


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


g = torch.Generator(device="cuda").manual_seed(3)
for name, cin, cout, act, gated, N in (("qkv 3072->9216", 3072, 9216, 0, False, 18226), ("out 3072->3072 gated", 3072, 3072, 0, True, 18226),
                                       ("ff1 3072->12288 gelu", 3072, 12288, 1, False, 18226), ("ff2 12288->3072 gated", 12288, 3072, 0, True, 18226),
                                       ("out 3072->3072 gated, M = 4608 (no tail)", 3072, 3072, 0, True, 4608)):
    w = torch.randn(cout, cin, device=dev, generator=g) * cin ** -0.5
    pc = ops.pack_conv(w, torch.randn(cout, device=dev, generator=g) * 0.1, dev)
    x = torch.randn(N, cin, device=dev, generator=g).to(BF)
    kw = dict(act=act)
    if gated:
        kw.update(resid=torch.randn(N, cout, device=dev, generator=g).to(BF), gate=torch.randn(2, cout, device=dev, generator=g), gate_split=226)
    outs, ts = {}, {}
    for rnd in range(3):
        for on in (False, True):
            m16(on)
            y = ops.linear(x, pc, **kw)
            torch.cuda.synchronize()
            outs[on] = y.clone()
            yb = torch.empty_like(y)
            ts.setdefault(on, []).append(timeit(lambda: ops.linear(x, pc, out=yb, **kw)))
    same = bool((outs[True].view(torch.int16) == outs[False].view(torch.int16)).all())
    t0, t1 = sorted(ts[False])[1], sorted(ts[True])[1]
    fl = 2.0 * N * cin * cout
    print(f"{name:42s} bit-identical: {same}   32x32x16 {t0:6.3f} ms ({fl / t0 / 1e9:6.1f} TF)   16x16x32 {t1:6.3f} ms ({fl / t1 / 1e9:6.1f} TF)   x{t1 / t0:.3f}", flush=True)
