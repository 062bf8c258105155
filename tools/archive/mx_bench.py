"""MXFP8 vs bf16 on the DiT linear shapes of the headline clip (M = 18226 tokens): interleaved rounds, median TFLOP/s."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dove_amd import ops  # noqa: E402

M = 18226
g = torch.Generator(device="cuda").manual_seed(0)
for name, K, N, act in (("qkv", 3072, 9216, 0), ("out", 3072, 3072, 0), ("ff1", 3072, 12288, 1), ("ff2", 12288, 3072, 0)):
    x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = torch.randn(N, K, device="cuda", generator=g) * K ** -0.5
    b = torch.zeros(N, device="cuda")
    pc, pm = ops.pack_conv(w, b, "cuda"), ops.pack_linear_mx(w, b, "cuda")
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    xq = ops.mx_quant(x)
    fl = 2.0 * M * K * N
    t = {"bf16": [], "mxfp8": [], "quant": []}
    for rnd in range(4):
        for kind in t:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                if kind == "bf16":
                    ops.linear(x, pc, act=act, out=out)
                elif kind == "mxfp8":
                    ops.linear_mx(xq, pm, act=act, out=out)
                else:
                    ops.mx_quant(x)
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                t[kind].append(e0.elapsed_time(e1) / 5)
    bf, mx, qt = (statistics.median(t[k]) for k in ("bf16", "mxfp8", "quant"))
    print(f"{name:4s} K={K:5d} N={N:5d}: bf16 {bf:6.3f} ms {fl / bf / 1e9:7.1f} TF | mxfp8 {mx:6.3f} ms {fl / mx / 1e9:7.1f} TF | "
          f"activation quant {qt:6.3f} ms ({M * K * 3 / qt / 1e6:6.1f} GB/s) | speed-up incl. quant {bf / (mx + qt):.2f}x", flush=True)
