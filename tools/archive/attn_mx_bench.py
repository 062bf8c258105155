"""bf16 vs MXFP8 attention (and their qkv_post kernels) at the headline clip's DiT shape: 48 heads x 18226 tokens."""
import math
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dove_amd import ops  # noqa: E402

N, H, Lt = 18226, 48, 226
npad = (N + 127) // 128 * 128
D = H * 64
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(N, 3 * D, device="cuda", generator=g).to(torch.bfloat16)
one, zero = torch.ones(64, device="cuda"), torch.zeros(64, device="cuda")
ang = torch.rand(N - Lt, 32, device="cuda", generator=g) * 6.28
cos, sin = ang.cos().repeat_interleave(2, 1).contiguous(), ang.sin().repeat_interleave(2, 1).contiguous()
qs = 0.125 * math.log2(math.e)
zb = lambda *s: torch.zeros(*s, dtype=torch.bfloat16, device="cuda")   # noqa: E731
zu = lambda *s: torch.zeros(*s, dtype=torch.uint8, device="cuda")      # noqa: E731
Qh, Kh, Vt = zb(H, npad, 64), zb(H, npad, 64), zb(H, 64, npad)
Q8, K8, V8, Vs = zu(H, npad, 64), zu(H, npad, 64), zu(H, 64, npad), zu(H, npad // 64, 64, 2)
out = zb(N, D)
out8 = zb(N, D)
fns = {
    "qkv_post bf16": lambda: ops.qkv_post(qkv, N, npad, H, Lt, one, zero, one, zero, cos, sin, qs, 1e-6, Qh, Kh, Vt),
    "qkv_post mxfp8": lambda: ops.qkv_post_mx(qkv, N, npad, H, Lt, one, zero, one, zero, cos, sin, qs, 1e-6, Q8, K8, V8, Vs),
    "attention bf16": lambda: ops.attention(Qh, Kh, Vt, N, npad, H, out),
    "attention mxfp8": lambda: ops.attention_mx(Q8, K8, V8, Vs, N, npad, H, out8),
}
t = {k: [] for k in fns}
for rnd in range(4):
    for k, f in fns.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            f()
        e1.record()
        torch.cuda.synchronize()
        if rnd:
            t[k].append(e0.elapsed_time(e1) / 5)
fl = 4.0 * N * N * 64 * H
for k in fns:
    ms = statistics.median(t[k])
    extra = f"{fl / ms / 1e9:7.1f} TFLOP/s" if k.startswith("attention") else f"{N * 3 * D * 2 / ms / 1e6:7.1f} GB/s read"
    print(f"{k:16s} {ms:7.3f} ms  {extra}", flush=True)
d = (out8.float() - out.float())
print(f"mxfp8 vs bf16 attention output: rel RMS {float(d.pow(2).mean().sqrt() / out.float().pow(2).mean().sqrt()):.4f}")
