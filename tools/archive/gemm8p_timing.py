"""s_memtime accounting of one persistent workgroup of gemm8p and gemm4x (TIMING build): ticks per tile in the K walk and in the epilogue,
for the DiT's plain / GELU / gated linears."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dove_amd import lib as _L, ops  # noqa: E402

_L.use_timing_build()
M = 18226
g = torch.Generator(device="cuda").manual_seed(1)
for name, K, N, opt in (("qkv", 3072, 9216, {}), ("ff1", 3072, 12288, {"act": 1}), ("ff2", 12288, 3072, {"gated": True})):
    w = torch.randn(N, K, device="cuda", generator=g) * K ** -0.5
    pc = ops.pack_conv(w, torch.zeros(N, device="cuda"), "cuda")
    x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    kw = {}
    if opt.get("act"):
        kw["act"] = 1
    if opt.get("gated"):
        kw["resid"] = torch.randn(1, 1, M, N, device="cuda", generator=g).to(torch.bfloat16)
        kw["gate"] = torch.randn(2, N, device="cuda", generator=g)
        kw["gate_split"] = 226
    for v in (1, 0):
        if v == 0 and kw:
            continue                                           # gemm4x's timing instantiation exists for the plain epilogue only
        os.environ["DOVE_GEMM8P"] = str(v)
        buf = torch.zeros(64, dtype=torch.int64, device="cuda")
        y = ops.conv(x.view(1, 1, M, K), pc, **kw)
        torch.cuda.synchronize()
        for _ in range(2):
            ops.conv(x.view(1, 1, M, K), pc, out=y, debug_buf=buf, **kw)
        torch.cuda.synchronize()
        t = buf.cpu().view(8, 8)
        for wv in ((0, 4) if v else (0,)):
            walk, b1, b2, epi, n, steps = (int(q) for q in t[wv][:6])
            n = max(n, 1)
            if v:
                print(f"{name} gemm8p wave {wv}: tiles {n}, K-32 phases per tile {steps}: K walk {walk / n:.0f} ticks = {walk / n / steps:.1f} per phase "
                      f"(at the LOAD barrier {b1 / n / steps:.1f}, at the MFMA barrier {b2 / n / steps:.1f}), epilogue {epi / n:.0f}", flush=True)
            else:
                print(f"{name} gemm4x wave {wv}: tiles {n}, K-32 steps per tile {steps}: K walk {walk / n:.0f} ticks = {walk / n / steps:.1f} per step "
                      f"(vmcnt wait {b1 / n / steps:.1f}, barrier {b2 / n / steps:.1f}), epilogue {epi / n:.0f}", flush=True)
