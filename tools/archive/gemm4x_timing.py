"""Cycle accounting of one persistent gemm4x workgroup (TIMING build, selected by passing debug_buf): per wave, s_memtime
ticks per tile in the K walk (of which: counted-vmcnt wait, step barrier) and in the epilogue."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dove_amd import lib as _L, ops  # noqa: E402

os.environ["DOVE_GEMM8P"] = "0"          # the product GEMM is gemm8p since round 3; gemm4x lives on in the TIMING build for this tool
_L.use_timing_build()          # s_memtime phase logs live only in the -DDOVE_TIMING_BUILD library

M = 18226
for K, N in ((3072, 9216), (3072, 3072), (12288, 3072)):
    w = torch.randn(N, K, device="cuda") * K ** -0.5
    pc = ops.pack_conv(w, torch.zeros(N, device="cuda"), "cuda")
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    buf = torch.zeros(64, dtype=torch.int64, device="cuda")
    y = ops.conv(x.view(1, 1, M, K), pc)
    torch.cuda.synchronize()
    for _ in range(2):
        ops.conv(x.view(1, 1, M, K), pc, out=y, debug_buf=buf)
    torch.cuda.synchronize()
    t = buf.cpu().view(8, 8)
    for wv in (0, 3):
        walk, wait, bar, epi, n, steps = (int(v) for v in t[wv][:6])
        n = max(n, 1)
        print(f"K={K} N={N} wave {wv}: tiles {n} steps/tile {steps}  K-walk {walk / n:.0f} ({walk / n / max(steps, 1):.1f}/step, "
              f"vmcnt-wait {wait / n / max(steps, 1):.1f}, barrier {bar / n / max(steps, 1):.1f})  epilogue {epi / n:.0f}")
