"""Cycle accounting of one persistent conv3x3_halo4x workgroup (TIMING build): per wave, s_memtime ticks per tile spent in the K walk, the
pre-epilogue barrier, the epilogue body and the store drain.  The product's 16x16x32 walk by default; DOVE_HALO_M16=0 in the environment:
the 32x32x16 walk of rounds 1-4."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dove_amd import lib as _L, ops  # noqa: E402

_L.use_timing_build()          # s_memtime phase logs live only in the -DDOVE_TIMING_BUILD library

cin, cout = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (128, 128)
w = torch.randn(cout, cin, 3, 3, 3, device="cuda") * (cin * 27) ** -0.5
pc = ops.pack_conv(w, torch.zeros(cout, device="cuda"), "cuda")
x = torch.randn(9, 720, 1280, cin, device="cuda").to(torch.bfloat16)
buf = torch.zeros(32, dtype=torch.int64, device="cuda")
y = ops.conv(x, pc)
torch.cuda.synchronize()
for _ in range(3):
    ops.conv(x, pc, out=y, debug_buf=buf)
torch.cuda.synchronize()
t = buf.cpu().view(4, 8)
for wv in range(4):
    walk, bar, body, drain, n, steps, total = (int(v) for v in t[wv][:7])
    n = max(n, 1)
    print(f"wave {wv}: tiles {n} steps/tile {steps}  per tile: K-walk {walk / n:.0f} ({walk / n / max(steps, 1):.1f}/step)  "
          f"barrier {bar / n:.0f}  epilogue body {body / n:.0f}  store drain {drain / n:.0f}   whole kernel {total} ticks")
