"""Per-phase cycle log of one conv3x3_halo8 workgroup (TIMING build): prints, for waves 0 (group A) and 4 (group B),
the s_memtime deltas of phase 1 (staging + ds_reads + drain), barrier 1, phase 2 (16 MFMAs issued), barrier 2."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dove_amd import lib as _L, ops  # noqa: E402

_L.use_timing_build()          # s_memtime phase logs live only in the -DDOVE_TIMING_BUILD library

cin, cout = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (128, 128)
w = torch.randn(cout, cin, 3, 3, 3, device="cuda") * (cin * 27) ** -0.5
pc = ops.pack_conv(w, torch.zeros(cout, device="cuda"), "cuda")
x = torch.randn(9, 720, 1280, cin, device="cuda").to(torch.bfloat16)
buf = torch.zeros(2 * 36 * 5, dtype=torch.int64, device="cuda")
y = ops.conv(x, pc)
torch.cuda.synchronize()
ops.conv(x, pc, out=y, debug_buf=buf)
torch.cuda.synchronize()
t = buf.cpu().view(2, 36, 5)
for g in range(2):
    d = t[g]
    ph1 = (d[:, 1] - d[:, 0]).float()
    b1 = (d[:, 2] - d[:, 1]).float()
    ph2 = (d[:, 3] - d[:, 2]).float()
    b2 = (d[:, 4] - d[:, 3]).float()
    tot = (d[1:, 0] - d[:-1, 0]).float()
    print(f"group {'AB'[g]}: phase1 {ph1.mean():.0f} (min {ph1.min():.0f} max {ph1.max():.0f})  barrier1 {b1.mean():.0f}  "
          f"phase2 {ph2.mean():.0f} (min {ph2.min():.0f} max {ph2.max():.0f})  barrier2 {b2.mean():.0f}  step {tot.mean():.0f}")
    print("   per-step [ph1 b1 ph2 b2]:", [(int(a), int(b), int(c), int(e)) for a, b, c, e in zip(ph1[:12], b1[:12], ph2[:12], b2[:12])])
print("A.step0.start - B.step0.start:", int(t[0, 0, 0] - t[1, 0, 0]))
