"""Cycle accounting of one conv3x3_halo4x workgroup (TIMING build, DOVE_CONV_HALO4X=1 DOVE_HALO4X_CFG=9): per wave the
total s_memtime ticks of the K walk, the part spent in the counted vmcnt wait and the part in the step barrier."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dove_amd import ops  # noqa: E402

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
full = buf.cpu()
t = full[:16].view(4, 4)
for wv in range(4):
    tot, wait, bar, n = (int(v) for v in t[wv])
    print(f"wave {wv}: steps {n}  total {tot} ticks = {tot / max(n, 1):.1f}/step   vmcnt-wait {wait / max(n, 1):.1f}/step   "
          f"barrier {bar / max(n, 1):.1f}/step   body {(tot - wait - bar) / max(n, 1):.1f}/step")
print("prologue ticks per wave:", full[16:20].tolist(), " epilogue (incl. store drain):", full[20:24].tolist())
