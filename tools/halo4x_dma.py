"""A/B of the LDS-DMA placement inside conv3x3_halo4x's K step (TIMING build; DOVE_HALO4X_DMA = 0 product order, 1 spread): time of
the headline conv shapes + equality with the product library's result.  Run once per variant (see tools/gpu_halo4x_dma.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dove_amd import lib as _L, ops  # noqa: E402

_L.use_timing_build()
g = torch.Generator(device="cuda").manual_seed(1)
out = []
for cin, cout, T, H, W in ((128, 128, 9, 720, 1280), (256, 256, 9, 360, 640), (512, 512, 3, 90, 160)):
    w = torch.randn(cout, cin, 3, 3, 3, device="cuda", generator=g) * (cin * 27) ** -0.5
    pc = ops.pack_conv(w, torch.zeros(cout, device="cuda"), "cuda")
    x = torch.randn(T, H, W, cin, device="cuda", generator=g).to(torch.bfloat16)
    y = ops.conv(x, pc)
    chk = float(y.float().abs().sum())                      # compared across variants by the shell script (same inputs, same seed)
    ts = []
    for rnd in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.conv(x, pc, out=y)
        e1.record()
        torch.cuda.synchronize()
        if rnd:
            ts.append(e0.elapsed_time(e1) / 5)
    ms = sorted(ts)[1]
    out.append(f"{cin}->{cout}@{T}x{H}x{W}: {ms:.3f} ms {2.0 * 27 * cin * cout * T * H * W / ms / 1e9:7.1f} TF (checksum {chk:.6e})")
print(f"DOVE_HALO4X_DMA={os.environ.get('DOVE_HALO4X_DMA', '0')}:  " + "   ".join(out), flush=True)
