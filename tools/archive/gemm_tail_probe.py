"""What the row tail of the DiT's N = 3072 linears costs: 16 384 rows (three full rounds of 256 tiles) against 18 226 rows (+ 1842 rows on igemm_fast)
and against other row counts around the round boundaries."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dove_amd import ops  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(1)
for K, N in ((3072, 3072), (12288, 3072), (3072, 9216), (3072, 12288)):
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
    pc = ops.pack_conv(w.float(), torch.zeros(N, device="cuda"), "cuda")
    line = []
    for m in (16384, 17408, 18226, 1842, 21760 if N == 3072 else 18226):
        x = torch.randn(m, K, device="cuda", generator=g).to(torch.bfloat16)
        y = torch.empty(m, N, dtype=torch.bfloat16, device="cuda")
        ts = []
        for rnd in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                ops.linear(x, pc, out=y)
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                ts.append(e0.elapsed_time(e1) / 8)
        ms = statistics.median(ts)
        line.append(f"{m} rows {ms:.3f} ms ({2.0 * m * K * N / ms / 1e9:.0f} TF)")
    print(f"{K}->{N}: " + "   ".join(line), flush=True)
