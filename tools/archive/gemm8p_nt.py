"""gemm8p: plain vs nontemporal (aux = 2) output stores, within one run (TIMING build, DOVE_IGEMM_ABLATE: 16 = forced off, 8 = forced on; the product rule turns them on for outputs > 256 MB)."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dove_amd import lib as _L, ops  # noqa: E402

_L.use_timing_build()
M = 18226
g = torch.Generator(device="cuda").manual_seed(1)
for name, K, N, kw in (("qkv", 3072, 9216, {}), ("ff1", 3072, 12288, {"act": 1}), ("ff2 shape plain", 12288, 3072, {})):
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
    pc = ops.pack_conv(w.float(), torch.zeros(N, device="cuda"), "cuda")
    x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    ys, ts = {}, {16: [], 8: []}
    for rnd in range(5):
        for v in (16, 8):
            os.environ["DOVE_IGEMM_ABLATE"] = str(v)
            y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
            ops.linear(x, pc, out=y, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                ops.linear(x, pc, out=y, **kw)
            e1.record()
            torch.cuda.synchronize()
            ys[v] = y
            if rnd:
                ts[v].append(e0.elapsed_time(e1) / 8)
    t0, t1 = statistics.median(ts[16]), statistics.median(ts[8])
    print(f"{name:16s}: plain stores {t0:.3f} ms | nt stores {t1:.3f} ms ({(t0 / t1 - 1) * 100:+.1f} %)  equal {bool(torch.equal(ys[16], ys[8]))}", flush=True)
