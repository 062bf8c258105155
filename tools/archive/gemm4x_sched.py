"""A/B of the LDS-DMA placement inside gemm4x's K step (TIMING build; DOVE_GEMM4X_SCHED = 0 product order, 1 / 2 sched_group_barrier
order with the DMAs in a burst, 1 = product: fenced groups of {1 DMA, 2 fragment reads, 4 MFMA}): time of the DiT's plain linears at 18 226 rows + a sampled check
against torch fp32.  Run once per variant (the library reads the variable once): see tools/runs/gpu_gemm4x_sched.sh."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dove_amd import lib as _L, ops  # noqa: E402

os.environ["DOVE_GEMM8P"] = "0"          # the product GEMM is gemm8p since round 3; gemm4x lives on in the TIMING build for this tool
_L.use_timing_build()
M = 18226
g = torch.Generator(device="cuda").manual_seed(1)
rows = torch.tensor([0, 255, 256, 4095, 9999, 16383, 16384, 18225], device="cuda")
out = []
for K, N in ((3072, 9216), (3072, 3072), (12288, 3072)):
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
    pc = ops.pack_conv(w.float(), None, "cuda")
    x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    y = ops.linear(x, pc)
    ref = x[rows].float() @ w.float().t()
    err = float((y[rows].float() - ref).abs().max() / ref.abs().max())
    ts = []
    for rnd in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.linear(x, pc, out=y)
        e1.record()
        torch.cuda.synchronize()
        if rnd:
            ts.append(e0.elapsed_time(e1) / 10)
    ms = sorted(ts)[1]
    out.append(f"{K}->{N}: {ms:.3f} ms {2.0 * M * K * N / ms / 1e9:7.1f} TF (rel err {err:.1e})")
print(f"DOVE_GEMM4X_SCHED={os.environ.get('DOVE_GEMM4X_SCHED', '1')}:  " + "   ".join(out), flush=True)
