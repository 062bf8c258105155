"""Within-run A/B of the eight-wave ping-pong GEMM (gemm8p_kernel) against gemm4x on the DiT's four linears at 18 226 rows
(TIMING build: DOVE_GEMM8P is read per call).  Checks bit-equality of the two kernels and a sampled comparison with torch fp32."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dove_amd import lib as _L, ops  # noqa: E402

_L.use_timing_build()
M = 18226
g = torch.Generator(device="cuda").manual_seed(1)
rows = torch.tensor([0, 31, 127, 128, 255, 256, 4095, 9999, 16383, 16384, 18175, 18176, 18225], device="cuda")
cases = (("qkv", 3072, 9216, {}), ("out", 3072, 3072, {"gated": True}), ("ff1", 3072, 12288, {"act": 1}), ("ff2", 12288, 3072, {"gated": True}),
         ("plain+resid", 3072, 3072, {"resid": True}), ("ragged 4100 rows", 3072, 3072, {"rows": 4100}),
         ("ff1 shape, plain", 3072, 12288, {}), ("qkv shape, GELU", 3072, 9216, {"act": 1}), ("ff2 shape, plain", 12288, 3072, {}))
ONLY = sys.argv[1] if len(sys.argv) > 1 else None          # e.g. "ff1": one case, few repeats (PMC passes)
ROUNDS = 2 if ONLY else 5
for name, K, N, opt in cases:
    if ONLY and name != ONLY:
        continue
    m = opt.get("rows", M)
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda", generator=g) * 0.1
    pc = ops.pack_conv(w.float(), b, "cuda")
    x = torch.randn(m, K, device="cuda", generator=g).to(torch.bfloat16)
    kw = {}
    if opt.get("act"):
        kw["act"] = 1
    if opt.get("gated") or opt.get("resid"):
        kw["resid"] = torch.randn(m, N, device="cuda", generator=g).to(torch.bfloat16)
    if opt.get("gated"):
        kw["gate"] = torch.randn(2, N, device="cuda", generator=g)
        kw["gate_split"] = 226
    VARS = (0, 1)                                              # DOVE_GEMM8P=0: gemm4x (the predecessor, TIMING build only), 1: gemm8p (the product kernel)
    ys, ts = {}, {v: [] for v in VARS}
    for rnd in range(ROUNDS):
        for v in VARS:
            os.environ["DOVE_GEMM8P"] = str(v)
            y = torch.empty(m, N, dtype=torch.bfloat16, device="cuda")
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
    r = rows[rows < m]
    ref = x[r].float() @ w.float().t() + b
    if "act" in kw:
        ref = torch.nn.functional.gelu(ref, approximate="tanh")
    if "gate" in kw:
        gt = torch.where((r >= 226)[:, None], kw["gate"][1][None], kw["gate"][0][None])
        ref = kw["resid"][r].float() + gt * ref
    elif "resid" in kw:
        ref = ref + kw["resid"][r].float()
    err = float((ys[1][r].float() - ref).abs().max() / ref.abs().max())
    t0, t1 = statistics.median(ts[0]), statistics.median(ts[1])
    fl = 2.0 * m * K * N / 1e9
    print(f"{name:18s} {K:5d}->{N:5d}: gemm4x {t0:.3f} ms {fl / t0:7.1f} TF | gemm8p {t1:.3f} ms {fl / t1:7.1f} TF ({(t0 / t1 - 1) * 100:+.1f} %)"
          f"  bit-equal {bool(torch.equal(ys[0], ys[1]))}  rel err vs fp32 {err:.1e}", flush=True)
