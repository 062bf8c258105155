"""Within-run A/B of experimental attention variants (tools/exp/attn_exp.hip) against the product kernel on the headline
shape (N = 18226, 48 heads x 64): interleaved rounds, median TFLOP/s, max error vs an fp32 torch reference on a small case
that forces the rescale path (spiked keys).  Usage: python tools/attn_ab.py [variants, comma separated]"""
import ctypes as C
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dove_amd import lib as L, ops  # noqa: E402

exp = C.CDLL(os.path.join(ROOT, "tools", "exp", "libattn_exp.so"))
exp.attn_exp.argtypes = [C.c_int] + [C.c_void_p] * 4 + [C.c_longlong, C.c_longlong, C.c_int, C.c_longlong, C.c_void_p]
exp.attn_exp2.argtypes = exp.attn_exp.argtypes
pipe = C.CDLL(os.path.join(ROOT, "tools", "exp", "libattn_pipe_exp.so"))      # variants 30+: software-pipelined kernel
pipe.attn_exp3.argtypes = exp.attn_exp.argtypes
g2 = C.CDLL(os.path.join(ROOT, "tools", "exp", "libattn2g_exp.so"))          # variants 40+: two wave groups one barrier phase apart
g2.attn_exp4.argtypes = exp.attn_exp.argtypes
qb2 = C.CDLL(os.path.join(ROOT, "tools", "exp", "libattn_qb2_exp.so"))        # variants 50+: two query blocks per wave, skewed
qb2.attn_exp5.argtypes = exp.attn_exp.argtypes
BF = torch.bfloat16
_swapped = {}


def vswap(V):
    """V^T in the quad-swapped key order (the product kernel's and variants 40+'s contract), converted once per tensor - not inside
    the timed loop."""
    k = V.data_ptr()
    if k not in _swapped:
        _swapped[k] = (V, ops.vt_quad_swap(V.clone()))
    return _swapped[k][1]


def run(variant, Q, K, V, N, npad, heads, out):
    if variant < 0:                                # the product kernel reads V^T in the quad-swapped key order
        return ops.attention(Q, K, vswap(V), N, npad, heads, out)
    if variant >= 50:
        rc = qb2.attn_exp5(variant, Q.data_ptr(), K.data_ptr(), vswap(V).data_ptr(), out.data_ptr(), N, npad, heads, out.shape[1], L.stream_ptr())
        assert rc == 0, rc
        return out
    if variant >= 40:
        rc = g2.attn_exp4(variant, Q.data_ptr(), K.data_ptr(), vswap(V).data_ptr(), out.data_ptr(), N, npad, heads, out.shape[1], L.stream_ptr())
        assert rc == 0, rc
        return out
    fn = pipe.attn_exp3 if variant >= 30 else (exp.attn_exp2 if variant >= 20 else exp.attn_exp)
    rc = fn(variant, Q.data_ptr(), K.data_ptr(), V.data_ptr(), out.data_ptr(), N, npad, heads, out.shape[1], L.stream_ptr())
    assert rc == 0, rc
    return out


def make(N, heads, seed, qs=0.5):
    npad = (N + 127) // 128 * 128
    g = torch.Generator(device="cuda").manual_seed(seed)
    Q = torch.zeros(heads, npad, 64, dtype=BF, device="cuda")
    K = torch.zeros(heads, npad, 64, dtype=BF, device="cuda")
    V = torch.zeros(heads, 64, npad, dtype=BF, device="cuda")
    Q[:, :N] = (torch.randn(heads, N, 64, device="cuda", generator=g) * qs).to(BF)
    K[:, :N] = torch.randn(heads, N, 64, device="cuda", generator=g).to(BF)
    V[:, :, :N] = torch.randn(heads, 64, N, device="cuda", generator=g).to(BF)
    return Q, K, V, npad


def reference(Q, K, V, N):
    q, k, v = Q[:, :N].float(), K[:, :N].float(), V[:, :, :N].float()
    s = torch.einsum("hqd,hkd->hqk", q, k) * 0.6931471805599453
    p = torch.softmax(s, dim=-1)
    return torch.einsum("hqk,hdk->hqd", p, v).permute(1, 0, 2).reshape(N, -1)


variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else list(range(0, 14))
# ---- correctness: ragged N, spiked keys early and late (forces the deferred-rescale path), large negative scores ----
N, heads = 1000, 4
Q, K, V, npad = make(N, heads, 1, qs=0.3)
K[:, 700] = (Q[:, 7].float() * 40).to(BF)
K[:, 3] = (Q[:, 300].float() * 30).to(BF)
Q[:, 500:520] *= 8.0
ref = reference(Q, K, V, N)
prod = None
for v in [-1] + variants:
    out = torch.zeros(N, heads * 64, dtype=BF, device="cuda")
    run(v, Q, K, V, N, npad, heads, out)
    torch.cuda.synchronize()
    err = (out.float() - ref).abs()
    if v == -1:
        prod = out.clone()
    elif v >= 40:
        print(f"variant {v:3d}: bit-identical to the product kernel: {bool(torch.equal(out, prod))}", flush=True)
    print(f"variant {v:3d}: max err {float(err.max()):.4f}  rms-rel {float(err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()):.2e}  "
          f"finite {bool(torch.isfinite(out.float()).all())}", flush=True)
# ---- speed at the headline shape ----
N, heads = 18226, 48
Q, K, V, npad = make(N, heads, 2)
out = torch.zeros(N, heads * 64, dtype=BF, device="cuda")
flop = 4.0 * heads * N * N * 64
base = run(-1, Q, K, V, N, npad, heads, torch.zeros_like(out)).clone()
for v in variants:
    if v >= 40:
        o2 = run(v, Q, K, V, N, npad, heads, torch.zeros_like(out))
        torch.cuda.synchronize()
        print(f"variant {v:3d} at N = {N}: bit-identical to the product kernel: {bool(torch.equal(o2, base))}  "
              f"max |diff| {float((o2.float() - base.float()).abs().max()):.3g}", flush=True)
times = {v: [] for v in [-1] + variants}
for rnd in range(4):
    for v in times:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            run(v, Q, K, V, N, npad, heads, out)
        e1.record()
        torch.cuda.synchronize()
        if rnd:
            times[v].append(e0.elapsed_time(e1) / 3)
for v, t in times.items():
    ms = statistics.median(t)
    print(f"variant {v:3d}: {ms:7.3f} ms  {flop / ms / 1e9:7.1f} TFLOP/s  (min {min(t):.3f} max {max(t):.3f})", flush=True)
