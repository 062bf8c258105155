"""Scouting for the next round: can the HBM-bound GroupNorm-apply of one frame-batch run in the shadow of the MFMA-bound conv of another?
A persistent conv owns every CU it is launched on (whole register file, 147 KB of LDS), so the two only overlap if they are kept on DISJOINT
CUs: two HIP streams created with hipExtStreamCreateWithCUMask - 256 - k CUs for the conv, k for gn_apply.  Timing library (DOVE_CU_LIMIT sizes
the persistent grid for the masked stream).  Prints: each kernel alone on the whole chip, alone on its partition, and both side by side."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dove_amd import lib as L
L.use_timing_build()
from dove_amd import ops

hip = C.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]


def masked_stream(lo, hi, ncu=256):
    words = (ncu + 31) // 32
    m = (C.c_uint32 * words)()
    for i in range(lo, hi):
        m[i // 32] |= 1 << (i % 32)
    h = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(h), words, m)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(h.value)


dev = torch.device("cuda", 0)
ncu = torch.cuda.get_device_properties(0).multi_processor_count
g = torch.Generator(device=dev).manual_seed(3)
T, H, W, Cc = 8, 720, 1280, 128
x = torch.randn(T, H, W, Cc, device=dev, generator=g).to(torch.bfloat16)
cache = torch.randn(2, H, W, Cc, device=dev, generator=g).to(torch.bfloat16)
pc = ops.pack_conv(torch.randn(Cc, Cc, 3, 3, 3) * (Cc * 27) ** -0.5, torch.zeros(Cc), dev)
gam, bet = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
stats = ops.groupnorm_stats(x, 1e-6)
y = torch.empty_like(x)
out = torch.empty_like(x)
x2 = x.clone()


def t_stream(st, fn, reps=5):
    with torch.cuda.stream(st):
        fn(); st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps):
            fn()
        e1.record(st)
    st.synchronize()
    return e0.elapsed_time(e1) / reps


conv = lambda: ops.conv(x, pc, cache=cache, out=out)
gn = lambda: ops.groupnorm_apply(x2, stats, gam, bet, silu=True, out=y)
main = torch.cuda.current_stream()
os.environ.pop("DOVE_CU_LIMIT", None)
tc, tg = t_stream(main, conv), t_stream(main, gn)
print(f"whole chip ({ncu} CUs): conv 128->128 3x3x3 on 8x720x1280 {tc:.3f} ms; gn_apply+SiLU on the same tensor {tg:.3f} ms ({4 * x.numel() / tg / 1e9:.2f} TB/s)")
for k in (8, 16, 24, 32):
    sa, sb = masked_stream(0, ncu - k, ncu), masked_stream(ncu - k, ncu, ncu)
    os.environ["DOVE_CU_LIMIT"] = str(ncu - k)
    tca = t_stream(sa, conv)
    tgb = t_stream(sb, gn)
    # side by side: one conv on A, as many gn_apply on B as fit its duration (what a pipelined VAE would put in its shadow)
    n_gn = max(1, int(tca / tgb))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    sa.wait_stream(main); sb.wait_stream(main)
    e0.record(main)
    sa.wait_event(e0); sb.wait_event(e0)
    reps = 4
    with torch.cuda.stream(sa):
        for _ in range(reps):
            conv()
    with torch.cuda.stream(sb):
        for _ in range(reps * n_gn):
            gn()
    main.wait_stream(sa); main.wait_stream(sb)
    e1.record(main)
    torch.cuda.synchronize()
    both = e0.elapsed_time(e1) / reps
    serial = tc + n_gn * tg
    print(f"k = {k:2d}: conv alone on {ncu - k} CUs {tca:.3f} ms (x{tca / tc:.3f}); gn_apply alone on {k} CUs {tgb:.3f} ms ({4 * x.numel() / tgb / 1e9:.2f} TB/s); "
          f"1 conv || {n_gn} gn_apply side by side {both:.3f} ms vs {serial:.3f} ms one after the other on the whole chip (x{both / serial:.3f})", flush=True)
os.environ.pop("DOVE_CU_LIMIT", None)
