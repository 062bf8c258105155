"""Streaming norm+SiLU variants at the decoder's two big shapes (tools/exp/gn_exp.hip) next to the product gn_apply."""
import ctypes as C
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dove_amd import ops  # noqa: E402

lib = C.CDLL(os.path.join(ROOT, "tools", "exp", "libgn_exp.so"))
lib.gn_exp.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
NAMES = {0: "copy U4", 7: "copy U8", 8: "copy U8 nt", 1: "U1", 2: "U2", 3: "U4", 4: "U8", 5: "U4 nt-store", 6: "U8 nt-store"}
for (T, H, W, Cc) in ((8, 720, 1280, 128), (4, 360, 640, 256)):
    x = torch.randn(T, H, W, Cc, device="cuda").to(torch.bfloat16)
    y = torch.empty_like(x)
    sc, sh = torch.rand(Cc, device="cuda") + 0.5, torch.randn(Cc, device="cuda")
    n16 = x.numel() // 8
    cpp_log = (Cc // 8).bit_length() - 1
    gb = x.numel() * 4 / 1e9
    stats = torch.stack([torch.zeros(32), torch.ones(32)], 1).contiguous().cuda()
    res = {}
    for blocks in (2048, 8192, 32768):
        for v, name in NAMES.items():
            ts = []
            for rnd in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    assert lib.gn_exp(v, x.data_ptr(), y.data_ptr(), n16, cpp_log, sc.data_ptr(), sh.data_ptr(), blocks,
                                      torch.cuda.current_stream().cuda_stream) == 0
                e1.record()
                torch.cuda.synchronize()
                if rnd:
                    ts.append(e0.elapsed_time(e1) / 5)
            res[(blocks, name)] = statistics.median(ts)
    ts = []
    for rnd in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.groupnorm_apply(x, stats, sc, sh, silu=True, out=y)
        e1.record()
        torch.cuda.synchronize()
        if rnd:
            ts.append(e0.elapsed_time(e1) / 5)
    print(f"shape {T}x{H}x{W}x{Cc} ({gb:.2f} GB in+out): product gn_apply {statistics.median(ts):.3f} ms = {gb / statistics.median(ts):.2f} TB/s")
    for name in NAMES.values():
        print(f"  {name:12s} " + "  ".join(f"{b:6d} blk: {res[(b, name)]:.3f} ms {gb / res[(b, name)]:5.2f} TB/s" for b in (2048, 8192, 32768)), flush=True)
