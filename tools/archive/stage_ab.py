"""L2 -> LDS staging rate of a GEMM's operand tiles with no compute (tools/exp/stage_exp.hip): gemm4x's 16-row x 64-B instructions against
8-row x 128-B (full-line) instructions, at the DiT linears' shapes."""
import ctypes as C
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = C.CDLL(os.path.join(ROOT, "tools", "exp", "libstage_exp.so"))
lib.stage_exp.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_void_p]
M = 18226
names = {0: "K32 64-B rows, 3 steps in flight (gemm4x)", 1: "K32 64-B rows, deep queue", 5: "K32 64-B rows, 1 step in flight",
         2: "K64 128-B rows, 1 step in flight", 3: "K64 128-B rows, 0 in flight", 4: "K64 128-B rows, deep queue"}
for K, N in ((3072, 12288), (12288, 3072), (3072, 9216)):
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    tiles = ((M + 255) // 256) * (N // 256)
    gb = tiles * K * 2 * 512 / 1e9
    st = torch.cuda.current_stream().cuda_stream
    t = {v: [] for v in names}
    for rnd in range(4):
        for v in names:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                assert lib.stage_exp(v, x.data_ptr(), w.data_ptr(), M, K, N, st) == 0
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                t[v].append(e0.elapsed_time(e1) / 4)
    print(f"{K}->{N}: {gb:.2f} GB staged per launch; MFMA-bound GEMM time at 1.9 PF = {2.0 * M * K * N / 1.9e12:.3f} ms", flush=True)
    for v in names:
        ms = statistics.median(t[v])
        print(f"   {names[v]:42s} {ms:.3f} ms  {gb / ms:6.2f} TB/s", flush=True)
