"""gfx950 instruction-throughput microbenchmarks (tools/exp/ubench_exp.hip): cycles per unrolled body for 1..4 waves/SIMD."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = C.CDLL(os.path.join(ROOT, "tools", "exp", "libubench_exp.so"))
lib.ubench.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
NAMES = {0: "32 v_exp_f32", 1: "32 v_add_f32", 2: "16 exp + 16 add interleaved", 3: "16 MFMA 32x32x16 (4 acc; two groups of 8)", 4: "8 MFMA + 32 add (4/gap)",
         5: "8 MFMA + 16 exp (2/gap)", 6: "8 MFMA + 16 exp + 40 add + 8 cvt", 7: "32 v_max3_f32", 8: "32 v_cvt_pk_bf16_f32",
         9: "32 v_permlane32_swap", 10: "16 v_pk_add_f32", 11: "8 MFMA then 16 exp (phases)", 12: "32 v_fma_f32"}
out = torch.zeros(4, dtype=torch.int64, device="cuda")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
for mode, name in NAMES.items():
    row = []
    for w in (1, 2, 3, 4):
        for _ in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = lib.ubench(mode, w, reps, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
            e1.record()
            assert rc == 0
            torch.cuda.synchronize()
        row.append(int(out[0]) / reps)
        if w == 4:
            ms = e0.elapsed_time(e1)
            ghz = int(out[0]) / (ms * 1e6)
            name = f"{name} [{ghz:.2f} Gtick/s wall, {int(out[0]) / max(int(out[2]), 1) * 0.1:.2f} Gtick/s vs 100MHz realtime]"
    print(f"{name:90s} ticks/body per wave @1,2,3,4 waves/SIMD: " + "  ".join(f"{v:8.1f}" for v in row) +
          "   per-SIMD ticks/body: " + "  ".join(f"{v / (i + 1):7.1f}" for i, v in enumerate(row)), flush=True)
