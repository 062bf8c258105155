"""Do a SIMD's matrix pipe and its VALU overlap when the MFMA stream and the softmax VALU stream come from two different waves
(tools/exp/coissue_exp.hip)?  Prints wall time per mode for sustained runs (long enough for the power management to settle)."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = C.CDLL(os.path.join(ROOT, "tools", "exp", "libcoissue_exp.so"))
lib.coissue.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
blocks = 256
for data in ("normal", "zeros"):
    ops = (torch.randn(blocks * 512 * 32, device="cuda") if data == "normal" else torch.zeros(blocks * 512 * 32, device="cuda")).to(torch.bfloat16)
    out = torch.zeros(blocks * 512, device="cuda")
    names = {1: "MFMA waves only (16 MFMA / body)", 2: "VALU waves only (softmax mix / body)", 3: "both kinds of waves side by side",
             4: "every wave: MFMA block then VALU block (iters/2 each, 8 waves)", 5: "every wave: same, no fence between the blocks",
             6: "side by side, MFMA on EVEN waves / VALU on odd waves", 7: "side by side, MFMA on waves 0,1,4,5 / VALU on 2,3,6,7",
             8: "MFMA on even waves only", 9: "VALU on odd waves only",
             10: "side by side (as mode 3), s_setprio 3 on the VALU waves", 11: "side by side (as mode 3), s_setprio 3 on the MFMA waves",
             20: "side by side, every MFMA followed by 1 x s_nop 7", 21: "side by side, 2 x s_nop 7", 22: "side by side, 3 x s_nop 7",
             23: "side by side, MFMAs on 8 independent accumulators",
             30: "MFMA waves alone, 1 x s_nop 7 after every MFMA", 31: "MFMA waves alone, 2 x s_nop 7", 32: "MFMA waves alone, 3 x s_nop 7",
             33: "MFMA waves alone, 8 independent accumulators"}
    hw = torch.zeros(32, dtype=torch.int32, device="cuda")
    res = {}
    for rnd in range(3):
        for mode in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 20, 21, 22, 23, 30, 31, 32, 33):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            assert lib.coissue(mode, ops.data_ptr(), out.data_ptr(), iters, blocks, hw.data_ptr(), None) == 0
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(mode, []).append(e0.elapsed_time(e1))
    print(f"# operands: {data}; {iters} bodies per wave, {blocks} workgroups x 8 waves (2 per SIMD)")
    for mode, t in res.items():
        ms = sorted(t)[1]
        per = ms * 1e6 / iters
        print(f"mode {mode}: {ms:8.3f} ms   {per:7.1f} ns per body   {names[mode]}", flush=True)
    ids = hw.cpu().tolist()
    # HW_REG_HW_ID (gfx9 layout): wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh [12], se [15:13]
    for b in range(2):
        print(f"   workgroup {b}: wave -> (simd, cu, se): " + "  ".join(f"w{w}:({(ids[b * 8 + w] >> 4) & 3},{(ids[b * 8 + w] >> 8) & 15},{(ids[b * 8 + w] >> 13) & 7})" for w in range(8)))
    t1, t2, t3 = (sorted(res[m])[1] for m in (1, 2, 3))
    print(f"   side by side / max(alone) = {t3 / max(t1, t2):.2f}   side by side / sum(alone) = {t3 / (t1 + t2):.2f}")
