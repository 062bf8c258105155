"""Structural alternatives for the dominant conv kernel, measured on a model of its K walk (tools/exp/convalt_exp.hip): GroupNorm-apply + SiLU
done in LDS on the staged halo and the fragment-read pressure of Winograd-domain accumulation (item 4).
    python tools/convalt.py [groups]          (one group = 9 steps of 32 MFMAs per wave)"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
so = os.path.join(ROOT, "tools", "exp", "libconvalt_exp.so")
if not os.path.exists(so):
    import subprocess
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", os.path.join(ROOT, "tools", "exp", "convalt_exp.hip"), "-o", so])
lib = C.CDLL(so)
lib.convalt.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
SHAPE = len(sys.argv) > 1 and sys.argv[1] == "shape"       # only the MFMA-shape A/B (modes 0 / 4 / 5 / 6)
_nums = [a for a in sys.argv[1:] if a.isdigit()]
groups = int(_nums[0]) if _nums else 4000
blocks = torch.cuda.get_device_properties(0).multi_processor_count
ZEROS = "zeros" in sys.argv            # all-zero operands: the walk's SCHEDULE-limited rate (no data-dependent clock give-back)
init = (torch.zeros(8192 * 8, device="cuda") if ZEROS else torch.randn(8192 * 8, device="cuda")).to(torch.bfloat16)
out = torch.zeros(blocks * 16 * 256, device="cuda")
names = {0: "baseline: 4x4 blocks, 16 fragment reads : 32 MFMAs per step (halo4x's register tile)",
         4: "same tile, operands held in registers (no LDS reads): the matrix pipe in this harness",
         1: "baseline + GroupNorm-apply/SiLU rewrite of the staged halo in LDS (2 slots per lane in 6 of 9 steps)",
         2: "Winograd F(2x2,3x3): 16 independent transform-domain blocks, 64 fragment reads : 32 MFMAs",
         3: "Winograd F(2,3) along W: 4 positions x 2x2 blocks, 32 fragment reads : 32 MFMAs",
         5: "the baseline walk in v_mfma_f32_16x16x32_bf16: 8x8 blocks of 16x16, 16 fragment reads : 64 MFMAs per step",
         6: "the 16x16x32 walk with every step reading the same (conflict-free) fragments",
         7: "16x16x32, fragments register-pipelined across the barrier, one read pinned behind every 4 MFMAs",
         8: "32x32x16, pipelined the same way (the product kernel's scheme): one read behind every 2 MFMAs",
         10: "16x16x32, the product kernel's step as written (asm MFMAs in AGPRs, one other instruction per MFMA pair), no LDS-DMA",
         9: "mode 10 + the product walk's LDS-DMA traffic (13.3 KB per step and workgroup from an L2-resident buffer)"}
res = {}
MODES = (0, 8, 5, 7, 10, 9) if SHAPE else (0, 4, 1, 2, 3, 5, 6, 7, 8, 10, 9)
for rnd in range(3):
    for mode in MODES:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        assert lib.convalt(mode, init.data_ptr(), out.data_ptr(), groups, blocks, None) == 0
        e1.record()
        torch.cuda.synchronize()
        res.setdefault(mode, []).append(e0.elapsed_time(e1))
steps = groups * 9
flop_step = 32 * 2 * 32 * 32 * 16 * 4 * blocks          # 32 MFMAs x 4 waves x CUs
base = sorted(res[0])[1]
print(f"# {blocks} workgroups x 4 waves (one per SIMD), {groups} groups of 9 steps, {'ALL-ZERO' if ZEROS else 'N(0,1)'} bf16 operands in LDS")
for mode in MODES:
    ms = sorted(res[mode])[1]
    ns = ms * 1e6 / steps
    print(f"mode {mode}: {ms:9.3f} ms  {ns:7.1f} ns per 32-MFMA step  {flop_step * steps / (ms * 1e-3) / 1e15:5.2f} PFLOP/s dense-equivalent  x{ms / base:5.3f} of the baseline   {names[mode]}")
t5, t7, t8 = (sorted(res[m])[1] for m in (5, 7, 8))
print(f"16x16x32 instead of 32x32x16: step time x{t5 / base:.3f} as compiled, x{t7 / t8:.3f} with both walks register-pipelined (same MACs, same fragment reads, same accumulator registers)")
if SHAPE:
    sys.exit(0)
t0, t1, t2, t3 = (sorted(res[m])[1] for m in (0, 1, 2, 3))
print(f"GN-apply in LDS: K walk +{100 * (t1 / t0 - 1):.1f} %.  At the bench's 589 ms of conv3x3_halo4x per clip that is +{589 * (t1 / t0 - 1):.0f} ms (model: no staging, "
      f"every conv norm-fused) against the 67.7 ms of gn_apply_kernel it would remove.")
print(f"Winograd F(2x2,3x3): MFMA work /2.25, step time x{t2 / t0:.2f} -> at best x{2.25 / (t2 / t0):.2f} before the input / output transforms "
      f"(32 VALU adds + 8 conversions per 4x4 patch and channel, 4x the accumulators to drain per output pixel).")
print(f"Winograd F(2,3) 1-D: MFMA work /1.5, step time x{t3 / t0:.2f} -> at best x{1.5 / (t3 / t0):.2f} before the transforms.")
