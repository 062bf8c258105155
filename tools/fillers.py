"""How many single-issue VALU instructions hide behind one MFMA of a one-wave-per-SIMD stream on gfx950 (tools/exp/fillers_exp.hip): ns per 32
MFMA-pipe cycles (two 16x16x32 or one 32x32x16) with K = 0 .. 8 fillers hand-placed behind the MFMAs (best of 5 launches of 20000 x 1024 cycles), for both MFMA shapes and three filler kinds.  python tools/fillers.py"""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = os.path.join(ROOT, "tools", "exp", "libfillers_exp.so")
if not os.path.exists(so):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", os.path.join(ROOT, "tools", "exp", "fillers_exp.hip"), "-o", so])
lib = C.CDLL(so)
lib.fillers.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int]
iters = 20000
KS = (0, 1, 2, 3, 4, 5, 6, 8)
for shape in (16, 32):
    for kind, name in ((0, "v_fma_f32"), (1, "GroupNorm+SiLU-like mix (2 trans in 7)"), (2, "v_exp_f32")):
        res = (C.c_float * 9)()
        assert lib.fillers(shape, kind, res, iters) == 0
        ns = [res[i] * 1e6 / (iters * 32) for i in range(8)]          # per 32 pipe cycles (2 x 16x16x32 or 1 x 32x32x16)
        print(f"{'16x16x32' if shape == 16 else '32x32x16'}  fillers: {name:40s} ns per 32 pipe cycles at K = " +
              "  ".join(f"{k}:{v:5.2f}" for k, v in zip(KS, ns)), flush=True)
