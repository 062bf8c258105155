"""What the MFMA pipes sustain on MI355X when they are the only thing running (tools/exp/mfma_storm_exp.hip): operand data x duty."""
import ctypes as C
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = C.CDLL(os.path.join(ROOT, "tools", "exp", "libmfma_storm_exp.so"))
lib.mfma_storm.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
g = torch.Generator(device="cuda").manual_seed(0)
n = 4 * 64 * 8
pat = {
    "zeros": (torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")),
    "ones": (torch.ones(n, device="cuda"), torch.ones(n, device="cuda")),
    "N(0,1) x N(0,1)": (torch.randn(n, device="cuda", generator=g), torch.randn(n, device="cuda", generator=g)),
    "weights N(0,.05) x SiLU(N(0,1))": (torch.randn(n, device="cuda", generator=g) * 0.05,
                                        torch.nn.functional.silu(torch.randn(n, device="cuda", generator=g))),
}
pat = {k: (a.to(torch.bfloat16).contiguous(), b.to(torch.bfloat16).contiguous()) for k, (a, b) in pat.items()}
rb = torch.randint(0, 65536, (2, n), device="cuda", generator=g).to(torch.int32)
rb = (rb & 0x7F7F | (rb & 0x8000)).to(torch.int16)          # random sign / exponent / mantissa bits, exponent < 0xff (no inf / nan)
pat["random bits"] = (rb[0].contiguous(), rb[1].contiguous())
out = torch.zeros(4, dtype=torch.int64, device="cuda")
sink = torch.zeros(4, device="cuda")


def run(nops, wps, reps, a, b):
    best = None
    for _ in range(3):
        assert lib.mfma_storm(nops, wps, reps, a.data_ptr(), b.data_ptr(), out.data_ptr(), sink.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
        torch.cuda.synchronize()
        ns = int(out[0]) * 10.0
        best = ns if best is None else min(best, ns)
    return best


reps = 20000
z = pat["zeros"]
import sys  # noqa: E402
ORDER_ONLY = len(sys.argv) > 1 and sys.argv[1] == "order"
t = run(99, 1, 2000, *z)
f_idle = 4096 * 2000 / t
print(f"s_nop loop, matrix pipes idle: {f_idle:.2f} GHz shader clock (4096 cycles per trip)", flush=True)
if ORDER_ONLY:
    # does the ORDER in which a register tile's products are issued, or the MFMA SHAPE, matter under the power limit?
    # Timed with events around the whole launch (wave 0's own clock flatters it at several waves per SIMD: the oldest wave issues first).
    def run_ev(mode, wps, reps, a, b):
        best = None
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            assert lib.mfma_storm(mode, wps, reps, a.data_ptr(), b.data_ptr(), out.data_ptr(), sink.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
            e1.record()
            torch.cuda.synchronize()
            ns = e0.elapsed_time(e1) * 1e6
            best = ns if best is None else min(best, ns)
        return best

    reps = 100000
    print("operand order / MFMA shape: PF over the whole launch (events) at 1 / 2 / 4 waves per SIMD; zeros = the clock is not the limit")
    for label, mode, flops, per_trip, wlist in (
            ("32x32x16: 16 distinct operand pairs, A and B change every MFMA (mode 0)", 0, 32768, 16, (1, 2, 4)),
            ("32x32x16: 4 x 4 tile ROW-MAJOR, 4 accumulators", 20, 32768, 16, (1, 2, 4)),
            ("32x32x16: 4 x 4 tile SNAKE (one operand changes per MFMA)", 21, 32768, 16, (1, 2, 4)),
            ("32x32x16: the same operand pair every time", 22, 32768, 16, (1, 2, 4)),
            ("32x32x16: 4 x 4 tile row-major on 16 AGPR accumulators (the conv's form)", 25, 32768, 32, (1,)),
            ("16x16x32: mode 0's operand order, 4 accumulators", 23, 16384, 16, (1, 2, 4)),
            ("16x16x32: 4 x 4 tile row-major on 16 accumulators", 24, 16384, 32, (1, 2, 4))):
        for pname in ("zeros", "N(0,1) x N(0,1)", "weights N(0,.05) x SiLU(N(0,1))"):
            a, b = pat[pname]
            row = [flops * 1024 * (reps * per_trip * w) / run_ev(mode, w, reps, a, b) / 1e6 for w in wlist]
            print(f"  {label:74s} {pname:32s} " + "  ".join(f"{v:5.2f} PF" for v in row), flush=True)
    sys.exit(0)
print("pattern                              | 4 waves/SIMD, back to back        | 1 wave/SIMD back to back | 1 wave + 48 nop cycles / MFMA | 1 wave + 96 nop cycles / MFMA")
for name, (a, b) in pat.items():
    t4 = run(0, 4, reps, a, b) / (reps * 16 * 4)
    t1 = run(0, 1, reps, a, b) / (reps * 16)
    t3 = run(3, 1, reps, a, b) / (reps * 16)
    t6 = run(6, 1, reps, a, b) / (reps * 16)
    pf = lambda ns: 32768 * 1024 / ns / 1e6                 # noqa: E731
    print(f"{name:36s} | {t4:6.2f} ns/MFMA/SIMD = {pf(t4):5.2f} PF | {t1:6.2f} ns = {pf(t1):5.2f} PF | {t3:6.2f} ns ({t3 * f_idle:5.1f} cyc @idle clk) "
          f"| {t6:6.2f} ns ({t6 * f_idle:5.1f} cyc @idle clk)", flush=True)

print("\naccumulator dependency distance (ns per MFMA per SIMD at 1 / 2 / 4 waves per SIMD; zeros = no data-dependent throttling):")
for label, mode in (("4 accumulators round robin (distance 4)", 0), ("2 accumulators alternating (distance 2)", 12),
                    ("1 accumulator back to back (distance 1)", 11), ("chains of 4 per accumulator (attention's order)", 14)):
    for pname in ("zeros", "N(0,1) x N(0,1)"):
        a, b = pat[pname]
        row = [run(mode, w, reps, a, b) / (reps * 16 * w) for w in (1, 2, 4)]
        print(f"  {label:50s} {pname:16s} " + "  ".join(f"{v:6.2f}" for v in row), flush=True)
