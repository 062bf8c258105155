"""Back-to-back timing of the DiT's HBM-bound glue kernels (ln_mod, qk_post + v_post) with two builds of the library in two processes on one box:
    python tools/small_ab.py            # parent: runs itself once per library and prints both
The second library (dove_amd/libdove_hip_prevnorm.so) is the product library linked with the PREVIOUS csrc/norm.hip."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    import torch
    from dove_amd import lib as L
    L.LIB_PATH = sys.argv[1]
    from dove_amd import ops
    dev = torch.device("cuda", 0)
    BF = torch.bfloat16
    N, heads = 18226, 48
    npad = (N + 127) // 128 * 128

    def timeit(fn, iters=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / iters * 1e3

    h = torch.randn(N, 3072, device=dev).to(BF)
    big = torch.randn(64, 1024, 1024, device=dev)          # 256 MB: flushes the Infinity Cache between calls
    g3 = torch.ones(3072, device=dev)
    mod = torch.randn(2, 2, 3072, device=dev) * 0.1
    o = torch.empty_like(h)
    qkv = torch.randn(N, 9216, device=dev).to(BF)
    g64 = torch.ones(64, device=dev)
    cs = torch.rand(N - 226, 64, device=dev)
    Qh = torch.zeros(heads, npad, 64, device=dev, dtype=BF)
    Kh = torch.zeros_like(Qh)
    Vt = torch.zeros(heads, 64, npad, device=dev, dtype=BF)
    n2 = torch.zeros(heads, 2, device=dev)

    def ln():
        big.add_(1.0)
        ops.layernorm_modulate(h, g3, g3, 1e-5, mod, 226, out=o)

    def qp():
        big.add_(1.0)
        ops.qkv_post(qkv, N, npad, heads, 226, g64, g64, g64, g64, cs, cs, 0.18, 1e-6, Qh, Kh, Vt, norm2=n2)

    def flush():
        big.add_(1.0)
    tf = timeit(flush)
    print(f"{os.path.basename(sys.argv[1])}: ln_mod {timeit(ln) - tf:7.1f} us   qkv_post (qk_post + v_post + clear) {timeit(qp) - tf:7.1f} us   "
          f"checks: {float(o.float().abs().sum()):.6e} {float(Qh.float().abs().sum()):.6e} {float(Kh.float().abs().sum()):.6e} {float(n2.sum()):.6e}")
else:
    for rnd in range(2):
        for so in ("libdove_hip_prevnorm.so", "libdove_hip.so"):
            subprocess.run([sys.executable, os.path.abspath(__file__), os.path.join(ROOT, "dove_amd", so)], check=True)
