"""Discover the operand / scale layout of the MX MFMA (tools/exp/mxprobe_exp.hip)."""
import ctypes as C
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = C.CDLL(os.path.join(ROOT, "tools", "exp", "libmxprobe_exp.so"))
lib.mxprobe.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 6
g = torch.Generator().manual_seed(0)
A = (torch.randn(32, 64, generator=g)).to(torch.float8_e4m3fn)
B = (torch.randn(32, 64, generator=g)).to(torch.float8_e4m3fn)
ea = torch.randint(120, 134, (32, 2, 4), generator=g, dtype=torch.int64)      # [row][khalf][byte]
eb = torch.randint(120, 134, (32, 2, 4), generator=g, dtype=torch.int64)
wa = (ea[..., 0] | (ea[..., 1] << 8) | (ea[..., 2] << 16) | (ea[..., 3] << 24)).to(torch.int32)
wb = (eb[..., 0] | (eb[..., 1] << 8) | (eb[..., 2] << 16) | (eb[..., 3] << 24)).to(torch.int32)
Ad, Bd, wad, wbd = A.view(torch.uint8).cuda(), B.view(torch.uint8).cuda(), wa.cuda(), wb.cuda()
out = torch.zeros(32, 32, device="cuda")
for layout in (0, 1):
    for ops in range(4):
        lib.mxprobe(ops, layout, Ad.data_ptr(), Bd.data_ptr(), wad.data_ptr(), wbd.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        got = out.cpu()
        best = None
        for byte in range(4):
            for transposed in (False, True):
                sa = torch.pow(2.0, ea[:, :, byte].float() - 127)          # [row][khalf]
                sb = torch.pow(2.0, eb[:, :, byte].float() - 127)
                Af = A.float().reshape(32, 2, 32) * sa[..., None]
                Bf = B.float().reshape(32, 2, 32) * sb[..., None]
                ref = Af.reshape(32, 64) @ Bf.reshape(32, 64).t()
                if transposed:
                    ref = ref.t()
                err = float((got - ref).abs().max() / ref.abs().max())
                if best is None or err < best[0]:
                    best = (err, byte, transposed)
        print(f"layout {layout} opsel {ops}: best match rel err {best[0]:.3e} with scale byte {best[1]}, transposed={best[2]}")
