"""Diagnostic for tests/test_parity_gpu.py::test_heavy_tailed_weights_stagewise: the decoder with x50 outlier channels in its mid-block convs,
HIP kernels vs the torch restatement of the same operator graph (tests/emu_ops.py, CPU), resnet by resnet, and both against the fp32 oracle."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dove_amd.ops as real
from dove_amd import config, weights
from dove_amd.vae import AutoencoderKLCogVideoX
from oracle.vae import OracleVAE
import emu_ops
import test_parity_gpu as tp

torch.set_num_threads(min(os.cpu_count() or 1, 64))
v, t, s = config.default_configs()
sc_v, _ = tp.heavy_tail_scales(v, t)
def rms(a, b): return float(((a.float().cpu() - b.float().cpu()) ** 2).mean().sqrt() / (b.float().cpu() ** 2).mean().sqrt())
g = torch.Generator().manual_seed(1)
z = torch.randn(1, 16, 3, 32, 32, generator=g) * 1.4
wv = weights.random_state_dict(weights.vae_param_shapes(v), 78)
for k, f in sc_v.items():
    wv[k] = wv[k] * f
for k in ("decoder.conv_out.conv.weight", "decoder.conv_out.conv.bias"):
    wv[k] = wv[k] * 0.25
rec = {}
orig = AutoencoderKLCogVideoX._resnet
def hooked(self, x, name, cache, zq=None):
    y = orig(self, x, name, cache, zq)
    rec.setdefault(self._tag, {}).setdefault(name, []).append(y.float().cpu())
    return y
AutoencoderKLCogVideoX._resnet = hooked
vae = AutoencoderKLCogVideoX(v, wv, "cuda"); vae._tag = "hip"
d_hip = vae.decode(z.cuda().to(torch.bfloat16)).sample.float().cpu()
torch.cuda.synchronize()
for n in emu_ops.ALL:
    setattr(real, n, getattr(emu_ops, n))
real.pack_conv = emu_ops.pack_conv
vae_e = AutoencoderKLCogVideoX(v, wv, "cpu"); vae_e._tag = "emu"
d_emu = vae_e.decode(z.to(torch.bfloat16)).sample.float()
d32 = OracleVAE(v, wv).decode(z)
dbf = OracleVAE(v, wv, torch.bfloat16).decode(z.to(torch.bfloat16)).float()
print("decode rms-rel vs fp32 oracle: hip %.3e  emu %.3e  bf16-oracle %.3e ; hip vs emu %.3e" % (rms(d_hip, d32), rms(d_emu, d32), rms(dbf, d32), rms(d_hip, d_emu)))
for name in rec["hip"]:
    for i, (a, b) in enumerate(zip(rec["hip"][name], rec["emu"][name])):
        oc = [3, 100, 257, 300, 444, 511] if a.shape[-1] == 512 else []
        extra = ""
        if oc:
            extra = "  outlier-ch |x| %.1f vs rest %.2f" % (float(b[..., oc].abs().mean()), float(b.abs().mean()))
        print(f"  {name} batch {i}: hip vs emu rms-rel {rms(a, b):.3e}  max-rel {float((a - b).abs().max() / b.abs().max()):.3e}{extra}")
