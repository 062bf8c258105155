"""Heavy-tailed VAE weights, ENCODER: HIP kernels vs the torch restatement of the same operator graph (tests/emu_ops.py, CPU), resnet by
resnet, and what the fp32 decoder makes of each latent's error."""
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
wv = weights.random_state_dict(weights.vae_param_shapes(v), 78)
for k, f in sc_v.items():
    wv[k] = wv[k] * f
video = tp.synth_clip(9, 256, 256, seed=4)
noise = torch.randn(1, 16, 3, 32, 32, generator=torch.Generator().manual_seed(10))
rec = {}
orig = AutoencoderKLCogVideoX._resnet
orig_ns = AutoencoderKLCogVideoX._norm_silu
def hooked(self, x, name, cache, zq=None):
    y = orig(self, x, name, cache, zq)
    rec.setdefault(self._tag, {}).setdefault(name, []).append(y.float().cpu())
    return y
def hooked_ns(self, x, name, zq=None):
    y = orig_ns(self, x, name, zq)
    if name.startswith("encoder.mid") or name == "encoder.norm_out":
        rec.setdefault(self._tag, {}).setdefault("NS:" + name, []).append(y.float().cpu())
        st = getattr(x, "gn_stats", None)
        from dove_amd import ops
        rec.setdefault(self._tag, {}).setdefault("ST:" + name, []).append(ops.groupnorm_stats_of(x, self.eps).float().cpu())
    return y
AutoencoderKLCogVideoX._resnet = hooked
AutoencoderKLCogVideoX._norm_silu = hooked_ns
vae = AutoencoderKLCogVideoX(v, wv, "cuda"); vae._tag = "hip"
m_hip = vae.encode(video.cuda().to(torch.bfloat16)).latent_dist.parameters.float().cpu()
torch.cuda.synchronize()
for n in emu_ops.ALL:
    setattr(real, n, getattr(emu_ops, n))
real.pack_conv = emu_ops.pack_conv
vae_e = AutoencoderKLCogVideoX(v, wv, "cpu"); vae_e._tag = "emu"
m_emu = vae_e.encode(video.to(torch.bfloat16)).latent_dist.parameters.float()
o32, obf = OracleVAE(v, wv), OracleVAE(v, wv, torch.bfloat16)
m32, mbf = o32.encode(video), obf.encode(video.to(torch.bfloat16)).float()
def lat(m):
    return m[:, :16] + torch.exp(0.5 * m[:, 16:].clamp(-30, 20)) * noise
print("moments rms-rel vs fp32 oracle: hip %.3e  emu %.3e  bf16-oracle %.3e ; hip vs emu %.3e" % (rms(m_hip, m32), rms(m_emu, m32), rms(mbf, m32), rms(m_hip, m_emu)))
d32 = o32.decode(lat(m32))
print("propagation through the fp32 decoder: hip-latent %.3e  emu-latent %.3e  bf16-latent %.3e" % (rms(o32.decode(lat(m_hip)), d32), rms(o32.decode(lat(m_emu)), d32), rms(o32.decode(lat(mbf)), d32)))
for name in rec["hip"]:
    for i, (a, b) in enumerate(zip(rec["hip"][name], rec["emu"][name])):
        if name.startswith("ST:"):
            print(f"  {name}: stats hip vs emu max rel diff mean {float(((a[:, 0] - b[:, 0]).abs() / (b[:, 0].abs() + 1e-3)).max()):.3e}  rstd {float(((a[:, 1] - b[:, 1]).abs() / b[:, 1].abs()).max()):.3e}")
        else:
            print(f"  {name} batch {i}: hip vs emu rms-rel {rms(a, b):.3e}  max-rel {float((a - b).abs().max() / b.abs().max()):.3e}  |x| max {float(b.abs().max()):.1f} mean {float(b.abs().mean()):.2f}")
