"""Heavy-tailed VAE weights, encoder -> decoder round trip on the synthetic clip (no DiT): which decoder input makes the HIP decoder lose
more than the bf16-emulated reference?  Cross-feeds: HIP decoder on the fp32 oracle's latent, on its own latent, on the bf16 oracle's."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from dove_amd import config, weights
from dove_amd.vae import AutoencoderKLCogVideoX
from oracle.vae import OracleVAE
import test_parity_gpu as tp
torch.set_num_threads(min(os.cpu_count() or 1, 64))
v, t, s = config.default_configs()
sc_v, _ = tp.heavy_tail_scales(v, t)
def rms(a, b): return float(((a.float().cpu() - b.float().cpu()) ** 2).mean().sqrt() / (b.float().cpu() ** 2).mean().sqrt())
for mode in ("plain", "heavy"):
    wv = weights.random_state_dict(weights.vae_param_shapes(v), 78)
    if mode == "heavy":
        for k, f in sc_v.items():
            wv[k] = wv[k] * f
    for k in ("decoder.conv_out.conv.weight", "decoder.conv_out.conv.bias"):
        wv[k] = wv[k] * 0.25
    video = tp.synth_clip(9, 256, 256, seed=4)
    noise = torch.randn(1, 16, 3, 32, 32, generator=torch.Generator().manual_seed(10))
    o32, obf = OracleVAE(v, wv), OracleVAE(v, wv, torch.bfloat16)
    vae = AutoencoderKLCogVideoX(v, wv, "cuda")
    def lat(m):   # mean + std * noise, like DiagonalGaussianDistribution.sample
        m = m.float().cpu()
        return m[:, :16] + torch.exp(0.5 * m[:, 16:].clamp(-30, 20)) * noise
    z32 = lat(o32.encode(video)); zbf = lat(obf.encode(video.to(torch.bfloat16))); zh = lat(vae.encode(video.cuda().to(torch.bfloat16)).latent_dist.parameters)
    print(mode, "latent rms-rel vs fp32: hip %.3e  bf16-oracle %.3e ; |z| rms %.3f max %.2f" % (rms(zh, z32), rms(zbf, z32), float(z32.pow(2).mean().sqrt()), float(z32.abs().max())))
    d32 = o32.decode(z32)
    hd = lambda z: vae.decode(z.cuda().to(torch.bfloat16)).sample.float().cpu()
    print(mode, "decoder on the SAME input z32: hip %.3e  bf16-oracle %.3e" % (rms(hd(z32), d32), rms(obf.decode(z32.to(torch.bfloat16)).float(), d32)))
    print(mode, "round trip (own latent): hip %.3e  bf16-oracle %.3e" % (rms(hd(zh), d32), rms(obf.decode(zbf.to(torch.bfloat16)).float(), d32)))
    print(mode, "propagation only (fp32 decoder on the perturbed latents): hip-latent %.3e  bf16-latent %.3e" % (rms(o32.decode(zh), d32), rms(o32.decode(zbf), d32)))
    print(mode, "cross: hip decoder on the bf16 oracle's latent %.3e ; bf16 oracle decoder on hip's latent %.3e" % (rms(hd(zbf), d32), rms(obf.decode(zh.to(torch.bfloat16)).float(), d32)), flush=True)
