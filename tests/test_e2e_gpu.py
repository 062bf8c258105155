"""-m gpu: the whole one-step SR operator through the C-ABI vs the fp32 CPU oracle (same seeded weights, clip,
posterior noise, text embedding).  Model = CogVideoX1.5-5B widths with 2 DiT layers so the oracle runs in
seconds; every kernel shape class of the full model is exercised.
Tolerance (floating point, stated): PSNR(HIP-bf16, oracle-fp32) must be >= PSNR(oracle-bf16-emulation of the
reference's rounding points, oracle-fp32) - 0.05 dB, and > 35 dB absolute.  Random-init weights saturate ~40 % of the
[0,1] output, so the PRE-CLAMP decoder output is gated as well (RMS-relative error, hip <= 1.25 x what the reference's
own bf16 run loses); per-stage gates are RMS-relative, not max-norm.  Full depth (42 layers) and the configs[0] size:
tests/test_parity_gpu.py."""
import os
import time

import pytest
import torch

from dove_amd import config, weights
from dove_amd.inference import process_video
from dove_amd.pipeline import CogVideoXPipeline
from oracle import dit as odit
from oracle.vae import OracleVAE

pytestmark = pytest.mark.gpu


def psnr(a, b):
    mse = ((a.float() - b.float()) ** 2).flatten(3).mean(-1)      # per frame
    return float((10 * torch.log10(1.0 / (mse + 1e-8))).mean())


def rms_rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float(((a - b) ** 2).mean().sqrt() / (b ** 2).mean().sqrt().clamp_min(1e-20))


@pytest.fixture(scope="module")
def setup(golden_dir):
    from safetensors.torch import load_file
    v, t, s = config.small_configs(num_layers=2)
    seed = 21
    pipe = CogVideoXPipeline.from_config(v, t, s, seed=seed, device="cuda")
    wv = weights.random_state_dict(weights.vae_param_shapes(v), seed)
    wt = weights.random_state_dict(weights.dit_param_shapes(t), seed)
    text = load_file(os.path.join(golden_dir, "empty_prompt_embedding.safetensors"))["prompt_embedding"]
    assert text.shape == (226, 4096) and text.dtype == torch.bfloat16
    return pipe, (v, t, s), wv, wt, text


def synth_clip(F, H, W, seed=42):
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    vid = torch.zeros(3, F, H, W)
    for c in range(3):
        for _ in range(6):
            fx, fy, ph = torch.rand(3, generator=g)
            for f in range(F):
                vid[c, f] += torch.sin(2 * 3.14159 * (fx * 4 * (xx + f) / W + fy * 4 * yy / H) + ph * 6.28) / 6
    return (vid + 0.03 * torch.randn(3, F, H, W, generator=g)).clamp(-1, 1)[None]


@pytest.mark.parametrize("F,H,W", [(9, 64, 64), (17, 48, 80)])
def test_process_video_psnr(setup, F, H, W):
    pipe, (v, t, s), wv, wt, text = setup
    video = synth_clip(F, H, W)
    T = 1 + (F - 1) // 4
    noise = torch.randn(1, 16, T, H // 8, W // 8, generator=torch.Generator().manual_seed(1))
    got = process_video(pipe, video.cuda(), empty_prompt_embedding=text, posterior_noise=noise.cuda()).float().cpu()
    torch.cuda.synchronize()
    tr32 = {}
    ref32 = odit.process_video(OracleVAE(v, wv), odit.OracleDiT(t, wt), s, video, text.float()[None], noise, trace=tr32)
    assert got.shape == ref32.shape == (1, 3, F, H, W)
    p_got = psnr(got, ref32)
    sat = float(((ref32 <= 0) | (ref32 >= 1)).float().mean())
    # the same computation up to the decoder output BEFORE the clamp (no saturated pixels hiding error)
    lat = pipe.vae.encode(video.cuda().to(torch.bfloat16)).latent_dist.sample(noise=noise.cuda()) * 0.7
    ncopy = lat.shape[2] % 2
    lat = torch.cat([lat[:, :, :1].repeat(1, 1, ncopy, 1, 1), lat], dim=2).permute(0, 2, 1, 3, 4).contiguous()
    from dove_amd.rope import prepare_rotary_positional_embeddings
    rope = prepare_rotary_positional_embeddings(height=H, width=W, num_frames=lat.shape[1], transformer_config=pipe.transformer.config,
                                                vae_scale_factor_spatial=8, device=lat.device)
    ts = torch.full((1,), 399, dtype=torch.long, device=lat.device)
    vel = pipe.transformer(hidden_states=lat, encoder_hidden_states=text[None].cuda(), timestep=ts, image_rotary_emb=rope,
                           return_dict=False)[0]
    dec = pipe.decode_latents(pipe.scheduler.get_velocity(vel, lat, ts)[:, ncopy:].contiguous())
    assert torch.equal(got, (dec.float() * 0.5 + 0.5).clamp(0, 1).to(torch.bfloat16).float().cpu())
    e_dec = rms_rel(dec, tr32["decoded"])
    print(f"[e2e {F}x{H}x{W}] PSNR(hip, fp32 oracle) = {p_got:.2f} dB ({100 * sat:.1f} % of the pixels saturated); "
          f"pre-clamp decoder output rms-rel {e_dec:.3e}")
    if (F, H, W) == (9, 64, 64):
        trbf = {}
        refbf = odit.process_video(OracleVAE(v, wv, torch.bfloat16), odit.OracleDiT(t, wt, torch.bfloat16), s, video,
                                   text[None], noise, trace=trbf).float()
        p_bf = psnr(refbf, ref32)
        e_bf = rms_rel(trbf["decoded"], tr32["decoded"])
        print(f"[e2e] bf16-emulated reference vs fp32 oracle: PSNR {p_bf:.2f} dB, pre-clamp rms-rel {e_bf:.3e}; per stage " +
              " ".join(f"{k} {rms_rel(trbf[k], tr32[k]):.2e}" for k in ("moments", "v", "x0")))
        assert p_got >= p_bf - 0.05, (p_got, p_bf)
        assert e_dec <= 1.25 * e_bf, (e_dec, e_bf)
    assert p_got > 35.0, p_got
    assert e_dec < 4e-2, e_dec


def test_noise_step_pre_noising(setup):
    """`--noise_step` (ref :449-457): latent <- add_noise(latent, eps, noise_step) before the DiT.  The reference draws eps
    from the global RNG; here the same draw is reproduced for the oracle by re-seeding (the posterior noise is injected,
    so eps is the only draw)."""
    pipe, (v, t, s), wv, wt, text = setup
    F, H, W = 9, 64, 64
    video = synth_clip(F, H, W, seed=13)
    noise = torch.randn(1, 16, 3, H // 8, W // 8, generator=torch.Generator().manual_seed(2))
    torch.manual_seed(4242)
    got = process_video(pipe, video.cuda(), noise_step=200, empty_prompt_embedding=text, posterior_noise=noise.cuda()).float().cpu()
    torch.manual_seed(4242)
    eps = torch.randn(1, 4, 16, H // 8, W // 8, device="cuda", dtype=torch.bfloat16).float().cpu()
    base = process_video(pipe, video.cuda(), noise_step=0, empty_prompt_embedding=text, posterior_noise=noise.cuda()).float().cpu()
    tr = {}
    ref = odit.process_video(OracleVAE(v, wv), odit.OracleDiT(t, wt), s, video, text.float()[None], noise, noise_step=200,
                             add_noise_eps=eps, trace=tr)
    ref0 = odit.process_video(OracleVAE(v, wv), odit.OracleDiT(t, wt), s, video, text.float()[None], noise)
    p, p_wrong = psnr(got, ref), psnr(base, ref)
    print(f"[noise_step=200] PSNR(hip, oracle) {p:.2f} dB; (hip without pre-noising vs oracle with) {p_wrong:.2f} dB; "
          f"oracle 200 vs 0: {psnr(ref, ref0):.2f} dB")
    assert p > 35.0 and p > p_wrong + 6.0, (p, p_wrong)
    # scheduler coefficients at t=200 through the bf16 cast (diffusers casts alphas_cumprod to the sample dtype first)
    a = pipe.scheduler.alphas_cumprod.to(torch.bfloat16)[200]
    x = torch.randn(1, 4, 16, 8, 8, device="cuda", dtype=torch.bfloat16)
    n = torch.randn_like(x)
    want = (float(a ** 0.5) * x.float() + float((1 - a) ** 0.5) * n.float()).to(torch.bfloat16)
    assert torch.equal(pipe.scheduler.add_noise(x, n, torch.tensor([200], device="cuda")), want)


def test_stage_parity(setup):
    """Per-stage check so a failure localises: VAE moments, DiT velocity, decode."""
    pipe, (v, t, s), wv, wt, text = setup
    F, H, W = 9, 64, 64
    video = synth_clip(F, H, W, seed=7)
    ov, od = OracleVAE(v, wv), odit.OracleDiT(t, wt)
    p_ref = ov.encode(video)
    p = pipe.vae.encode(video.cuda().to(torch.bfloat16)).latent_dist.parameters.float().cpu()
    rel = float((p - p_ref).abs().max() / p_ref.abs().max())
    print(f"[stage] encode moments rel-max-err {rel:.4f} rms-rel {rms_rel(p, p_ref):.3e}")
    assert rel < 0.05 and rms_rel(p, p_ref) < 1.5e-2
    g = torch.Generator().manual_seed(3)
    hidden = torch.randn(1, 4, 16, H // 8, W // 8, generator=g)
    rope = odit.rope_3d(64, 2, H // 16, W // 16)
    ts = torch.tensor([399])
    v_ref = od.forward(hidden, text.float()[None], ts, rope)
    v_got = pipe.transformer(hidden_states=hidden.cuda().to(torch.bfloat16), encoder_hidden_states=text[None].cuda(),
                             timestep=ts.cuda(), image_rotary_emb=tuple(r.cuda() for r in rope), return_dict=False)[0].float().cpu()
    rel = float((v_got - v_ref).abs().max() / v_ref.abs().max())
    print(f"[stage] DiT velocity rel-max-err {rel:.4f} rms-rel {rms_rel(v_got, v_ref):.3e}")
    assert rel < 0.05 and rms_rel(v_got, v_ref) < 1.5e-2
    z = torch.randn(1, 16, 3, H // 8, W // 8, generator=g)
    d_ref = ov.decode(z)
    d = pipe.vae.decode(z.cuda().to(torch.bfloat16)).sample.float().cpu()
    rel = float((d - d_ref).abs().max() / d_ref.abs().max())
    print(f"[stage] decode rel-max-err {rel:.4f} rms-rel {rms_rel(d, d_ref):.3e}")
    assert rel < 0.05 and rms_rel(d, d_ref) < 1.5e-2


def test_vae_tiling_gpu(golden_dir):
    """--is_vae_st path (enable_tiling): HIP tiled encode/decode vs the oracle's restatement of diffusers' tiling; rms-relative
    gates of the un-tiled stage test (the blend adds no new rounding: one bf16 rounding per blended element)."""
    v, t, s = config.small_configs(num_layers=1)
    v["sample_height"], v["sample_width"] = 96, 160            # tiles 48x80 px so a 9x120x192 clip needs 3x3 tiles
    seed = 23
    pipe = CogVideoXPipeline.from_config(v, t, s, seed=seed, device="cuda")
    ov = OracleVAE(v, weights.random_state_dict(weights.vae_param_shapes(v), seed))
    video = synth_clip(9, 120, 192, seed=5)
    pipe.vae.enable_tiling()
    p = pipe.vae.encode(video.cuda().to(torch.bfloat16)).latent_dist.parameters.float().cpu()
    p_ref = ov.encode(video, tiling=True)
    rel = float((p - p_ref).abs().max() / p_ref.abs().max())
    print(f"[tiling] encode rel-max-err {rel:.4f} rms-rel {rms_rel(p, p_ref):.3e}")
    assert p.shape == p_ref.shape and rel < 0.05 and rms_rel(p, p_ref) < 1.5e-2
    z = torch.randn(1, 16, 3, 15, 24, generator=torch.Generator().manual_seed(2))
    d = pipe.vae.decode(z.cuda().to(torch.bfloat16)).sample.float().cpu()
    d_ref = ov.decode(z, tiling=True)
    rel = float((d - d_ref).abs().max() / d_ref.abs().max())
    print(f"[tiling] decode rel-max-err {rel:.4f} rms-rel {rms_rel(d, d_ref):.3e}")
    assert d.shape == d_ref.shape and rel < 0.05 and rms_rel(d, d_ref) < 1.5e-2


def test_vae_tiling_real_tile_geometry_gpu():
    """The tile geometry every published number of the reference ran with (`--is_vae_st`, inference.sh:8; sample size 480x720 ->
    240x360 px tiles, 30x45 latent tiles, strides 200x288 / 25x36, blends 40x72 px and 5x9 latents): a 5x288x432 clip = 2x2 tiles
    in both directions, against the oracle's tiled restatement.  Also: the un-tiled call must differ (tiling changes GroupNorm
    scope), i.e. the switch really took the tiled path."""
    v, t, s = config.small_configs(num_layers=1)
    assert (v["sample_height"], v["sample_width"]) == (480, 720)
    seed = 29
    pipe = CogVideoXPipeline.from_config(v, t, s, seed=seed, device="cuda")
    ov = OracleVAE(v, weights.random_state_dict(weights.vae_param_shapes(v), seed))
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    video = synth_clip(5, 288, 432, seed=6)
    xb = video.cuda().to(torch.bfloat16)
    plain = pipe.vae.encode(xb).latent_dist.parameters.float().cpu()
    pipe.vae.enable_tiling()
    t0 = time.time()
    p = pipe.vae.encode(xb).latent_dist.parameters.float().cpu()
    p_ref = ov.encode(video, tiling=True)
    rel = float((p - p_ref).abs().max() / p_ref.abs().max())
    print(f"[tiling 240x360] encode rel-max-err {rel:.4f} rms-rel {rms_rel(p, p_ref):.3e}; tiled vs un-tiled rms-rel {rms_rel(p, plain):.3e}")
    assert p.shape == p_ref.shape == (1, 32, 2, 36, 54)
    assert rel < 0.05 and rms_rel(p, p_ref) < 1.5e-2
    assert rms_rel(p, plain) > 4 * rms_rel(p, p_ref), "tiled and un-tiled encodes coincide: the tiled path did not run"
    z = torch.randn(1, 16, 2, 36, 54, generator=torch.Generator().manual_seed(2))
    d = pipe.vae.decode(z.cuda().to(torch.bfloat16)).sample.float().cpu()
    d_ref = ov.decode(z, tiling=True)
    rel = float((d - d_ref).abs().max() / d_ref.abs().max())
    print(f"[tiling 240x360] decode rel-max-err {rel:.4f} rms-rel {rms_rel(d, d_ref):.3e}  ({time.time() - t0:.0f} s incl. the oracle)")
    assert d.shape == d_ref.shape == (1, 3, 8, 288, 432)        # an even latent batch has no first-frame special case: 2 -> 8
    assert rel < 0.05 and rms_rel(d, d_ref) < 1.5e-2


def test_vae_tile_batching_bit_identical_gpu():
    """enable_tiling() runs all tiles of one shape as ONE batch (dove_conv_desc.nb and the *_nb GroupNorm / pool operators: per-tile conv
    caches, zero padding and GroupNorm scope inside one launch).  Per tile that is the arithmetic of the one-tile-at-a-time loop, so the
    two must agree bit for bit - on the production kernels: 128x128-px tiles (every level >= 16 rows: conv3x3_halo4x with fused
    statistics, the upsample variant, igemm_fast's strided / (3,1,1) convs, smallk), 17 frames = two frame-batches (caches in use),
    a 17x312x304 clip = 3x3 tiles in 4 shape classes (4, 2, 2, 1 tiles)."""
    v, t, s = config.small_configs(num_layers=1)
    v["sample_height"], v["sample_width"] = 256, 256
    pipe = CogVideoXPipeline.from_config(v, t, s, seed=31, device="cuda")
    vae = pipe.vae
    vae.enable_tiling()
    xb = synth_clip(17, 312, 304, seed=8).cuda().to(torch.bfloat16)
    z = torch.randn(1, 16, 5, 39, 38, generator=torch.Generator().manual_seed(3)).cuda().to(torch.bfloat16)
    assert vae.tile_batching
    p1 = vae.encode(xb).latent_dist.parameters
    d1 = vae.decode(z).sample
    vae.tile_batching = False
    p0 = vae.encode(xb).latent_dist.parameters
    d0 = vae.decode(z).sample
    vae.tile_batching = True
    assert p1.shape[:3] == (1, 32, 5) and d1.shape[:3] == (1, 3, 17)     # (diffusers' row / column limits do not divide this size evenly)
    assert bool(torch.isfinite(p1).all()) and bool(torch.isfinite(d1).all())
    assert torch.equal(p1, p0), f"batched tiled encode differs from the tile loop: max {float((p1.float() - p0.float()).abs().max())}"
    assert torch.equal(d1, d0), f"batched tiled decode differs from the tile loop: max {float((d1.float() - d0.float()).abs().max())}"
    # the frame-batches of a class on 1 / 2 / 3 streams (per-conv events order them; 25 frames = three frame-batches, 7 latent frames = three): the same bits
    xb3 = synth_clip(25, 312, 304, seed=9).cuda().to(torch.bfloat16)
    z3 = torch.randn(1, 16, 7, 39, 38, generator=torch.Generator().manual_seed(5)).cuda().to(torch.bfloat16)
    default = vae.tile_batch_streams
    outs = {}
    try:
        for k in (1, 2, 3):
            vae.tile_batch_streams = k
            outs[k] = (vae.encode(xb3).latent_dist.parameters.clone(), vae.decode(z3).sample.clone())
            torch.cuda.synchronize()
    finally:
        vae.tile_batch_streams = default
    for k in (2, 3):
        assert torch.equal(outs[k][0], outs[1][0]) and torch.equal(outs[k][1], outs[1][1]), f"tile_batch_streams = {k} changes bits"


def test_two_stream_vae_is_bit_identical(setup):
    """Alternate frame-batches on two HIP streams (per-conv events) must not change a single bit."""
    pipe, (v, t, s), wv, wt, text = setup
    video = synth_clip(17, 48, 80, seed=11).cuda().to(torch.bfloat16)
    z = torch.randn(1, 16, 5, 6, 10, generator=torch.Generator().manual_seed(4)).cuda().to(torch.bfloat16)
    outs = {}
    default = pipe.vae.n_streams
    for n in (1, 2):
        pipe.vae.n_streams = n
        outs[n] = (pipe.vae.encode(video).latent_dist.parameters.clone(), pipe.vae.decode(z).sample.clone())
        torch.cuda.synchronize()
    pipe.vae.n_streams = default
    assert torch.equal(outs[1][0], outs[2][0]) and torch.equal(outs[1][1], outs[2][1])


def test_fullsize_vae_causal_prefix_and_determinism(setup):
    """BASELINE configs[1] size (33x720x1280), through properties that do not need an oracle run at that size:
    (i) causal prefix - the VAE is causal in time and normalises per frame-batch, so a clip cut at a frame-batch boundary
        must reproduce the uncut result BIT-exactly: encode(clip[:17]) == encode(clip)[:5], decode(z[:3]) == decode(z)[:9]
        (same kernels, same shapes per batch; a conv cache or persistent-tile bug that leaks later frames breaks it);
    (ii) determinism - the same call twice gives identical bits (no atomics, no run-to-run scheduling dependence);
    (iii) outputs finite and the right shape."""
    pipe = setup[0]
    lr = synth_clip(33, 180, 320, seed=5)[0].cuda()                                   # [3,33,180,320] in [-1,1]
    clip = torch.nn.functional.interpolate(lr.permute(1, 0, 2, 3), size=(720, 1280), mode="bilinear",
                                           align_corners=False).permute(1, 0, 2, 3)[None].to(torch.bfloat16).contiguous()
    full = pipe.vae.encode(clip).latent_dist.parameters.clone()
    again = pipe.vae.encode(clip).latent_dist.parameters.clone()
    pre = pipe.vae.encode(clip[:, :, :17].contiguous()).latent_dist.parameters.clone()
    torch.cuda.synchronize()
    assert full.shape == (1, 32, 9, 90, 160) and pre.shape == (1, 32, 5, 90, 160)
    assert bool(torch.isfinite(full.float()).all())
    assert torch.equal(full, again), "encode is not deterministic"
    assert torch.equal(pre, full[:, :, :5]), "encoder output of a frame-batch-aligned prefix changed with later frames"
    z = (full[:, :16].float() * 0.7).to(torch.bfloat16).contiguous()
    dec = pipe.vae.decode(z).sample.clone()
    dec2 = pipe.vae.decode(z).sample.clone()
    dpre = pipe.vae.decode(z[:, :, :3].contiguous()).sample.clone()
    torch.cuda.synchronize()
    assert dec.shape == (1, 3, 33, 720, 1280) and dpre.shape == (1, 3, 9, 720, 1280)
    assert bool(torch.isfinite(dec.float()).all())
    assert torch.equal(dec, dec2), "decode is not deterministic"
    assert torch.equal(dpre, dec[:, :, :9]), "decoder output of a frame-batch-aligned prefix changed with later latents"



def test_from_pretrained_and_lora_through_hip(setup, tmp_path):
    """SURVEY 8(f) row 3 on the GPU: a DOVE-layout checkpoint directory (transformer as sharded fp32 safetensors + index
    json, /root/reference/finetune/scripts/prepare_sft_ckpt.py:43-69; vae single file; scheduler config) is loaded by
    `CogVideoXPipeline.from_pretrained`, packed for the kernels and run; then a rank-4 LoRA on to_q / to_k / to_v /
    to_out.0 (/root/reference/inference_script.py:616-621) is fused and `process_video` must follow the oracle run on
    W + (alpha/r) B A - and must NOT match the oracle on the un-adapted weights."""
    import json

    from safetensors.torch import save_file
    _, (v, t, s), wv, wt, text = setup
    root = str(tmp_path / "ckpt")
    for d in ("vae", "transformer", "scheduler"):
        os.makedirs(os.path.join(root, d))
    json.dump(v, open(os.path.join(root, "vae", "config.json"), "w"))
    json.dump(t, open(os.path.join(root, "transformer", "config.json"), "w"))
    json.dump(dict(s, _class_name="CogVideoXDPMScheduler", _diffusers_version="0.32.0"),
              open(os.path.join(root, "scheduler", "scheduler_config.json"), "w"))
    save_file({k: x.contiguous() for k, x in wv.items()}, os.path.join(root, "vae", "diffusion_pytorch_model.safetensors"))
    keys = list(wt)
    cuts = [0, len(keys) // 3, 2 * len(keys) // 3, len(keys)]
    wmap = {}
    for i in range(3):
        fn = f"diffusion_pytorch_model-{i + 1:05d}-of-00003.safetensors"
        save_file({k: wt[k].float().contiguous() for k in keys[cuts[i]:cuts[i + 1]]}, os.path.join(root, "transformer", fn))
        wmap.update({k: fn for k in keys[cuts[i]:cuts[i + 1]]})
    json.dump({"metadata": {"total_size": 0}, "weight_map": wmap},
              open(os.path.join(root, "transformer", "diffusion_pytorch_model.safetensors.index.json"), "w"))
    pipe = CogVideoXPipeline.from_pretrained(root, torch_dtype=torch.bfloat16).to("cuda")
    F, H, W = 9, 64, 64
    video = synth_clip(F, H, W, seed=17)
    noise = torch.randn(1, 16, 3, H // 8, W // 8, generator=torch.Generator().manual_seed(6))
    run = lambda p: process_video(p, video.cuda(), empty_prompt_embedding=text, posterior_noise=noise.cuda()).float().cpu()  # noqa: E731
    got0 = run(pipe)
    ref0 = odit.process_video(OracleVAE(v, wv), odit.OracleDiT(t, wt), s, video, text.float()[None], noise)
    p0 = psnr(got0, ref0)
    # adapter: rank 4, alpha 8 (scale alpha/r = 2), strong enough to move the output well above the bf16 noise floor
    D = t["num_attention_heads"] * t["attention_head_dim"]
    g = torch.Generator().manual_seed(8)
    lora, wt2 = {}, dict(wt)
    for i in range(t["num_layers"]):
        for m in ("to_q", "to_k", "to_v", "to_out.0"):
            mod = f"transformer_blocks.{i}.attn1.{m}"
            A = torch.randn(4, D, generator=g) * D ** -0.5
            B = torch.randn(D, 4, generator=g) * 0.5
            lora[f"transformer.{mod}.lora_A.weight"], lora[f"transformer.{mod}.lora_B.weight"] = A, B
            wt2[mod + ".weight"] = wt[mod + ".weight"] + 2.0 * (B @ A)
    os.makedirs(tmp_path / "lora")
    save_file(lora, str(tmp_path / "lora" / "pytorch_lora_weights.safetensors"),
              metadata={"lora_adapter_metadata": json.dumps({"r": 4, "lora_alpha": 8})})
    pipe.load_lora_weights(str(tmp_path / "lora"), weight_name="pytorch_lora_weights.safetensors", adapter_name="test_1")
    pipe.fuse_lora(components=["transformer"], lora_scale=1.0)
    got1 = run(pipe)
    ref1 = odit.process_video(OracleVAE(v, wv), odit.OracleDiT(t, wt2), s, video, text.float()[None], noise)
    p1, p_cross = psnr(got1, ref1), psnr(got1, ref0)
    print(f"[ckpt+lora] PSNR loaded ckpt vs oracle {p0:.2f} dB; fused LoRA vs oracle(W+2BA) {p1:.2f} dB; "
          f"fused vs oracle(W) {p_cross:.2f} dB")
    assert p0 > 35.0 and p1 > 35.0, (p0, p1)
    assert p_cross < p1 - 6.0, (p_cross, p1)                 # the adapter really changed what the kernels compute


def test_diffusers_stage_goldens(golden_dir):
    """Consumes tests/golden/diffusers_stages.safetensors (tools/capture_goldens.py, run where diffusers + the DOVE
    checkpoint exist) when present: the REAL reference's per-stage tensors pin both the oracle and the HIP path.  Needs
    the same checkpoint here (DOVE_MODEL_PATH); skipped otherwise - until someone provides both, model-arithmetic
    parity stays 'pinned vs the clean-room oracle, unpinned vs diffusers' (DESIGN.md section 2)."""
    from safetensors.torch import load_file
    path = os.path.join(golden_dir, "diffusers_stages.safetensors")
    model = os.environ.get("DOVE_MODEL_PATH", "")
    if not os.path.exists(path) or not os.path.isdir(model):
        pytest.skip("diffusers_stages.safetensors and/or DOVE_MODEL_PATH not available")
    gold = load_file(path)
    pipe = CogVideoXPipeline.from_pretrained(model, torch_dtype=torch.bfloat16)
    video, text, noise = gold["video"], gold["text"][0].to(torch.bfloat16), gold["noise"]
    p = pipe.vae.encode(video.cuda().to(torch.bfloat16)).latent_dist.parameters
    assert rms_rel(p, gold["moments"]) < 2e-2
    rope = (gold["rope_cos"].cuda(), gold["rope_sin"].cuda())
    ts = torch.tensor([399], device="cuda")
    vel = pipe.transformer(hidden_states=gold["latent"].cuda().to(torch.bfloat16), encoder_hidden_states=text[None].cuda(),
                           timestep=ts, image_rotary_emb=rope, return_dict=False)[0]
    assert rms_rel(vel, gold["velocity"]) < 6e-2
    out = process_video(pipe, video.cuda(), empty_prompt_embedding=text, posterior_noise=noise.cuda())
    assert psnr(out.float().cpu(), gold["sr"]) > 35.0


def test_long_clip_chunked_tiled_configs3(golden_dir):
    """BASELINE configs[3] on ONE GPU: a 129x1088x1920 clip (the script's padding of 1080p) through the reference's chunk loop
    (`--chunk_len 33 --overlap_t 8`: chunks (0,33),(25,58),(50,83),(75,129), ref :249-279) with the FULL 42-layer model and
    diffusers' VAE spatial tiling switched on (`--is_vae_st`, 240x360-px tiles with linear blends) - "tiled VAE + chunked DiT".
    No oracle run exists at this size; checked: exact-once coverage of the stitched clip (the reference's own check, ref
    :705-729), finiteness, the first chunk's kept region == `process_video` on that chunk alone (bit-exact: chunks are
    independent calls), and that tiling really ran (a tiled decode differs from the untiled one)."""
    from safetensors.torch import load_file

    from dove_amd import tiling
    from dove_amd.inference import run_clip
    v, t, s = config.default_configs()
    pipe = CogVideoXPipeline.from_config(v, t, s, seed=1234, device="cuda", init_device="cuda")
    pipe.vae.enable_tiling()
    text = load_file(os.path.join(golden_dir, "empty_prompt_embedding.safetensors"))["prompt_embedding"]
    F, H, W = 129, 1088, 1920
    lr = synth_clip(F, H // 8, W // 8, seed=9)[0].cuda()                              # cheap content, upsampled x8
    video = torch.nn.functional.interpolate(lr.permute(1, 0, 2, 3), size=(H, W), mode="bilinear",
                                            align_corners=False).permute(1, 0, 2, 3)[None].to(torch.bfloat16).contiguous()
    torch.cuda.reset_peak_memory_stats()
    import time
    t0 = time.time()
    gen = torch.Generator(device="cuda").manual_seed(5)
    out, wc = run_clip(pipe, video, chunk_len=33, overlap_t=8, empty_prompt_embedding=text, generator=gen)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f"[configs3] 129x1088x1920, 4 chunks, tiled VAE, 42 layers: {dt:.1f} s ({F / dt:.1f} frames/s), "
          f"peak HBM {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
    assert out.shape == (1, 3, F, H, W) and int(wc.min()) == 1 and int(wc.max()) == 1
    assert bool(torch.isfinite(out).all())
    items = tiling.plan(video.shape, 33, 8, (0, 0), (32, 32))
    assert [it[0][:2] for it in items] == [(0, 33), (25, 58), (50, 83), (75, 129)]
    (t0_, t1_, h0, h1, w0, w1), region = items[0]
    gen = torch.Generator(device="cuda").manual_seed(5)
    piece = process_video(pipe, video[:, :, t0_:t1_], empty_prompt_embedding=text, generator=gen).float().cpu()
    ref_out, ref_wc = torch.zeros_like(out), torch.zeros_like(wc)
    tiling.stitch(ref_out, ref_wc, piece, region)
    kept = ref_wc.bool()
    assert torch.equal(out[kept], ref_out[kept]), "chunk 0 of the stitched clip differs from process_video on that chunk"
    pipe.vae.disable_tiling()
    gen = torch.Generator(device="cuda").manual_seed(5)
    untiled = process_video(pipe, video[:, :, :33], empty_prompt_embedding=text, generator=gen).float().cpu()
    assert not torch.equal(untiled, piece), "enable_tiling() had no effect at 1088x1920"


def test_cli_random_init_npy_and_png(golden_dir, tmp_path):
    """python -m dove_amd.cli with the reference's flags (ref :506-778) on a tiny .npy clip: pad to 8N+1 frames / multiples of 16,
    x4 upscale, one process_video, un-pad, uint8 frames out (.npy and --png_save), coverage check inside."""
    import numpy as np

    from dove_amd import cli
    inp, out = tmp_path / "in", tmp_path / "out"
    inp.mkdir()
    rng = np.random.default_rng(0)
    clip = rng.integers(0, 256, size=(7, 20, 28, 3), dtype=np.uint8)            # 7 frames -> padded to 9; 80x112 -> 80x112 (x4)
    np.save(inp / "clip0.npy", clip)
    emb = os.path.join(golden_dir, "empty_prompt_embedding.safetensors")
    common = ["--input_dir", str(inp), "--random_init", "--num_layers", "2", "--prompt_embedding", emb]
    cli.main(common + ["--output_path", str(out)])
    res = np.load(out / "clip0.npy")
    assert res.shape == (7, 80, 112, 3) and res.dtype == np.uint8 and res.std() > 0
    cli.main(common + ["--output_path", str(out), "--png_save"])
    assert len(list((out / "clip0").glob("*.png"))) == 7
    # the rest of the reference's parser (ref :507-554): accepted with the reference's meaning
    out2 = tmp_path / "out2"
    cli.main(common + ["--output_path", str(out2), "--upscale_mode", "bicubic", "--is_cpu_offload", "--fps", "24", "--save_format", "yuv420p",
                       "--gt_dir", str(out), "--eval_metrics", "psnr"])
    res2 = np.load(out2 / "clip0.npy")
    assert res2.shape == res.shape and not np.array_equal(res2, res)          # bicubic input differs from bilinear input
    with pytest.raises(ValueError, match="bfloat16"):
        cli.main(common + ["--output_path", str(out2), "--dtype", "float16"])
    with pytest.raises(NotImplementedError, match="pyiqa"):
        cli.main(common + ["--output_path", str(out2), "--eval_metrics", "psnr,lpips", "--gt_dir", str(out)])
    # the torch route of --upscale_mode (pad, F.interpolate, normalise) agrees with the fused HIP kernel on its own mode
    from dove_amd import prepost
    fr = torch.from_numpy(clip)
    a = prepost.preprocess_frames(fr, 4)[0]
    b = prepost.preprocess_frames_torch(fr.cuda(), 2, 12, 4, 4, "bilinear", torch.bfloat16)[None]
    assert a.shape == b.shape and float((a.float() - b.float()).abs().max()) <= 2 ** -7
