"""-m gpu: T5 text-encoder path on the HIP operators (ref :429-444) against transformers' own T5EncoderModel (the real
reference implementation of this component, fp32 on CPU) and the non-empty-prompt branch of process_video end to end."""
import pytest
import torch

import emu_ops as E
from dove_amd import config, ops, t5 as dt5, weights
from dove_amd.inference import process_video
from dove_amd.pipeline import CogVideoXPipeline
from oracle import dit as odit
from oracle.vae import OracleVAE
from test_t5_cpu import SMALL, hf_model

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def close(name, got, ref, rtol=1.6e-2, afrac=4e-3):
    got, ref = got.float().cpu(), ref.float().cpu()
    err = (got - ref).abs()
    tol = rtol * ref.abs() + afrac * ref.abs().max() + 1e-6
    assert got.shape == ref.shape and torch.isfinite(got).all() and not bool((err > tol).any()), \
        f"{name}: max err {float(err.max()):.4g} (ref max {float(ref.abs().max()):.4g})"


def test_t5_operators():
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(226, 4096, generator=g) * 3).to(BF)
    w = 1 + 0.2 * torch.randn(4096, generator=g)
    close("rmsnorm", ops.rmsnorm(x.cuda(), w.cuda(), 1e-6), E.rmsnorm(x, w, 1e-6))
    y = torch.randn(226, 2 * 1024, generator=g).to(BF)
    close("gated_gelu", ops.gated_gelu(y.cuda()), E.gated_gelu(y))
    for N, H in ((226, 64), (77, 4), (300, 2)):
        qkv = (torch.randn(N, 3 * H * 64, generator=g) * 0.5).to(BF)
        bias = torch.randn(H, N, N, generator=g) * 2
        # P is rounded to bf16 on both sides: 2 ulp
        close(f"attention_bias_{N}_{H}", ops.attention_bias(qkv.cuda(), bias.cuda(), H), E.attention_bias(qkv, bias, H), rtol=3e-2, afrac=8e-3)


@pytest.mark.parametrize("name,cfg", [("small", SMALL),
                                      ("xxl_width_2_layers", dict(dt5.T5_XXL_CONFIG, vocab_size=1000, num_layers=2))])
def test_t5_encoder_vs_transformers(name, cfg):
    m = hf_model(cfg)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    enc = dt5.T5EncoderModel(cfg, sd, "cuda")
    ids = torch.randint(0, cfg["vocab_size"], (1, 226), generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        want = m(input_ids=ids)[0]
        want_bf = m.to(BF)(input_ids=ids)[0].float()                 # what the reference's bf16 text encoder returns
    got = enc(ids.cuda())[0].float().cpu()
    rms = lambda a: float((a - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt())   # noqa: E731
    print(f"[t5 {name}] rms-rel vs transformers fp32: HIP {rms(got):.3e}, transformers bf16 {rms(want_bf):.3e}")
    assert got.shape == want.shape
    assert rms(got) <= 1.25 * rms(want_bf) + 1e-3


class _FakeTokenizer:
    """Stands in for transformers' T5 tokenizer (needs the checkpoint's sentencepiece model): deterministic ids per prompt."""

    def __call__(self, prompt, padding=None, max_length=226, truncation=True, add_special_tokens=True, return_tensors="pt"):
        g = torch.Generator().manual_seed(sum(map(ord, prompt)))
        n = min(len(prompt.split()) + 1, max_length)
        ids = torch.zeros(1, max_length, dtype=torch.long)            # pad id 0, like T5
        ids[0, :n] = torch.randint(2, 1000, (n,), generator=g)
        ids[0, n - 1] = 1                                              # </s>
        return type("Enc", (), {"input_ids": ids})()


def test_process_video_with_text_prompt():
    """Non-empty prompt: tokenizer -> T5 encoder (HIP) -> DiT, vs the oracle fed with transformers' fp32 T5 output."""
    v, t, s = config.small_configs(num_layers=2)
    seed = 29
    cfg5 = dict(dt5.T5_XXL_CONFIG, vocab_size=1000, num_layers=2)
    m = hf_model(cfg5)
    sd5 = {k: x.detach().clone() for k, x in m.state_dict().items()}
    pipe = CogVideoXPipeline.from_config(v, t, s, seed=seed, device="cuda")
    pipe.text_encoder = dt5.T5EncoderModel(cfg5, sd5, "cuda")
    pipe.tokenizer = _FakeTokenizer()
    wv = weights.random_state_dict(weights.vae_param_shapes(v), seed)
    wt = weights.random_state_dict(weights.dit_param_shapes(t), seed)
    g = torch.Generator().manual_seed(4)
    F, H, W = 9, 64, 64
    video = torch.rand(1, 3, F, H, W, generator=g) * 2 - 1
    noise = torch.randn(1, 16, 3, H // 8, W // 8, generator=g)
    prompt = "a clean sharp video of a red fox running through fresh snow"
    got = process_video(pipe, video.cuda(), prompt=prompt, posterior_noise=noise.cuda()).float().cpu()
    empty = process_video(pipe, video.cuda(), prompt="", empty_prompt_embedding=torch.zeros(226, 4096, dtype=BF),
                          posterior_noise=noise.cuda()).float().cpu()
    with torch.no_grad():
        text = m(input_ids=_FakeTokenizer()(prompt).input_ids)[0]
    ref = odit.process_video(OracleVAE(v, wv), odit.OracleDiT(t, wt), s, video, text, noise)
    mse = lambda a, b: ((a - b) ** 2).flatten(3).mean(-1)     # noqa: E731
    p = float((10 * torch.log10(1.0 / (mse(got, ref) + 1e-8))).mean())
    p_empty = float((10 * torch.log10(1.0 / (mse(empty, ref) + 1e-8))).mean())
    print(f"[prompt] PSNR(hip with T5 prompt, oracle with transformers T5) {p:.2f} dB; zero-text run vs the same oracle {p_empty:.2f} dB")
    assert p > 35.0 and p > p_empty + 3.0
