"""-m gpu: the graph-level C entry points (dove_create ... dove_sr_clip) against the Python facade.  Both drive the
same operator kernels in the same order, so with the host-computed tables handed over (RoPE, timestep projection) every stage
must agree BIT FOR BIT; with the tables computed inside the library (host libm) the result may differ in the last bf16 bit."""
import time

import pytest
import torch

from dove_amd import config, weights
from dove_amd.graph import GraphContext
from dove_amd.inference import process_video
from dove_amd.pipeline import CogVideoXPipeline
from dove_amd.rope import prepare_rotary_positional_embeddings

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def both():
    v, t, s = config.small_configs(num_layers=2)
    seed = 31
    wv = weights.random_state_dict(weights.vae_param_shapes(v), seed)
    wt = weights.random_state_dict(weights.dit_param_shapes(t), seed)
    pipe = CogVideoXPipeline.from_config(v, t, s, seed=seed, device="cuda")
    ctx = GraphContext(v, t, wv, wt, "cuda")
    return pipe, ctx, (v, t, s, wv, wt)


def rope_for(pipe, T, h, w):
    return prepare_rotary_positional_embeddings(height=h * 8, width=w * 8, num_frames=T, transformer_config=pipe.transformer.config,
                                                vae_scale_factor_spatial=8, device="cuda")


@pytest.mark.parametrize("F,H,W", [(9, 64, 64), (17, 48, 80), (25, 32, 48)])
def test_stages_bit_exact(both, F, H, W):
    pipe, ctx, _ = both
    g = torch.Generator().manual_seed(F)
    video = (torch.rand(1, 3, F, H, W, generator=g) * 2 - 1).to(BF).cuda()
    m_py = pipe.vae.encode(video).latent_dist.parameters[0]
    m_c = ctx.vae_encode(video[0].contiguous())
    torch.cuda.synchronize()
    assert torch.equal(m_c, m_py), f"encode differs: {float((m_c.float() - m_py.float()).abs().max())}"
    T = m_py.shape[1]
    Td = T + T % 2
    hidden = torch.randn(Td, 16, H // 8, W // 8, generator=g).to(BF).cuda()
    text = (torch.randn(226, 4096, generator=g) * 0.15).to(BF).cuda()
    rope = rope_for(pipe, Td, H // 8, W // 8)
    ts = torch.tensor([399], device="cuda")
    v_py = pipe.transformer(hidden_states=hidden[None], encoder_hidden_states=text[None], timestep=ts, image_rotary_emb=rope, return_dict=False)[0][0]
    v_c = ctx.dit_forward(hidden, text, 399, rope=rope, timestep_proj=pipe.transformer.timestep_projection(399))
    torch.cuda.synchronize()
    assert torch.equal(v_c, v_py), f"DiT differs: {float((v_c.float() - v_py.float()).abs().max())}"
    v_own = ctx.dit_forward(hidden, text, 399)                  # tables from the library's own host math
    rel = float((v_own.float() - v_py.float()).abs().max() / v_py.float().abs().max())
    assert rel < 2e-2, rel
    z = torch.randn(16, T, H // 8, W // 8, generator=g).to(BF).cuda()
    d_py = pipe.vae.decode(z[None], _range01=True, _prescale=1 / 0.7).sample[0]
    d_c = ctx.vae_decode(z, prescale=1 / 0.7, range01=True)
    torch.cuda.synchronize()
    assert torch.equal(d_c, d_py), f"decode differs: {float((d_c.float() - d_py.float()).abs().max())}"
    z2 = z[:, :2].contiguous()                                  # an even number of latent frames decodes to 8 frames, not 5
    assert torch.equal(ctx.vae_decode(z2), pipe.vae.decode(z2[None]).sample[0])


def test_sr_clip_equals_process_video(both):
    pipe, ctx, _ = both
    g = torch.Generator().manual_seed(77)
    F, H, W = 9, 64, 96
    video = (torch.rand(1, 3, F, H, W, generator=g) * 2 - 1).to(BF).cuda()
    noise = torch.randn(1, 16, 3, H // 8, W // 8, generator=g).cuda()
    text = (torch.randn(226, 4096, generator=g) * 0.15).to(BF).cuda()
    want = process_video(pipe, video, empty_prompt_embedding=text, posterior_noise=noise)[0]
    sa, s1 = pipe.scheduler._coeffs(torch.tensor([399]), BF)
    rope = rope_for(pipe, 4, H // 8, W // 8)
    got = ctx.sr_clip(video[0].contiguous(), noise[0].contiguous(), text, 399, sa, s1, rope=rope,
                      timestep_proj=pipe.transformer.timestep_projection(399))
    torch.cuda.synchronize()
    assert got.shape == want.shape == (3, F, H, W)
    assert torch.equal(got, want), f"sr_clip differs from process_video: {float((got.float() - want.float()).abs().max())}"
    # the whole clip again with nothing precomputed by the host
    own = ctx.sr_clip(video[0].contiguous(), noise[0].contiguous(), text, 399, sa, s1)
    d = (own.float() - want.float()).abs()                     # libm vs torch in the sinusoid / RoPE tables: a last-bit change of a few
    assert float(d.max()) < 0.1 and float(d.mean()) < 5e-3, (float(d.max()), float(d.mean()))   # table entries, amplified by random weights


def _clip_inputs(F, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    video = (torch.rand(1, 3, F, H, W, generator=g) * 2 - 1).to(BF).cuda()
    T = 1 + (F - 1) // 4
    noise = torch.randn(1, 16, T, H // 8, W // 8, generator=g).cuda()
    text = (torch.randn(226, 4096, generator=g) * 0.15).to(BF).cuda()
    return video, noise, text, T


def test_option_weight_sums_off_bit_exact(both):
    """DOVE_OPT_WEIGHT_SUMS = 0 == pipe.vae.weight_sums = False: no conv is handed a pack-time weight sum (w_first / w_sub / w_pair) - the
    reference's per-tap arithmetic for validating a checkpoint.  Decode (which holds all three forms: cache-less first frames, sub-pixel
    upsamplers, frame pairs behind the time-doubling upsamplers) through the C graph == the Python facade with the same switch, bit for
    bit; the switch does change the result (so it did switch), by no more than the summed weights' one extra rounding."""
    from dove_amd import lib as L
    pipe, ctx, _ = both
    T, h, w = 3, 32, 48                                          # the upsampled grids reach 16 x 32 low-res: the sub-pixel and pair forms engage
    z = torch.randn(16, T, h, w, generator=torch.Generator().manual_seed(8)).to(BF).cuda()
    try:
        on_py = pipe.vae.decode(z[None], _prescale=1 / 0.7).sample[0]
        on_c = ctx.vae_decode(z, prescale=1 / 0.7)
        pipe.vae.weight_sums = False
        ctx.set_option(L.OPT_WEIGHT_SUMS, 0)
        assert ctx.get_option(L.OPT_WEIGHT_SUMS) == 0
        off_py = pipe.vae.decode(z[None], _prescale=1 / 0.7).sample[0]
        off_c = ctx.vae_decode(z, prescale=1 / 0.7)
        torch.cuda.synchronize()
        assert torch.equal(on_c, on_py) and torch.equal(off_c, off_py)
        d = float(((on_py.float() - off_py.float()) ** 2).mean().sqrt() / (off_py.float() ** 2).mean().sqrt())
        print(f"[weight sums off] decoded clip, summed vs per-tap weights: rms-rel {d:.3e}")
        assert 0.0 < d < 3e-2, d
    finally:
        pipe.vae.weight_sums = True
        ctx.set_option(L.OPT_WEIGHT_SUMS, 1)


def test_option_vae_streams_bit_exact(both):
    """DOVE_OPT_VAE_STREAMS = pipe.vae.n_streams: 2 (the default on both sides) alternates the frame-batches of a stage between the caller's
    stream and an internal one, ordered by one event per causal conv, with the arena tracking which stream released each block; 1 runs them
    one after the other.  Same kernels on the same inputs: encode (33 frames = 4 batches) and decode (9 latent frames = 4 batches) must not
    change a bit between the two, on either side of the C boundary, three times in a row (a missed cross-stream ordering would show as a
    flaky difference, not a steady one)."""
    from dove_amd import lib as L
    pipe, ctx, _ = both
    g = torch.Generator().manual_seed(21)
    video = (torch.rand(3, 33, 64, 96, generator=g) * 2 - 1).to(BF).cuda()
    z = torch.randn(16, 9, 16, 24, generator=g).to(BF).cuda()
    assert ctx.get_option(L.OPT_VAE_STREAMS) == 2 and pipe.vae.n_streams == 2
    try:
        ctx.set_option(L.OPT_VAE_STREAMS, 1)
        pipe.vae.n_streams = 1
        m1, d1 = ctx.vae_encode(video), ctx.vae_decode(z, prescale=1 / 0.7, range01=True)
        mp, dp = pipe.vae.encode(video[None]).latent_dist.parameters[0], pipe.vae.decode(z[None], _prescale=1 / 0.7, _range01=True).sample[0]
        torch.cuda.synchronize()
        assert torch.equal(m1, mp) and torch.equal(d1, dp)
        ctx.set_option(L.OPT_VAE_STREAMS, 2)
        pipe.vae.n_streams = 2
        for rep in range(3):
            m2, d2 = ctx.vae_encode(video), ctx.vae_decode(z, prescale=1 / 0.7, range01=True)
            mq, dq = pipe.vae.encode(video[None]).latent_dist.parameters[0], pipe.vae.decode(z[None], _prescale=1 / 0.7, _range01=True).sample[0]
            torch.cuda.synchronize()
            assert torch.equal(m2, m1) and torch.equal(d2, d1), f"C graph: two streams changed the result (repetition {rep})"
            assert torch.equal(mq, m1) and torch.equal(dq, d1), f"facade: two streams changed the result (repetition {rep})"
    finally:
        ctx.set_option(L.OPT_VAE_STREAMS, 2)
        pipe.vae.n_streams = 2


def test_option_vae_tiling_bit_exact(both):
    """DOVE_OPT_VAE_TILING = pipe.vae.enable_tiling() (`--is_vae_st`, ref :643-645): encode, decode and the whole clip through the C
    graph == the Python facade with the same switch, bit for bit; 3 x 3 tiles in both directions (sample size 96 x 160 -> 48 x 80 px
    tiles on a 9 x 112 x 192 clip), ragged last tiles."""
    from dove_amd import lib as L
    pipe, ctx, _ = both
    F, H, W = 9, 112, 192
    video, noise, text, T = _clip_inputs(F, H, W, 5)
    cfg = pipe.vae.config
    old = (cfg["sample_height"], cfg["sample_width"], pipe.vae.use_tiling)
    try:
        plain = ctx.vae_encode(video[0].contiguous())
        cfg["sample_height"], cfg["sample_width"] = 96, 160
        ctx.set_option(L.OPT_VAE_SAMPLE_HEIGHT, 96)
        ctx.set_option(L.OPT_VAE_SAMPLE_WIDTH, 160)
        pipe.vae.enable_tiling()
        ctx.enable_tiling()
        assert ctx.get_option(L.OPT_VAE_TILING) == 1 and ctx.get_option(L.OPT_VAE_SAMPLE_WIDTH) == 160
        m_py = pipe.vae.encode(video).latent_dist.parameters[0]
        m_c = ctx.vae_encode(video[0].contiguous())
        torch.cuda.synchronize()
        assert torch.equal(m_c, m_py), f"tiled encode differs: {float((m_c.float() - m_py.float()).abs().max())}"
        assert not torch.equal(m_c, plain), "the option did not switch the tiled path on"
        z = torch.randn(16, T, H // 8, W // 8, generator=torch.Generator().manual_seed(6)).to(BF).cuda()
        d_py = pipe.vae.decode(z[None], _range01=True, _prescale=1 / 0.7).sample[0]
        d_c = ctx.vae_decode(z, prescale=1 / 0.7, range01=True)
        torch.cuda.synchronize()
        assert torch.equal(d_c, d_py), f"tiled decode differs: {float((d_c.float() - d_py.float()).abs().max())}"
        want = process_video(pipe, video, empty_prompt_embedding=text, posterior_noise=noise)[0]
        sa, s1 = pipe.scheduler._coeffs(torch.tensor([399]), BF)
        got = ctx.sr_clip(video[0].contiguous(), noise[0].contiguous(), text, 399, sa, s1, rope=rope_for(pipe, T + T % 2, H // 8, W // 8),
                          timestep_proj=pipe.transformer.timestep_projection(399))
        torch.cuda.synchronize()
        assert torch.equal(got, want), f"tiled sr_clip differs: {float((got.float() - want.float()).abs().max())}"
    finally:
        cfg["sample_height"], cfg["sample_width"], pipe.vae.use_tiling = old
        ctx.set_option(L.OPT_VAE_SAMPLE_HEIGHT, old[0])
        ctx.set_option(L.OPT_VAE_SAMPLE_WIDTH, old[1])
        ctx.enable_tiling(False)


def test_option_noise_step_bit_exact(both):
    """`--noise_step` (ref :449-457) through dove_sr_clip's pre_noise argument == process_video(noise_step=200) with the same eps."""
    pipe, ctx, _ = both
    F, H, W = 9, 64, 96
    video, noise, text, T = _clip_inputs(F, H, W, 8)
    torch.manual_seed(4242)
    want = process_video(pipe, video, noise_step=200, empty_prompt_embedding=text, posterior_noise=noise)[0]
    torch.manual_seed(4242)
    Td = T + T % 2
    eps = torch.randn(1, Td, 16, H // 8, W // 8, device="cuda", dtype=BF)         # the one draw process_video makes (randn_like(latent))
    sa, s1 = pipe.scheduler._coeffs(torch.tensor([399]), BF)
    na, n1 = pipe.scheduler._coeffs(torch.tensor([200]), BF)
    kw = dict(rope=rope_for(pipe, Td, H // 8, W // 8), timestep_proj=pipe.transformer.timestep_projection(399))
    got = ctx.sr_clip(video[0].contiguous(), noise[0].contiguous(), text, 399, sa, s1, pre_noise=(eps[0], na, n1), **kw)
    base = ctx.sr_clip(video[0].contiguous(), noise[0].contiguous(), text, 399, sa, s1, **kw)
    torch.cuda.synchronize()
    assert torch.equal(got, want), f"pre-noised sr_clip differs: {float((got.float() - want.float()).abs().max())}"
    assert not torch.equal(got, base)
    got32 = ctx.sr_clip(video[0].contiguous(), noise[0].contiguous(), text, 399, sa, s1, pre_noise=(eps[0].float(), na, n1), **kw)
    assert torch.equal(got32, want), "fp32 eps must be rounded to the latent dtype first"


def test_option_mxfp8_bit_exact(both):
    """DOVE_OPT_DIT_LINEAR_MXFP8 + DOVE_OPT_DIT_ATTN_MXFP8 (BASELINE configs[4]) through the C graph == the facade's mxfp8 variant."""
    from dove_amd import lib as L
    from dove_amd.transformer import CogVideoXTransformer3DModel
    pipe, _, (v, t, s, wv, wt) = both
    tr8 = CogVideoXTransformer3DModel(t, wt, "cuda", linear_precision="mxfp8", attention_precision="mxfp8")
    pipe8 = CogVideoXPipeline(pipe.vae, tr8, pipe.scheduler)
    ctx8 = GraphContext(v, t, wv, wt, "cuda", dit_linear_precision="mxfp8", dit_attention_precision="mxfp8")
    assert ctx8.get_option(L.OPT_DIT_LINEAR_MXFP8) == 1 and ctx8.get_option(L.OPT_DIT_ATTN_MXFP8) == 1
    with pytest.raises(RuntimeError, match="before dove_finalize_weights"):
        ctx8.set_option(L.OPT_DIT_LINEAR_MXFP8, 0)
    F, H, W = 9, 64, 96
    video, noise, text, T = _clip_inputs(F, H, W, 9)
    Td = T + T % 2
    g = torch.Generator().manual_seed(10)
    hidden = torch.randn(Td, 16, H // 8, W // 8, generator=g).to(BF).cuda()
    rope = rope_for(pipe, Td, H // 8, W // 8)
    ts = torch.tensor([399], device="cuda")
    v_py = tr8(hidden_states=hidden[None], encoder_hidden_states=text[None], timestep=ts, image_rotary_emb=rope, return_dict=False)[0][0]
    v_c = ctx8.dit_forward(hidden, text, 399, rope=rope, timestep_proj=tr8.timestep_projection(399))
    v_16 = pipe.transformer(hidden_states=hidden[None], encoder_hidden_states=text[None], timestep=ts, image_rotary_emb=rope, return_dict=False)[0][0]
    torch.cuda.synchronize()
    assert torch.equal(v_c, v_py), f"mxfp8 DiT differs: {float((v_c.float() - v_py.float()).abs().max())}"
    assert not torch.equal(v_c, v_16)
    want = process_video(pipe8, video, empty_prompt_embedding=text, posterior_noise=noise)[0]
    sa, s1 = pipe.scheduler._coeffs(torch.tensor([399]), BF)
    got = ctx8.sr_clip(video[0].contiguous(), noise[0].contiguous(), text, 399, sa, s1, rope=rope, timestep_proj=tr8.timestep_projection(399))
    torch.cuda.synchronize()
    assert torch.equal(got, want), f"mxfp8 sr_clip differs: {float((got.float() - want.float()).abs().max())}"


def test_failed_stage_does_not_leak_the_arena(both):
    """A stage that returns early on an error used to leave its arena blocks live for ever (the workspace shrank for every later
    call); now every top-level call starts from an empty arena: an exhausted call followed by a regrown workspace succeeds."""
    pipe, ctx, _ = both
    F, H, W = 9, 64, 64
    video = (torch.rand(3, F, H, W, generator=torch.Generator().manual_seed(3)) * 2 - 1).to(BF).cuda()
    want = ctx.vae_encode(video)
    small = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
    ctx.set_workspace(small.numel(), buffer=small)              # lent memory, far too small: the stage must fail cleanly ...
    for _ in range(3):                                          # ... every time, without consuming what little there is
        with pytest.raises(RuntimeError, match="workspace exhausted"):
            ctx.vae_encode(video)
    need = ctx.workspace_bytes(F, H, W)
    big = torch.empty(need, dtype=torch.uint8, device="cuda")
    ctx.set_workspace(need, buffer=big)                         # the same context on enough lent memory: nothing stale blocks it
    got = ctx.vae_encode(video)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    ctx.set_workspace(1 << 20)                                  # a library-owned arena that is too small is regrown, not an error
    assert torch.equal(ctx.vae_encode(video), want)


def test_sr_clip_full_size_timing():
    """Headline clip through ONE C call: same kernels, no Python between them, activations from the library's arena."""
    v, t, s = config.default_configs()
    seed = 1234
    wv = weights.LazyStateDict(weights.vae_param_shapes(v), seed, "cuda")
    wt = weights.LazyStateDict(weights.dit_param_shapes(t), seed, "cuda")     # fp32, like the facade reads them (22 GB until finalize)
    ctx = GraphContext(v, t, wv, wt, "cuda")
    pipe = CogVideoXPipeline.from_config(v, t, s, seed=seed, device="cuda", init_device="cuda")
    g = torch.Generator(device="cuda").manual_seed(1)
    F, H, W = 33, 720, 1280
    video = (torch.rand(1, 3, F, H, W, device="cuda", generator=g) * 2 - 1).to(BF)
    noise = torch.randn(1, 16, 9, H // 8, W // 8, device="cuda", generator=g)
    text = (torch.randn(226, 4096, device="cuda", generator=g) * 0.15).to(BF)
    sa, s1 = pipe.scheduler._coeffs(torch.tensor([399]), BF)
    rope = rope_for(pipe, 10, H // 8, W // 8)
    tp = pipe.transformer.timestep_projection(399)
    times = {}
    for name, fn in (("python facade", lambda: process_video(pipe, video, empty_prompt_embedding=text, posterior_noise=noise)[0]),
                     ("dove_sr_clip", lambda: ctx.sr_clip(video[0], noise[0], text, 399, sa, s1, rope=rope, timestep_proj=tp))):
        torch.cuda.empty_cache()        # the facade's activations live in torch's caching allocator (one pool per VAE stream), the C graph's in
        out = fn()                      # its own hipMalloc'ed arena: neither can use what the other holds cached
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            out = fn()
        torch.cuda.synchronize()
        times[name] = (time.perf_counter() - t0) / 2
        assert bool(torch.isfinite(out.float()).all())
    print(f"[graph] 33x720x1280: python facade {times['python facade'] * 1e3:.1f} ms, dove_sr_clip {times['dove_sr_clip'] * 1e3:.1f} ms; "
          f"arena high water {ctx.workspace_high_water() / 2**30:.1f} GiB of {ctx.workspace_bytes(F, H, W) / 2**30:.1f} GiB requested")
    torch.cuda.empty_cache()
    a = process_video(pipe, video, empty_prompt_embedding=text, posterior_noise=noise)[0]
    b = ctx.sr_clip(video[0], noise[0], text, 399, sa, s1, rope=rope, timestep_proj=tp)
    assert torch.equal(a, b), "full-size clip: the C graph and the Python facade disagree"


@pytest.mark.parametrize("R", [2, 3, 4])
def test_vae_sharded_halo_exchange_c_level(both, R):
    """dove_comm_init_custom: R contexts on ONE GPU play the ranks of one clip, the transport is an in-process mailbox (send = copy
    into a queued buffer, recv = copy out of it).  Halos only flow rank -> rank + 1, so running the ranks one after the other
    satisfies every receive.  Gathered frames must equal the single-context result bit for bit (encode: 33 frames = 4 batches,
    decode: 9 latent frames = 4 batches; with R = 3 one rank owns two batches)."""
    import ctypes as C
    pipe, ctx0, (v, t, s, wv, wt) = both
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    box = {}                                                   # (src, dst) -> list of uint8 tensors in send order
    log = []
    keep = []                                                  # consumed messages stay allocated until the device has synchronised (foreign streams)
    seen = {}                                                  # rank -> {"send": streams, "recv": streams} the callbacks were handed
    compute = torch.cuda.current_stream().cuda_stream

    def make(rank):
        def send(peer, p, n, st):
            buf = torch.empty(n, dtype=torch.uint8, device="cuda")
            assert hip.hipMemcpyAsync(buf.data_ptr(), p, n, 3, st) == 0
            box.setdefault((rank, peer), []).append(buf)
            log.append((rank, peer, n))
            seen.setdefault(rank, {}).setdefault("send", set()).add(st or 0)
            return 0

        def recv(peer, p, n, st):
            q = box.get((peer, rank))
            if not q:
                return -1                                       # nothing was sent: the layer orders of the two ranks diverged
            buf = q.pop(0)
            keep.append(buf)
            if buf.numel() != n:
                return -2
            assert hip.hipMemcpyAsync(p, buf.data_ptr(), n, 3, st) == 0   # (the sender finished before this rank started: device-wide sync below)
            seen.setdefault(rank, {}).setdefault("recv", set()).add(st or 0)
            return 0
        return send, recv

    g = torch.Generator().manual_seed(100 + R)
    F, H, W = 33, 48, 64
    video = (torch.rand(3, F, H, W, generator=g) * 2 - 1).to(BF).cuda()
    z = torch.randn(16, 9, H // 8, W // 8, generator=g).to(BF).cuda()
    m_ref = ctx0.vae_encode(video)
    d_ref = ctx0.vae_decode(z, prescale=1 / 0.7, range01=True)
    torch.cuda.synchronize()
    ranks = []
    for r in range(R):
        c = GraphContext(v, t, wv, wt, "cuda")
        c.comm_init_custom(r, R, *make(r))
        ranks.append(c)
    # Two passes: the first RECORDS each rank's halo list (every receive posted where its conv consumes it), the second PRE-POSTS the whole list
    # on the context's receive stream before the rank's first kernel.  Either way no transfer may be handed the caller's stream (VERDICT r05
    # weak #4: ncclSend / ncclRecv on the compute stream block the convs behind the peer), and the bits must not change.
    for pass_ in range(2):
        m = torch.full_like(m_ref, float("nan"))
        d = torch.full_like(d_ref, float("nan"))
        covered_m, covered_d = 0, 0
        n0 = len(log)
        for r, c in enumerate(ranks):                               # rank order = dependency order
            first, count = c.shard_frames(0, F)
            out = c.vae_encode(video, out=torch.full_like(m_ref, float("nan")))
            torch.cuda.synchronize()
            assert bool(torch.isnan(out[:, :first].float()).all()) and bool(torch.isnan(out[:, first + count:].float()).all())   # other ranks' frames untouched
            m[:, first:first + count] = out[:, first:first + count]
            covered_m += count
            st = c.halo_stats()
            assert (st["preposted"], st["blocking"]) == ((0, 0) if r == 0 else ((0, st["blocking"]) if pass_ == 0 else (st["preposted"], 0))), (pass_, r, st)
            assert (r == 0) == (st["preposted"] + st["blocking"] == 0) and (r == R - 1) == (st["sent"] == 0), (pass_, r, st)
        assert covered_m == m_ref.shape[1] and torch.equal(m, m_ref), f"sharded encode differs (pass {pass_})"
        n_enc_msgs = len(log) - n0
        for r, c in enumerate(ranks):
            first, count = c.shard_frames(1, 9)
            out = c.vae_decode(z, prescale=1 / 0.7, range01=True, out=torch.full_like(d_ref, float("nan")))
            torch.cuda.synchronize()
            d[:, first:first + count] = out[:, first:first + count]
            covered_d += count
            st = c.halo_stats()
            assert (st["blocking"] == 0) == (r == 0 or pass_ == 1) and (st["preposted"] > 0) == (r > 0 and pass_ == 1), (pass_, r, st)
        assert covered_d == d_ref.shape[1] and torch.equal(d, d_ref), f"sharded decode differs (pass {pass_})"
        assert all(not q for q in box.values()), "unconsumed halo messages"
        assert n_enc_msgs > 0 and len(log) - n0 > n_enc_msgs and all(dst == src + 1 for src, dst, _ in log)
        keep.clear()
    for r in range(R):
        sends, recvs = seen.get(r, {}).get("send", set()), seen.get(r, {}).get("recv", set())
        assert compute not in sends and compute not in recvs, f"rank {r}: a halo transfer was handed the caller's stream"
        assert len(sends) <= 1 and len(recvs) <= 1 and not (sends & recvs), (r, sends, recvs)    # one send stream, one receive stream, distinct
    for c in ranks:
        c.comm_destroy()


class _Mailbox:
    """In-process transport for R contexts that run as R THREADS on one GPU: send = copy into a queued device buffer (buffered: never
    blocks), recv = wait for the peer's message, copy out.  The library hands the callbacks the CALLER's stream for the symmetric exchanges
    and the context's own send / receive streams for the VAE halos (ABI 14), so a message carries an event: recorded on the sender's stream
    behind the copy in, waited for on the receiver's stream before the copy out.  Consumed buffers stay referenced (``keep``) until the test
    has synchronised: torch's allocator knows nothing of copies queued on foreign streams."""

    def __init__(self):
        import ctypes as C
        import threading
        self.cv = threading.Condition()
        self.q = {}
        self.log = []
        self.keep = []
        self.streams = {}                                       # rank -> set of stream handles its callbacks were handed
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        self.hip.hipEventCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
        self.hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
        self.hip.hipStreamWaitEvent.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
        self._C = C

    def fns(self, rank):
        C = self._C

        def send(peer, p, n, st):
            buf = torch.empty(n, dtype=torch.uint8, device="cuda")
            if self.hip.hipMemcpyAsync(buf.data_ptr(), p, n, 3, st) != 0:
                return -3
            ev = C.c_void_p()
            if self.hip.hipEventCreateWithFlags(C.byref(ev), 2) != 0 or self.hip.hipEventRecord(ev, st) != 0:
                return -4
            with self.cv:
                self.q.setdefault((rank, peer), []).append((buf, ev))
                self.log.append((rank, peer, n))
                self.streams.setdefault(rank, {}).setdefault("send", set()).add(st or 0)
                self.cv.notify_all()
            return 0

        def recv(peer, p, n, st):
            with self.cv:
                if not self.cv.wait_for(lambda: self.q.get((peer, rank)), timeout=120):
                    return -1                                   # the peer never sent: the two ranks' exchange orders diverged (or it died)
                buf, ev = self.q[(peer, rank)].pop(0)
                self.keep.append(buf)
                self.streams.setdefault(rank, {}).setdefault("recv", set()).add(st or 0)
            if buf.numel() != n:
                return -2
            if self.hip.hipStreamWaitEvent(st, ev, 0) != 0:
                return -4
            return 0 if self.hip.hipMemcpyAsync(p, buf.data_ptr(), n, 3, st) == 0 else -3
        return send, recv


def _run_ranks(ctxs, fn):
    """fn(rank, ctx) on one thread per rank; returns the results in rank order, re-raising the first failure."""
    import threading
    res, err = [None] * len(ctxs), []

    def work(r):
        try:
            torch.cuda.set_device(0)
            res[r] = fn(r, ctxs[r])
        except BaseException as e:                              # noqa: BLE001
            err.append((r, e))
    ths = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(len(ctxs))]
    for t in ths:
        t.start()
    for t in ths:
        t.join(300)
    assert not any(t.is_alive() for t in ths), "a rank thread hung"
    if err:
        raise AssertionError(f"rank {err[0][0]} failed: {err[0][1]}") from err[0][1]
    return res


@pytest.mark.parametrize("R", [2, 4, 8])
def test_one_clip_sharded_c_level(both, R):
    """BASELINE configs[2] from the four graph calls alone (ABI 12): R contexts = R ranks (threads on one GPU, mailbox transport) share ONE
    clip.  R = 8 on 33 frames = PAIRED PIECES (5,4,4,4,4,4,4,4 decoded frames; GroupNorm sums swapped with the partner, Upsample3D's piece
    role); dove_dit_forward shards rows / heads with its all-to-alls (score bound in the K blocks); dove_sr_clip chains encode -> gather of
    the moments -> sharded DiT -> decode.  Every stage and the whole clip must equal the single-context result BIT FOR BIT."""
    pipe, ctx0, (v, t, s, wv, wt) = both
    g = torch.Generator().manual_seed(200 + R)
    F, H, W = 33, 128, 192                                      # every VAE level >= 16 rows: the LDS-halo conv with fused statistics runs at each
    video = (torch.rand(3, F, H, W, generator=g) * 2 - 1).to(BF).cuda()
    T = 1 + (F - 1) // 4
    noise = torch.randn(16, T, H // 8, W // 8, generator=g).cuda()
    text = (torch.randn(226, 4096, generator=g) * 0.15).to(BF).cuda()
    z = torch.randn(16, T, H // 8, W // 8, generator=g).to(BF).cuda()
    hidden = torch.randn(T + T % 2, 16, H // 8, W // 8, generator=g).to(BF).cuda()
    rope = rope_for(pipe, T + T % 2, H // 8, W // 8)
    tproj = pipe.transformer.timestep_projection(399)
    sa, s1 = pipe.scheduler._coeffs(torch.tensor([399]), BF)
    m_ref = ctx0.vae_encode(video)
    d_ref = ctx0.vae_decode(z, prescale=1 / 0.7, range01=True)
    v_ref = ctx0.dit_forward(hidden, text, 399, rope=rope, timestep_proj=tproj)
    clip_ref = ctx0.sr_clip(video, noise, text, 399, sa, s1, rope=rope, timestep_proj=tproj)
    torch.cuda.synchronize()
    box = _Mailbox()
    ctxs = []
    for r in range(R):
        c = GraphContext(v, t, wv, wt, "cuda")
        c.comm_init_custom(r, R, *box.fns(r))
        ctxs.append(c)
    try:
        # ---- VAE stages: every rank writes only its frames ----
        outs = _run_ranks(ctxs, lambda r, c: (c.shard_frames(0, F), c.vae_encode(video, out=torch.full_like(m_ref, float("nan")))))
        torch.cuda.synchronize()
        m = torch.full_like(m_ref, float("nan"))
        for (first, count), o in outs:
            assert bool(torch.isnan(o[:, :first].float()).all()) and bool(torch.isnan(o[:, first + count:].float()).all())
            m[:, first:first + count] = o[:, first:first + count]
        assert sum(cnt for (_, cnt), _ in outs) == m_ref.shape[1] and torch.equal(m, m_ref), "sharded encode differs"
        outs = _run_ranks(ctxs, lambda r, c: (c.shard_frames(1, T), c.vae_decode(z, prescale=1 / 0.7, range01=True, out=torch.full_like(d_ref, float("nan")))))
        torch.cuda.synchronize()
        d = torch.full_like(d_ref, float("nan"))
        for (first, count), o in outs:
            d[:, first:first + count] = o[:, first:first + count]
        counts = [cnt for (_, cnt), _ in outs]
        if R == 8:
            assert counts == [5, 4, 4, 4, 4, 4, 4, 4], counts      # paired pieces: BASELINE's "frame-chunk = 4"
        assert sum(counts) == d_ref.shape[1] and torch.equal(d, d_ref), "sharded decode differs"
        # ---- the same two stages again: every rank now PRE-POSTS its halo receives (recorded above) on its own receive stream while the
        # ranks really run concurrently - same bits, nothing left to a blocking receive, and no halo on the callers' stream ----
        box.keep.clear()
        outs = _run_ranks(ctxs, lambda r, c: (c.shard_frames(0, F), c.vae_encode(video, out=torch.full_like(m_ref, float("nan"))), c.halo_stats()))
        torch.cuda.synchronize()
        m = torch.full_like(m_ref, float("nan"))
        # (a rank whose first work item is the UPPER piece of a split frame-batch - the odd ranks at R = 8 - gets its partner's GroupNorm sums and its
        # halos from the same peer in one message order: a transport without channels cannot post those receives ahead, include/dove_hip.h)
        ahead = lambda r: r > 0 and not (R == 8 and r % 2 == 1)   # noqa: E731
        for r, ((first, count), o, st) in enumerate(outs):
            m[:, first:first + count] = o[:, first:first + count]
            assert (st["blocking"] == 0) == (r == 0 or ahead(r)) and (st["preposted"] > 0) == ahead(r), (r, st)
        assert torch.equal(m, m_ref), "sharded encode with pre-posted halos differs"
        outs = _run_ranks(ctxs, lambda r, c: (c.shard_frames(1, T), c.vae_decode(z, prescale=1 / 0.7, range01=True, out=torch.full_like(d_ref, float("nan"))), c.halo_stats()))
        torch.cuda.synchronize()
        d = torch.full_like(d_ref, float("nan"))
        for r, ((first, count), o, st) in enumerate(outs):
            d[:, first:first + count] = o[:, first:first + count]
            assert (st["blocking"] == 0) == (r == 0 or ahead(r)) and (st["preposted"] > 0) == ahead(r), (r, st)
        assert torch.equal(d, d_ref), "sharded decode with pre-posted halos differs"
        box.keep.clear()
        # ---- DiT: rows / heads sharded, the velocity complete on every rank ----
        vs = _run_ranks(ctxs, lambda r, c: c.dit_forward(hidden, text, 399, rope=rope, timestep_proj=tproj))
        torch.cuda.synchronize()
        for r, vv in enumerate(vs):
            assert torch.equal(vv, v_ref), f"sharded DiT differs on rank {r}: {float((vv.float() - v_ref.float()).abs().max())}"
        # ---- the whole clip ----
        clips = _run_ranks(ctxs, lambda r, c: (c.shard_frames(1, T), c.sr_clip(video, noise, text, 399, sa, s1, rope=rope, timestep_proj=tproj)))
        torch.cuda.synchronize()
        clip = torch.full_like(clip_ref, float("nan"))
        for (first, count), o in clips:
            clip[:, first:first + count] = o[:, first:first + count]
        assert torch.equal(clip, clip_ref), f"sharded sr_clip differs: {float((clip.float() - clip_ref.float()).abs().max())}"
        assert all(not q for q in box.q.values()), "unconsumed messages"
        pair = [(a, b) for a, b, n in box.log if n == 65 * 8]
        assert (len(pair) > 0) == (R > 4), "GroupNorm pair sums travel exactly when frame-batches are split"
    finally:
        for c in ctxs:
            c.comm_destroy()


def test_comm_init_rccl_single_rank(both):
    """The RCCL transport binding (librccl opened at run time, ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy) on the one GPU a
    test box has: a 1-rank communicator must come up, leave the stages unchanged and go away again."""
    pipe, ctx0, (v, t, s, _wv, _wt) = both
    g = torch.Generator().manual_seed(5)
    video = (torch.rand(3, 17, 32, 48, generator=g) * 2 - 1).to(BF).cuda()
    ref = ctx0.vae_encode(video)
    uid = GraphContext.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    ctx0.comm_init_rccl(uid, 0, 1)
    assert ctx0.shard_frames(0, 17) == (0, 5)
    got = ctx0.vae_encode(video)
    torch.cuda.synchronize()
    ctx0.comm_destroy()
    assert torch.equal(got, ref)
