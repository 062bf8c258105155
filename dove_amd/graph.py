"""Python face of the graph-level C entry points (include/dove_hip.h: dove_create ... dove_sr_clip): a context that owns the
packed weights and one workspace arena inside libdove_hip.so and runs whole stages - what a non-Python host would bind
(INTEGRATION.md section 3).  The default facade (dove_amd.pipeline) drives the operator-level entry points from Python; this
class makes the same arithmetic reachable with four calls and is tested bit-for-bit against it (tests/test_graph_gpu.py)."""
from __future__ import annotations

import ctypes as C

import torch

from . import lib as L


def model_config(vae_cfg: dict, dit_cfg: dict) -> "L.ModelConfig":
    m = L.ModelConfig()
    boc = list(vae_cfg["block_out_channels"])
    m.vae_in_channels, m.vae_out_channels, m.vae_latent_channels = vae_cfg["in_channels"], vae_cfg["out_channels"], vae_cfg["latent_channels"]
    m.vae_num_blocks = len(boc)
    for i, v in enumerate(boc):
        m.vae_block_out_channels[i] = v
    m.vae_layers_per_block = vae_cfg.get("layers_per_block", 3)
    m.vae_temporal_compression = vae_cfg.get("temporal_compression_ratio", 4)
    m.vae_enc_batch = vae_cfg.get("num_sample_frames_batch_size", 8)
    m.vae_dec_batch = vae_cfg.get("num_latent_frames_batch_size", 2)
    m.vae_norm_eps, m.vae_scaling_factor = vae_cfg.get("norm_eps", 1e-6), vae_cfg["scaling_factor"]
    m.dit_heads, m.dit_head_dim, m.dit_num_layers = dit_cfg["num_attention_heads"], dit_cfg["attention_head_dim"], dit_cfg["num_layers"]
    m.dit_in_channels, m.dit_out_channels = dit_cfg["in_channels"], dit_cfg["out_channels"]
    m.dit_patch, m.dit_patch_t = dit_cfg["patch_size"], dit_cfg["patch_size_t"]
    m.dit_text_dim, m.dit_time_embed_dim = dit_cfg["text_embed_dim"], dit_cfg["time_embed_dim"]
    m.dit_max_text = dit_cfg.get("max_text_seq_length", 226)
    m.dit_flip_sin_to_cos = int(dit_cfg.get("flip_sin_to_cos", True))
    m.dit_norm_eps, m.dit_freq_shift = dit_cfg.get("norm_eps", 1e-5), float(dit_cfg.get("freq_shift", 0))
    return m


class GraphContext:
    def __init__(self, vae_cfg: dict, dit_cfg: dict, vae_sd, dit_sd, device="cuda", dit_linear_precision="bf16", dit_attention_precision="bf16"):
        """``dit_linear_precision`` / ``dit_attention_precision`` = "mxfp8": BASELINE configs[4] (DOVE_OPT_DIT_LINEAR_MXFP8 is chosen before
        the weights are finalized: they are quantised there)."""
        self.device = torch.device(device if ":" in str(device) else f"{device}:{torch.cuda.current_device()}")
        self._h = C.c_void_p()
        cfg = model_config(vae_cfg, dit_cfg)
        L.check(L.load().dove_create(self.device.index, C.byref(cfg), C.byref(self._h)), "dove_create")
        self.vae_cfg, self.dit_cfg = vae_cfg, dit_cfg
        self.set_option(L.OPT_VAE_SAMPLE_HEIGHT, vae_cfg.get("sample_height", 480))
        self.set_option(L.OPT_VAE_SAMPLE_WIDTH, vae_cfg.get("sample_width", 720))
        self.set_option(L.OPT_DIT_LINEAR_MXFP8, int(dit_linear_precision == "mxfp8"))
        self.set_option(L.OPT_DIT_ATTN_MXFP8, int(dit_attention_precision == "mxfp8"))
        keep = []
        for sd in (vae_sd, dit_sd):
            for name in sd.keys():
                t = sd[name].to(self.device).contiguous()
                if t.dtype not in (torch.float32, torch.bfloat16):
                    t = t.float()
                keep.append(t)                                  # borrowed by the library until finalize returns
                shape = (C.c_longlong * t.dim())(*t.shape)
                L.check(L.load().dove_set_weight(self._h, name.encode(), L.ptr(t), shape, t.dim(), L.dt_code(t)), f"dove_set_weight({name})")
        L.check(L.load().dove_finalize_weights(self._h), "dove_finalize_weights")
        del keep

    def __del__(self):
        try:
            if self._h:
                L.load().dove_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def set_option(self, option: int, value: int):
        L.check(L.load().dove_set_option(self._h, int(option), int(value)), f"dove_set_option({option})")

    def get_option(self, option: int) -> int:
        return int(L.load().dove_get_option(self._h, int(option)))

    def enable_tiling(self, on: bool = True):
        """pipe.vae.enable_tiling() (`--is_vae_st`, ref :643-645) for vae_encode / vae_decode / sr_clip."""
        self.set_option(L.OPT_VAE_TILING, int(on))

    def useful_ranks(self, stage: int, n: int) -> int:
        return int(L.load().dove_comm_useful_ranks(self._h, stage, n))

    # ---- one clip on several ranks (halo-exact VAE; include/dove_hip.h "one clip on several GPUs") ----
    def comm_init_rccl(self, unique_id: bytes, rank: int, nranks: int):
        L.check(L.load().dove_comm_init(self._h, unique_id, rank, nranks), "dove_comm_init")
        self._rank, self._nranks = rank, nranks

    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        L.check(L.load().dove_comm_unique_id(buf), "dove_comm_unique_id")
        return buf.raw

    def comm_init_custom(self, rank: int, nranks: int, send, recv):
        """``send(peer, dev_ptr, nbytes, stream)`` / ``recv(...)`` -> 0 on success: any transport (tests: an in-process mailbox)."""
        self._xfer = (L.XFER_FN(lambda user, peer, p, n, st: send(peer, p, n, st)), L.XFER_FN(lambda user, peer, p, n, st: recv(peer, p, n, st)))
        L.check(L.load().dove_comm_init_custom(self._h, rank, nranks, self._xfer[0], self._xfer[1], None), "dove_comm_init_custom")
        self._rank, self._nranks = rank, nranks

    def halo_stats(self) -> dict:
        """Counters of the last vae_encode / vae_decode of a multi-rank context (include/dove_hip.h DOVE_STAT_*)."""
        return dict(preposted=self.get_option(L.STAT_HALO_PREPOSTED), blocking=self.get_option(L.STAT_HALO_BLOCKING),
                    sent=self.get_option(L.STAT_HALO_SENT), communicators=self.get_option(L.STAT_HALO_COMMUNICATORS))

    def comm_destroy(self):
        L.load().dove_comm_destroy(self._h)

    def shard_frames(self, stage: int, n: int):
        """(first, count) of the output frames this rank produces: stage 0 = vae_encode (n pixel frames), 1 = vae_decode (n latent)."""
        first, count = C.c_int(), C.c_int()
        L.check(L.load().dove_shard_frames(self._h, stage, n, C.byref(first), C.byref(count)), "dove_shard_frames")
        return first.value, count.value

    def workspace_bytes(self, F, H, W) -> int:
        return int(L.load().dove_workspace_bytes(self._h, F, H, W))

    def workspace_high_water(self) -> int:
        return int(L.load().dove_workspace_high_water(self._h))

    def set_workspace(self, nbytes: int, buffer: torch.Tensor | None = None):
        """Library-owned arena of ``nbytes`` (regrown on demand), or - with ``buffer`` - caller memory the library uses as is (kept alive
        here; a stage that needs more fails with "workspace exhausted")."""
        if buffer is not None:
            L.require_cuda(buffer)
            assert buffer.is_contiguous() and buffer.numel() * buffer.element_size() >= nbytes
        self._ws_keep = buffer
        L.check(L.load().dove_set_workspace(self._h, L.ptr(buffer) if buffer is not None else None, nbytes), "dove_set_workspace")

    @staticmethod
    def _aux(rope=None, timestep_proj=None):
        if rope is None and timestep_proj is None:
            return None, ()
        a = L.DitAux()
        keep = []
        if rope is not None:
            cos, sin = (r.float().contiguous() for r in rope)
            a.rope_cos, a.rope_sin = cos.data_ptr(), sin.data_ptr()
            keep += [cos, sin]
        if timestep_proj is not None:
            tp = timestep_proj.float().contiguous()
            a.timestep_proj = tp.data_ptr()
            keep.append(tp)
        return a, keep

    def vae_encode(self, x: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        """x [3,F,H,W] -> moments [2L, T, H/8, W/8] bf16 (with a multi-rank communicator only this rank's frames are written)."""
        L.require_cuda(x)
        _, F, H, W = x.shape
        lat = self.vae_cfg["latent_channels"]
        out = out if out is not None else torch.empty(2 * lat, 1 + (F - 1) // self.vae_cfg.get("temporal_compression_ratio", 4), H // 8, W // 8, dtype=torch.bfloat16, device=x.device)
        L.check(L.load().dove_vae_encode(self._h, L.ptr(x), L.dt_code(x), F, H, W, L.ptr(out), L.BF16, L.stream_ptr()), "dove_vae_encode")
        return out

    def dit_forward(self, hidden: torch.Tensor, text: torch.Tensor, timestep: int, rope=None, timestep_proj=None) -> torch.Tensor:
        """hidden [T,C,h,w], text [L, text_dim] bf16 -> v [T,C,h,w] (hidden's dtype)."""
        L.require_cuda(hidden, text)
        assert text.dtype == torch.bfloat16
        T, _, h, w = hidden.shape
        out = torch.empty_like(hidden)
        aux, keep = self._aux(rope, timestep_proj)
        L.check(L.load().dove_dit_forward(self._h, L.ptr(hidden), L.dt_code(hidden), T, h, w, L.ptr(text), text.shape[0], int(timestep),
                                          C.byref(aux) if aux is not None else None, L.ptr(out), L.dt_code(out), L.stream_ptr()),
                "dove_dit_forward")
        return out

    def vae_decode(self, z: torch.Tensor, prescale: float = 1.0, range01: bool = False, out: torch.Tensor | None = None) -> torch.Tensor:
        """z [L,T,h,w] -> video [3, 1+4(T-1), 8h, 8w] bf16 (with a multi-rank communicator only this rank's frames are written)."""
        L.require_cuda(z)
        _, T, h, w = z.shape
        out = out if out is not None else torch.empty(self.vae_cfg["out_channels"], int(L.load().dove_vae_decode_num_frames(self._h, T)), 8 * h, 8 * w,
                          dtype=torch.bfloat16, device=z.device)
        L.check(L.load().dove_vae_decode(self._h, L.ptr(z), L.dt_code(z), T, h, w, prescale, int(range01), L.ptr(out), L.BF16, L.stream_ptr()),
                "dove_vae_decode")
        return out

    def sr_clip(self, video: torch.Tensor, noise: torch.Tensor, text: torch.Tensor, timestep: int, sqrt_alpha: float,
                sqrt_one_minus_alpha: float, rope=None, timestep_proj=None, pre_noise=None) -> torch.Tensor:
        """process_video on device buffers: video [3,F,H,W] in [-1,1], noise [L,T,h,w], text [Lt, text_dim] bf16 -> [3,F,H,W] in [0,1].
        ``pre_noise`` = (eps [T',L,h,w], sqrt_alpha_n, sqrt_one_minus_alpha_n): the `--noise_step` pre-noising (ref :449-457)."""
        L.require_cuda(video, noise, text)
        _, F, H, W = video.shape
        out = torch.empty(3, F, H, W, dtype=torch.bfloat16, device=video.device)
        aux, keep = self._aux(rope, timestep_proj)
        pre = None
        if pre_noise is not None:
            eps, sa, sb = pre_noise
            L.require_cuda(eps)
            eps = eps.contiguous()
            pre = L.PreNoise(eps.data_ptr(), L.dt_code(eps), float(sa), float(sb))
        L.check(L.load().dove_sr_clip(self._h, L.ptr(video), L.dt_code(video), F, H, W, L.ptr(noise), L.dt_code(noise), L.ptr(text), text.shape[0],
                                      int(timestep), sqrt_alpha, sqrt_one_minus_alpha, C.byref(aux) if aux is not None else None,
                                      C.byref(pre) if pre is not None else None, L.ptr(out), L.BF16, L.stream_ptr()), "dove_sr_clip")
        return out
