"""ctypes binding of libdove_hip.so (include/dove_hip.h).  No CPU fallback: every op raises if the
library is missing or the call fails -- the product path never routes through torch math or the oracle."""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdove_hip.so")
_lib = None

F32, BF16 = 0, 1


class ConvDesc(C.Structure):
    """dove_conv_desc (include/dove_hip.h).  ``struct_size`` is filled in on construction: the library refuses a descriptor whose
    size is not the one it was built with (a stale binding would otherwise have its missing tail fields read from stray memory)."""
    _fields_ = [
        ("struct_size", C.c_uint), ("reserved", C.c_uint),
        ("x", C.c_void_p), ("cache", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p),
        ("resid", C.c_void_p), ("gate", C.c_void_p), ("out", C.c_void_p),
        ("t_in", C.c_int), ("h_in", C.c_int), ("w_in", C.c_int), ("cin", C.c_int),
        ("t_out", C.c_int), ("h_out", C.c_int), ("w_out", C.c_int), ("cout_pad", C.c_int), ("cout_store", C.c_int),
        ("kt", C.c_int), ("kh", C.c_int), ("kw", C.c_int), ("stride", C.c_int), ("pad_h", C.c_int), ("pad_w", C.c_int),
        ("up", C.c_int), ("tmode", C.c_int), ("act", C.c_int),
        ("ldo", C.c_longlong), ("ldr", C.c_longlong), ("gate_split", C.c_longlong),
        ("gn_partial", C.c_void_p), ("out_f32", C.c_int),
        ("nb", C.c_int), ("cache_stride", C.c_longlong), ("w_first", C.c_void_p), ("w_sub", C.c_void_p),
        ("w_pair", C.c_void_p), ("tdup", C.c_int), ("reserved2", C.c_int),
    ]

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.struct_size = C.sizeof(ConvDesc)


class ModelConfig(C.Structure):
    """dove_model_config (include/dove_hip.h): the fields of vae/config.json and transformer/config.json the graph needs."""
    _fields_ = [(n, C.c_int) for n in ("vae_in_channels", "vae_out_channels", "vae_latent_channels", "vae_num_blocks")] + \
               [("vae_block_out_channels", C.c_int * 8)] + \
               [(n, C.c_int) for n in ("vae_layers_per_block", "vae_temporal_compression", "vae_enc_batch", "vae_dec_batch")] + \
               [("vae_norm_eps", C.c_float), ("vae_scaling_factor", C.c_float)] + \
               [(n, C.c_int) for n in ("dit_heads", "dit_head_dim", "dit_num_layers", "dit_in_channels", "dit_out_channels", "dit_patch",
                                       "dit_patch_t", "dit_text_dim", "dit_time_embed_dim", "dit_max_text", "dit_flip_sin_to_cos")] + \
               [("dit_norm_eps", C.c_float), ("dit_freq_shift", C.c_float)]


class DitAux(C.Structure):
    _fields_ = [("rope_cos", C.c_void_p), ("rope_sin", C.c_void_p), ("timestep_proj", C.c_void_p)]


class PreNoise(C.Structure):
    """dove_pre_noise: `--noise_step` of the graph-level dove_sr_clip."""
    _fields_ = [("eps", C.c_void_p), ("eps_dtype", C.c_int), ("sqrt_alpha", C.c_float), ("sqrt_one_minus_alpha", C.c_float)]


# dove_set_option keys (include/dove_hip.h)
OPT_VAE_TILING, OPT_VAE_SAMPLE_HEIGHT, OPT_VAE_SAMPLE_WIDTH, OPT_DIT_LINEAR_MXFP8, OPT_DIT_ATTN_MXFP8, OPT_WEIGHT_SUMS, OPT_VAE_STREAMS = 1, 2, 3, 4, 5, 6, 7
STAT_HALO_PREPOSTED, STAT_HALO_BLOCKING, STAT_HALO_SENT, STAT_HALO_COMMUNICATORS = 100, 101, 102, 103


_VP, _I, _LL, _F = C.c_void_p, C.c_int, C.c_longlong, C.c_float
XFER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p)   # dove_xfer_fn
GROUP_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)                                  # dove_group_fn

# name -> argtypes; the symbol list doubles as the export check in tests/test_abi.py
SIGNATURES = {
    "dove_conv_igemm_bf16": [C.POINTER(ConvDesc), _VP],
    "dove_groupnorm_stats_bf16": [_VP, _LL, _LL, _I, _F, _VP, _I, _VP, _VP],
    "dove_groupnorm_finalize_partials": [_VP, _LL, C.c_double, _F, _VP, _VP, _VP],
    "dove_groupnorm_stats_nb_bf16": [_VP, _I, _LL, _LL, _I, _F, _VP, _I, _VP, _VP],
    "dove_groupnorm_finalize_partials_nb": [_VP, _LL, _I, C.c_double, _F, _VP, C.c_size_t, _VP, _VP],
    "dove_groupnorm_apply_nb_bf16": [_VP, _VP, _I, _I, _I, _I, _I, _VP, _VP, _VP, _I, _VP, _I, _I, _I, _I, C.POINTER(C.c_int), _VP],
    "dove_avgpool_time_nb_bf16": [_VP, _I, _I, _LL, _VP, _VP],
    "dove_groupnorm_sums_bf16": [_VP, _LL, _LL, _I, _VP, _I, _VP, _VP],
    "dove_groupnorm_finalize_sums": [_VP, C.c_double, _F, _VP, _VP],
    "dove_groupnorm_sums_from_partials": [_VP, _LL, _VP, _VP, _VP],
    "dove_groupnorm_apply_bf16": [_VP, _VP, _I, _I, _I, _I, _VP, _VP, _VP, _I, _VP, _I, _I, _I, C.POINTER(C.c_int), _VP],
    "dove_layernorm_modulate_bf16": [_VP, _VP, _LL, _I, _F, _VP, _VP, _VP, _LL, _VP],
    "dove_qkv_post_bf16": [_VP, _LL, _LL, _I, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP, _F, _F, _VP, _VP, _VP, _I, _VP, _VP],
    "dove_vt_quad_swap_bf16": [_VP, _LL, _LL, _VP],
    "dove_ulysses_place_bf16": [_VP, _VP, _VP, C.POINTER(C.c_longlong), _I, _I, _LL, _LL, _VP, _VP, _VP, _VP, _VP],
    "dove_cl_im2col3x3_from_ncthw": [_VP, _I, _I, _I, _I, _I, _I, _F, _F, _VP, _VP],
    "dove_conv_out_gather": [_VP, _LL, _I, _I, _I, _I, _VP, _F, _F, _F, _F, _VP, _I, _VP],
    "dove_attention_fwd_bf16": [_VP, _VP, _VP, _VP, _LL, _LL, _I, _I, _LL, _VP, _VP],
    "dove_qkv_post_mxfp8": [_VP, _LL, _LL, _I, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP, _F, _F, _VP, _VP, _VP, _VP, _VP],
    "dove_attention_fwd_mxfp8": [_VP, _VP, _VP, _VP, _VP, _LL, _LL, _I, _I, _LL, _VP],
    "dove_cl_from_ncthw": [_VP, _I, _I, _LL, _I, _F, _F, _VP, _VP],
    "dove_ncthw_from_cl": [_VP, _LL, _I, _LL, _F, _F, _F, _F, _VP, _I, _VP],
    "dove_avgpool_time_bf16": [_VP, _I, _LL, _VP, _VP],
    "dove_posterior_sample": [_VP, _LL, _I, _LL, _VP, _I, _VP, _I, _VP],
    "dove_axpby": [_VP, _VP, _VP, _I, _LL, _F, _F, _VP],
    "dove_patchify": [_VP, _I, _I, _I, _I, _I, _I, _I, _VP, _LL, _VP],
    "dove_unpatchify": [_VP, _LL, _I, _I, _I, _I, _I, _I, _VP, _I, _VP],
    "dove_gemv_bf16": [_VP, _VP, _VP, _I, _I, _I, _VP, _VP],
    "dove_blend_edge_bf16": [_VP, _VP, _I, _I, _I, _I, _I, _I, _I, _I, _VP],
    "dove_preprocess_u8": [_VP, _I, _I, _I, _I, _I, _I, _I, _VP, _I, _VP],
    "dove_postprocess_u8": [_VP, _I, _I, _I, _I, _I, _I, _I, _VP, _VP],
    "dove_rmsnorm_bf16": [_VP, _VP, _LL, _I, _F, _VP, _VP],
    "dove_gated_gelu_bf16": [_VP, _VP, _LL, _I, _VP],
    "dove_attention_bias_bf16": [_VP, _VP, _VP, _LL, _VP, _VP, _LL, _I, _I, _I, _VP],
    "dove_mx_quant_bf16": [_VP, _LL, _I, _VP, _VP, _VP],
    "dove_create": [_I, C.POINTER(ModelConfig), C.POINTER(_VP)],
    "dove_set_weight": [_VP, C.c_char_p, _VP, C.POINTER(C.c_longlong), _I, _I],
    "dove_finalize_weights": [_VP],
    "dove_set_workspace": [_VP, _VP, C.c_size_t],
    "dove_vae_encode": [_VP, _VP, _I, _I, _I, _I, _VP, _I, _VP],
    "dove_dit_forward": [_VP, _VP, _I, _I, _I, _I, _VP, _I, _I, C.POINTER(DitAux), _VP, _I, _VP],
    "dove_vae_decode": [_VP, _VP, _I, _I, _I, _I, _F, _I, _VP, _I, _VP],
    "dove_sr_clip": [_VP, _VP, _I, _I, _I, _I, _VP, _I, _VP, _I, _I, _F, _F, C.POINTER(DitAux), C.POINTER(PreNoise), _VP, _I, _VP],
    "dove_set_option": [_VP, _I, _LL],
    "dove_comm_unique_id": [_VP],
    "dove_comm_init": [_VP, _VP, _I, _I],
    "dove_comm_init_custom": [_VP, _I, _I, XFER_FN, XFER_FN, _VP],
    "dove_comm_set_group": [_VP, GROUP_FN, GROUP_FN],
    "dove_shard_frames": [_VP, _I, _I, C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "dove_conv_out_gather_cl": [_VP, _LL, _I, _I, _I, _I, _VP, _VP, _I, _VP],
    "dove_tile_gather_bf16": [_VP, _I, _I, _I, _I, _I, _I, _I, _I, C.POINTER(C.c_int), C.POINTER(C.c_int), _I, _VP, _VP],
    "dove_linear_mxfp8": [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _LL, _I, _I, _LL, _LL, _LL, _I, _VP],
}
PLAIN = {"dove_last_error": (C.c_char_p, []), "dove_abi_version": (C.c_int, []), "dove_comm_destroy": (None, [C.c_void_p]),
         "dove_conv_gn_partial_rows": (C.c_longlong, [C.POINTER(ConvDesc)]),
         "dove_conv_kernel_name": (C.c_char_p, [C.POINTER(ConvDesc)]),
         "dove_conv_partial_launches": (C.c_int, [C.POINTER(ConvDesc)]),
         "dove_attention_head_paths": (C.c_int, [C.POINTER(C.c_float), _I, C.POINTER(C.c_int)]),
         "dove_attention_path_name": (C.c_char_p, [_I]),
         "dove_device_info": (C.c_int, [_I, C.c_char_p, _I, C.POINTER(C.c_int), C.POINTER(C.c_longlong)]),
         "dove_destroy": (None, [_VP]),
         "dove_vae_decode_num_frames": (C.c_int, [_VP, _I]),
         "dove_get_option": (C.c_longlong, [_VP, _I]),
         "dove_comm_useful_ranks": (C.c_int, [_VP, _I, _I]),
         "dove_workspace_bytes": (C.c_size_t, [_VP, _I, _I, _I]),
         "dove_workspace_high_water": (C.c_size_t, [_VP])}


def kernel_source_sha256() -> str:
    """sha256 over the kernel sources (csrc/*.hip, csrc/*.h, include/dove_hip.h; names and bytes, sorted): what a measurement of the kernels
    is a measurement OF.  tools/pmc_bench_traffic.py stores it in profiles/pmc_traffic.json and bench.py replays that summary only when it
    equals the tree's - a PMC pass taken on other kernel code (round 4: the MFMA-shape move) can no longer be replayed as this build's."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.h")))
    files.append(os.path.join(os.path.dirname(_HERE), "include", "dove_hip.h"))
    for fn in files:
        h.update(os.path.basename(fn).encode() + b"\0")
        with open(fn, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()


def use_timing_build():
    """tools/ only: bind the separate -DDOVE_TIMING_BUILD library (ablation switches, s_memtime phase logs; built by
    ``dove_amd/csrc/build.sh timing``) instead of the product library.  Must be called before the first ``load()``."""
    global LIB_PATH
    assert _lib is None, "use_timing_build() must precede the first load()"
    path = os.path.join(_HERE, "libdove_hip_timing.so")
    if not os.path.exists(path):
        import subprocess
        subprocess.check_call(["bash", os.path.join(_HERE, "csrc", "build.sh"), "timing"])
    LIB_PATH = path


def load():
    """Load libdove_hip.so (built by ``__graft_entry__.build()`` / dove_amd/csrc/build.sh)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(dove_amd has no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    for name, argt in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argt
        fn.restype = C.c_int
    for name, (res, argt) in PLAIN.items():
        fn = getattr(lib, name)
        fn.argtypes = argt
        fn.restype = res
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {load().dove_last_error().decode()}")


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def dt_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"dove_amd supports float32 / bfloat16 boundary tensors, got {t.dtype}")


def require_cuda(*tensors):
    cur = None
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("dove_amd ops need tensors on the HIP device (`cuda`); there is no CPU path")
        if t is not None:
            # kernels launch on the CURRENT device's stream (stream_ptr) and the library keeps per-device state keyed by
            # hipGetDevice(): a tensor of another GPU would be dereferenced there
            if cur is None:
                cur = torch.cuda.current_device()
            if t.device.index != cur:
                raise RuntimeError(f"dove_amd op called with a tensor on cuda:{t.device.index} while cuda:{cur} is current; "
                                   "wrap the call in torch.cuda.device(tensor.device)")
        if t is not None and not t.is_contiguous():
            raise RuntimeError("dove_amd ops need contiguous tensors")
