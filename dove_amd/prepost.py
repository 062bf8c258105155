"""GPU-side script-level pre/post-processing and dependency-free frame I/O.

Mirrors /root/reference/inference_script.py: ``preprocess_video_match`` padding (:192-235), the bilinear upscale and
[-1,1] normalisation (:670-679), ``remove_padding_and_extra_frames`` (:238-246, called with the reference's hard-coded
``pad*4`` at :731) and the uint8 conversion of the savers (:124).  Video decoding (decord/H.264) stays outside: clips
come in as uint8 arrays ([F,H,W,3] ``.npy``) or PNG folders."""
from __future__ import annotations

import os

import torch

from . import ops, tiling


def preprocess_frames(frames_u8: torch.Tensor, upscale: int = 4, dtype=torch.bfloat16, device="cuda", upscale_mode: str = "bilinear"):
    """[F,H,W,3] uint8 -> ([1,3,F',H',W'] in [-1,1] on the GPU, pad_f, pad_h, pad_w, original_shape).

    ``upscale_mode`` is the reference's ``--upscale_mode`` (ref :672: ``F.interpolate(video, size, mode=..., align_corners=False)``).
    "bilinear" (the default of every documented run) is the fused HIP kernel ``dove_preprocess_u8``; any other mode torch accepts with
    ``align_corners=False`` ("bicubic") runs the same three script steps - pad, ``F.interpolate``, ``/255*2-1`` - with torch on
    the device: script-level pre-processing, not part of the accelerated operator."""
    F, H, W, C = frames_u8.shape
    assert C == 3 and frames_u8.dtype == torch.uint8
    pad_f, pad_h, pad_w = tiling.match_padding(F, H, W)
    if upscale_mode == "bilinear":
        video = ops.preprocess_u8(frames_u8.to(device).contiguous(), pad_f, pad_h, pad_w, upscale, dtype)
    else:
        video = preprocess_frames_torch(frames_u8.to(device), pad_f, pad_h, pad_w, upscale, upscale_mode, dtype)
    return video[None], pad_f, pad_h, pad_w, (F, H, W, C)


def preprocess_frames_torch(frames_u8, pad_f, pad_h, pad_w, upscale, upscale_mode, dtype):
    """The reference's own three steps (``preprocess_video_match`` padding :220-233, ``F.interpolate`` :672, normalise :673-676):
    [F,H,W,3] uint8 -> [3,F',H',W'] in [-1,1]."""
    v = frames_u8.permute(0, 3, 1, 2).float()                               # [F,C,H,W] like decord + permute (ref :206)
    if pad_f:
        v = torch.cat([v, v[-1:].repeat(pad_f, 1, 1, 1)], dim=0)            # repeat the last frame (ref :222-224)
    if pad_h or pad_w:
        v = torch.nn.functional.pad(v, (0, pad_w, 0, pad_h))                # zeros, bottom / right (ref :232)
    v = torch.nn.functional.interpolate(v, size=(v.shape[2] * upscale, v.shape[3] * upscale), mode=upscale_mode, align_corners=False)
    return (v / 255.0 * 2.0 - 1.0).permute(1, 0, 2, 3).contiguous().to(dtype)


def postprocess_frames(video: torch.Tensor, pad_f: int, pad_h: int, pad_w: int, crop_scale: int = 4) -> torch.Tensor:
    """[1,3,F,H,W] in [0,1] -> [F',H',W',3] uint8 with the padding removed.  ``crop_scale`` is the reference's hard-coded 4
    (ref :731 multiplies the LR pads by 4 regardless of --upscale)."""
    _, _, F, H, W = video.shape
    return ops.postprocess_u8(video[0].contiguous(), F - pad_f, H - pad_h * crop_scale, W - pad_w * crop_scale)


def load_frames(path: str) -> torch.Tensor:
    """``.npy`` ([F,H,W,3] uint8) or a folder of PNG/JPG frames -> uint8 tensor [F,H,W,3]."""
    import numpy as np
    if os.path.isdir(path):
        from PIL import Image
        names = sorted(n for n in os.listdir(path) if n.lower().endswith((".png", ".jpg", ".jpeg")))
        if not names:
            raise ValueError(f"no frames in {path}")
        arr = np.stack([np.asarray(Image.open(os.path.join(path, n)).convert("RGB")) for n in names])
    elif path.lower().endswith(".npy"):
        arr = np.load(path)
    else:
        raise ValueError(f"unsupported input {path}: H.264 decoding (decord) is outside the accelerated path; "
                         "convert the clip to a PNG folder or an .npy array")
    if arr.dtype != np.uint8 or arr.ndim != 4 or arr.shape[3] != 3:
        raise ValueError(f"expected uint8 [F,H,W,3], got {arr.dtype} {arr.shape}")
    return torch.from_numpy(arr.copy())


def save_frames_as_png(frames_u8: torch.Tensor, output_dir: str):
    """Same file naming as the reference's ``save_frames_as_png`` (ref :111-128)."""
    from PIL import Image
    os.makedirs(output_dir, exist_ok=True)
    for i, fr in enumerate(frames_u8.cpu().numpy()):
        Image.fromarray(fr).save(os.path.join(output_dir, f"{i:03d}.png"))
