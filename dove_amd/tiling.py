"""Host-side chunk/tile planner and stitcher (integer logic only).

Re-statement of the reference's script-level memory tiling
(/root/reference/inference_script.py:238-361 and the stitch/coverage loop :685-729).  Each
(time-chunk x spatial-tile) is an INDEPENDENT one-step SR call; half of every interior overlap is
discarded on each side and every output voxel must be written exactly once.  Pinned against golden
vectors produced by the reference's own functions (tests/golden/tiler_golden.json).
The time-chunk axis is also the multi-GPU shard axis (dove_amd/dist.py).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple


def _axis_starts(total: int, length: int, overlap: int, allow_empty_fix: bool) -> List[int]:
    stride = length - overlap
    starts = list(range(0, total - overlap, stride))
    if (allow_empty_fix and not starts) or (starts and starts[-1] + length < total):
        starts.append(total - length)
    return starts


def make_temporal_chunks(F: int, chunk_len: int, overlap_t: int = 8) -> List[Tuple[int, int]]:
    """(start, end) frame ranges; ``chunk_len == 0`` means one chunk.  A short trailing chunk is merged into
    its predecessor (ref :274-277).  Raises ValueError when chunk_len <= overlap_t (ref :262-263)."""
    if chunk_len == 0:
        return [(0, F)]
    if chunk_len - overlap_t <= 0:
        raise ValueError("chunk_len must be greater than overlap")
    starts = _axis_starts(F, chunk_len, overlap_t, allow_empty_fix=False)
    chunks = [(s, min(s + chunk_len, F)) for s in starts]
    if len(chunks) >= 2 and chunks[-1][1] - chunks[-1][0] < chunk_len:
        tail = chunks.pop()
        chunks[-1] = (chunks[-1][0], tail[1])
    return chunks


def make_spatial_tiles(H: int, W: int, tile_size_hw: Sequence[int], overlap_hw: Sequence[int] = (32, 32)):
    """(h0, h1, w0, w1) tiles; (0, 0) tile size means one tile.  Keeps the reference's "merge last row/col"
    behaviour (ref :303-327): a start whose tile would cross the border is dropped and the previous tile
    is stretched to the border."""
    th, tw = tile_size_hw
    oh, ow = overlap_hw
    if th == 0 or tw == 0:
        return [(0, H, 0, W)]
    sh, sw = th - oh, tw - ow
    if sh <= 0 or sw <= 0:
        raise ValueError("Tile size must be greater than overlap")

    def starts(total, length, overlap):
        st = _axis_starts(total, length, overlap, allow_empty_fix=True)
        if len(st) >= 2 and st[-1] + length > total:
            st.pop()
        return st

    tiles = []
    for h0 in starts(H, th, oh):
        h1 = min(h0 + th, H)
        if h1 + sh > H:
            h1 = H
        for w0 in starts(W, tw, ow):
            w1 = min(w0 + tw, W)
            if w1 + sw > W:
                w1 = W
            tiles.append((h0, h1, w0, w1))
    return tiles


def get_valid_tile_region(t0, t1, h0, h1, w0, w1, video_shape, overlap_t, overlap_h, overlap_w):
    """Which part of a processed chunk/tile is kept and where it lands (ref :332-361)."""
    _, _, F, H, W = video_shape
    out = {}
    for key, lo, hi, full, ov in (("t", t0, t1, F, overlap_t), ("h", h0, h1, H, overlap_h), ("w", w0, w1, W, overlap_w)):
        n = hi - lo
        vs = 0 if lo == 0 else ov // 2
        ve = n if hi == full else n - ov // 2
        out[f"valid_{key}_start"], out[f"valid_{key}_end"] = vs, ve
        out[f"out_{key}_start"], out[f"out_{key}_end"] = lo + vs, lo + ve
    return out


def remove_padding_and_extra_frames(video, pad_F: int, pad_H: int, pad_W: int):
    """Crop [B,C,F,H,W] (ref :238-246)."""
    if pad_F > 0:
        video = video[:, :, :-pad_F]
    if pad_H > 0:
        video = video[:, :, :, :-pad_H]
    if pad_W > 0:
        video = video[..., :-pad_W]
    return video


def match_padding(F: int, H: int, W: int):
    """Padding rule of ``preprocess_video_match(is_match=True)`` (ref :220-232): frames -> 8N+1 (repeat last),
    H and W -> multiples of 16 (zeros, bottom/right).  Returns (pad_f, pad_h, pad_w)."""
    rem = (F - 1) % 8
    return (8 - rem if rem else 0), (16 - H % 16) % 16, (16 - W % 16) % 16


def plan(video_shape, chunk_len=0, overlap_t=8, tile_size_hw=(0, 0), overlap_hw=(32, 32)):
    """The (chunk x tile) work list of ref :565-574,682-683, with overlaps zeroed when the axis is untiled."""
    _, _, F, H, W = video_shape
    ov_t = overlap_t if chunk_len > 0 else 0
    ov_hw = tuple(overlap_hw) if tuple(tile_size_hw) != (0, 0) else (0, 0)
    items = []
    for (t0, t1) in make_temporal_chunks(F, chunk_len, ov_t):
        for (h0, h1, w0, w1) in make_spatial_tiles(H, W, tile_size_hw, ov_hw):
            items.append(((t0, t1, h0, h1, w0, w1),
                          get_valid_tile_region(t0, t1, h0, h1, w0, w1, video_shape, ov_t, ov_hw[0], ov_hw[1])))
    return items


def stitch(output, write_count, piece, region):
    """Copy the valid part of a processed piece into ``output`` and bump ``write_count`` (ref :712-720)."""
    r = region
    dst = (slice(None), slice(None), slice(r["out_t_start"], r["out_t_end"]),
           slice(r["out_h_start"], r["out_h_end"]), slice(r["out_w_start"], r["out_w_end"]))
    src = (slice(None), slice(None), slice(r["valid_t_start"], r["valid_t_end"]),
           slice(r["valid_h_start"], r["valid_h_end"]), slice(r["valid_w_start"], r["valid_w_end"]))
    output[dst] = piece[src].to(output.dtype)
    write_count[dst] += 1


def check_coverage(write_count):
    """ref :724-729 exits the process on either condition; here they raise."""
    if bool((write_count == 0).any()):
        raise RuntimeError("Error: Lack of write in region !!!")
    if bool((write_count > 1).any()):
        raise RuntimeError("Error: Write count > 1 in region !!!")
