"""Host tiler vs golden vectors produced by the reference's own functions (tools/make_tiler_goldens.py)."""
import json
import os

import pytest
import torch

from dove_amd import tiling


@pytest.fixture(scope="module")
def gold(golden_dir):
    with open(os.path.join(golden_dir, "tiler_golden.json")) as f:
        return json.load(f)


def test_temporal_chunks(gold):
    assert len(gold["temporal"]) >= 10
    for c in gold["temporal"]:
        if "raises" in c:
            with pytest.raises(ValueError, match=c["msg"]):
                tiling.make_temporal_chunks(c["F"], c["chunk_len"], c["overlap_t"])
        else:
            assert [list(x) for x in tiling.make_temporal_chunks(c["F"], c["chunk_len"], c["overlap_t"])] == c["out"], c


def test_spatial_tiles(gold):
    for c in gold["spatial"]:
        if "raises" in c:
            with pytest.raises(ValueError, match=c["msg"]):
                tiling.make_spatial_tiles(c["H"], c["W"], c["tile"], c["overlap"])
        else:
            assert [list(x) for x in tiling.make_spatial_tiles(c["H"], c["W"], c["tile"], c["overlap"])] == c["out"], c


def test_valid_regions_and_coverage(gold):
    for plan, cov in zip(gold["region"], gold["coverage"]):
        items = tiling.plan(plan["shape"], plan["chunk_len"], plan["overlap_t"] or 8, plan["tile"], plan["overlap_hw"] if plan["tile"] != [0, 0] else (32, 32))
        assert len(items) == len(plan["regions"])
        _, _, F, H, W = plan["shape"]
        wc = torch.zeros(1, 1, F, H, W, dtype=torch.int32)
        out = torch.zeros(1, 1, F, H, W)
        for (args, reg), g in zip(items, plan["regions"]):
            assert list(args) == g["args"]
            assert reg == g["out"]
            t0, t1, h0, h1, w0, w1 = args
            tiling.stitch(out, wc, torch.ones(1, 1, t1 - t0, h1 - h0, w1 - w0), reg)
        assert int(wc.min()) == cov["min"] == 1 and int(wc.max()) == cov["max"] == 1
        tiling.check_coverage(wc)
        assert bool((out == 1).all())


def test_unpad(gold):
    for c in gold["unpad"]:
        out = tiling.remove_padding_and_extra_frames(torch.zeros(c["shape"]), *c["pad"])
        assert list(out.shape) == c["out_shape"]


def test_match_padding():
    # ref :220-232 by hand: 33 frames 180x320 -> pad H to 192; 100 frames -> 105 = 8*13+1
    assert tiling.match_padding(33, 180, 320) == (0, 12, 0)
    assert tiling.match_padding(100, 256, 256) == (5, 0, 0)
    assert tiling.match_padding(8, 178, 316) == (1, 14, 4)
    assert tiling.match_padding(1, 16, 16) == (0, 0, 0)


def test_coverage_errors():
    wc = torch.ones(1, 1, 2, 2, 2, dtype=torch.int32)
    tiling.check_coverage(wc)
    wc[0, 0, 0, 0, 0] = 0
    with pytest.raises(RuntimeError, match="Lack of write"):
        tiling.check_coverage(wc)
    wc[0, 0, 0, 0, 0] = 2
    with pytest.raises(RuntimeError, match="Write count > 1"):
        tiling.check_coverage(wc)
