"""Generate tests/golden/tiler_golden.json by IMPORTING the reference's own inference_script.py.

Runs only in the build container (needs /root/reference).  The reference module imports several
third-party packages that are absent here (diffusers, torchvision, cv2, pyiqa, imageio, decord);
they are replaced by empty stub modules -- none of them is touched by the pure-Python host helpers
whose outputs we record (make_temporal_chunks :249-279, make_spatial_tiles :282-329,
get_valid_tile_region :332-361, remove_padding_and_extra_frames :238-246, and the padding rule of
preprocess_video_match :220-232 which is restated inline because decord is unavailable).
The JSON holds inputs and expected outputs only (no reference source text).
"""
import importlib.machinery
import importlib.util
import json
import os
import sys
import types

REF = "/root/reference/inference_script.py"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "tiler_golden.json")


def stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def load_reference():
    import transformers  # noqa: F401  (real; must be imported before the stubs)
    import torch  # noqa: F401

    class _Any:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return self

        def __getattr__(self, n):
            return _Any()

    tv = stub("torchvision")
    tv.transforms = stub("torchvision.transforms", ToTensor=_Any, Compose=_Any, Lambda=_Any)
    tv.io = stub("torchvision.io", write_video=_Any())
    d = stub("diffusers", CogVideoXDPMScheduler=_Any, CogVideoXPipeline=_Any)
    d.models = stub("diffusers.models")
    d.models.embeddings = stub("diffusers.models.embeddings", get_3d_rotary_pos_embed=_Any())
    stub("cv2")
    stub("pyiqa")
    io = stub("imageio")
    io.v3 = stub("imageio.v3")
    stub("decord", bridge=_Any())
    spec = importlib.util.spec_from_file_location("ref_inference_script", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    import torch

    ref = load_reference()
    g = {"temporal": [], "spatial": [], "region": [], "unpad": [], "coverage": []}
    for F, cl, ov in [(33, 0, 0), (33, 0, 8), (33, 17, 8), (129, 33, 8), (33, 12, 8), (33, 4, 0), (33, 4, 8), (9, 9, 8),
                      (49, 25, 8), (100, 33, 8), (17, 16, 4), (65, 33, 16), (33, 33, 8), (41, 33, 8), (8, 8, 8), (33, 9, 8)]:
        try:
            g["temporal"].append({"F": F, "chunk_len": cl, "overlap_t": ov, "out": ref.make_temporal_chunks(F, cl, ov)})
        except Exception as e:  # noqa: BLE001
            g["temporal"].append({"F": F, "chunk_len": cl, "overlap_t": ov, "raises": type(e).__name__, "msg": str(e)})
    for H, W, ts, ov in [(1080, 1920, (544, 960), (32, 32)), (720, 1280, (384, 672), (32, 32)), (720, 1280, (0, 0), (32, 32)),
                         (1088, 1920, (544, 960), (32, 32)), (768, 1280, (256, 256), (32, 32)), (512, 512, (256, 256), (64, 64)),
                         (720, 1280, (32, 32), (32, 32)), (256, 256, (256, 256), (32, 32)), (1024, 1024, (384, 384), (32, 32)),
                         (720, 1280, (720, 640), (32, 32)), (200, 300, (128, 128), (16, 48))]:
        try:
            g["spatial"].append({"H": H, "W": W, "tile": list(ts), "overlap": list(ov),
                                 "out": [list(t) for t in ref.make_spatial_tiles(H, W, ts, ov)]})
        except Exception as e:  # noqa: BLE001
            g["spatial"].append({"H": H, "W": W, "tile": list(ts), "overlap": list(ov), "raises": type(e).__name__, "msg": str(e)})
    # valid regions + exact-once coverage for whole chunk x tile plans
    for F, H, W, cl, ovt, ts, ovhw in [(129, 1080, 1920, 33, 8, (544, 960), (32, 32)), (33, 720, 1280, 17, 8, (384, 672), (32, 32)),
                                       (33, 768, 1280, 0, 0, (256, 256), (32, 32)), (49, 512, 512, 25, 8, (256, 256), (64, 64)),
                                       (33, 64, 96, 12, 8, (0, 0), (0, 0))]:
        ov_t = ovt if cl > 0 else 0
        ov_hw = ovhw if tuple(ts) != (0, 0) else (0, 0)
        chunks = ref.make_temporal_chunks(F, cl, ov_t)
        tiles = ref.make_spatial_tiles(H, W, ts, ov_hw)
        shape = (1, 3, F, H, W)
        wc = torch.zeros(F, H, W, dtype=torch.int32)
        regs = []
        for (t0, t1) in chunks:
            for (h0, h1, w0, w1) in tiles:
                r = ref.get_valid_tile_region(t0, t1, h0, h1, w0, w1, shape, ov_t, ov_hw[0], ov_hw[1])
                regs.append({"args": [t0, t1, h0, h1, w0, w1], "out": r})
                wc[r["out_t_start"]:r["out_t_end"], r["out_h_start"]:r["out_h_end"], r["out_w_start"]:r["out_w_end"]] += 1
        g["region"].append({"shape": list(shape), "chunk_len": cl, "overlap_t": ov_t, "tile": list(ts), "overlap_hw": list(ov_hw),
                            "regions": regs})
        g["coverage"].append({"shape": list(shape), "chunk_len": cl, "overlap_t": ov_t, "tile": list(ts), "overlap_hw": list(ov_hw),
                              "min": int(wc.min()), "max": int(wc.max())})
    for shp, pf, ph, pw in [((1, 3, 40, 16, 24), 7, 4, 8), ((1, 3, 33, 768, 1280), 0, 48, 0), ((1, 3, 9, 32, 32), 0, 0, 0),
                            ((1, 3, 17, 64, 64), 3, 0, 16)]:
        out = ref.remove_padding_and_extra_frames(torch.zeros(shp), pf, ph, pw)
        g["unpad"].append({"shape": list(shp), "pad": [pf, ph, pw], "out_shape": list(out.shape)})
    with open(OUT, "w") as f:
        json.dump(g, f, separators=(",", ":"))
    print("wrote", OUT, {k: len(v) for k, v in g.items()})


if __name__ == "__main__":
    main()
