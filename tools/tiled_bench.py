"""VAE-only timing of the reference's published configuration (--is_vae_st = enable_tiling, ref inference_script.py:642-645) against the
untiled path on the 33x720x1280 clip: encode and decode separately, HIP-event timed, N repetitions.
    python tools/tiled_bench.py [--reps 3] [--mode both|tiled|untiled]
Prints one JSON line: ms per stage and mode, the FLOP ratio tiled/untiled (overlapping tiles recompute), and
tiled_vs_ideal = (untiled_ms x flop_ratio) / tiled_ms  (1.0 = tiling costs exactly its extra FLOPs)."""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dove_amd import config, weights
from dove_amd.vae import AutoencoderKLCogVideoX

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--mode", default="both")
ap.add_argument("--frames", type=int, default=33)
ap.add_argument("--height", type=int, default=720)
ap.add_argument("--width", type=int, default=1280)
ap.add_argument("--c-level", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda", 0)
v, t, s = config.default_configs()
vae = AutoencoderKLCogVideoX(v, weights.LazyStateDict(weights.vae_param_shapes(v), 1234, dev), dev, torch.bfloat16)
video = bench.prepare_clip(bench.synth_lr_clip(args.frames, args.height // 4, args.width // 4, seed=42, device=dev), 4).to(torch.bfloat16)
T = 1 + (args.frames - 1) // 4
z = torch.randn(1, 16, T, args.height // 8, args.width // 8, device=dev, generator=torch.Generator(device=dev).manual_seed(7)).to(torch.bfloat16)


def timed(fn):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(args.reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); o = fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts), o


def tile_flop_ratio(H, W, th, tw, sh, sw):
    hs = [min(th, H - i) for i in range(0, H, sh)]
    ws = [min(tw, W - j) for j in range(0, W, sw)]
    return sum(hs) * sum(ws) / (H * W), len(hs) * len(ws)


res = {}
outs = {}
for mode in (["untiled", "tiled_loop", "tiled_1stream", "tiled_class_streams_only", "tiled", "tiled_3batch_streams", "tiled_4batch_streams"] if args.mode == "both" else [args.mode]):
    if mode.startswith("tiled"):
        vae.enable_slicing(); vae.enable_tiling()
        vae.tile_batching = mode != "tiled_loop"       # tiled_loop: one tile at a time (the round-3 path)
        vae.tile_streams = 1 if mode == "tiled_1stream" else 2
        vae.tile_batch_streams = {"tiled": 2, "tiled_3batch_streams": 3, "tiled_4batch_streams": 4}.get(mode, 1)     # round 6: frame-batches of a class go round-robin to k streams
    else:
        vae.disable_tiling()
    enc_ms, m = timed(lambda: vae.encode(video).latent_dist.parameters)
    dec_ms, d = timed(lambda: vae.decode(z, _range01=True).sample)
    res[mode] = {"encode_ms": enc_ms, "decode_ms": dec_ms, "vae_ms": enc_ms + dec_ms}
    outs[mode] = (m, d)
if args.c_level:
    # the same two modes through the graph-level C entry points (dove_vae_encode / dove_vae_decode with DOVE_OPT_VAE_TILING; one stream)
    import copy
    from dove_amd.graph import GraphContext
    t1 = copy.deepcopy(t); t1["num_layers"] = 1
    ctx = GraphContext(v, t1, weights.LazyStateDict(weights.vae_param_shapes(v), 1234, dev), weights.LazyStateDict(weights.dit_param_shapes(t1), 1234, dev), dev)
    from dove_amd import lib as L_
    for mode in ("untiled_1stream", "untiled", "tiled"):
        ctx.enable_tiling(mode == "tiled")
        ctx.set_option(L_.OPT_VAE_STREAMS, 1 if mode == "untiled_1stream" else 2)     # round 6: frame-batches on two streams below the C boundary too
        enc_ms, m = timed(lambda: ctx.vae_encode(video[0]))
        dec_ms, d = timed(lambda: ctx.vae_decode(z[0], range01=True))
        mode_py = "untiled" if mode.startswith("untiled") else mode
        res["c_level_" + mode] = {"encode_ms": enc_ms, "decode_ms": dec_ms, "vae_ms": enc_ms + dec_ms,
                                  "bit_identical_to_python": (bool(torch.equal(m, outs[mode_py][0][0]) and torch.equal(d, outs[mode_py][1][0])) if mode_py in outs else None),
                                  "workspace_high_water_gb": ctx.workspace_high_water() / 1e9}
p = vae._tiling_params()
re_, ne = tile_flop_ratio(args.height, args.width, p["smin_h"], p["smin_w"], int(p["smin_h"] * (1 - p["of_h"])), int(p["smin_w"] * (1 - p["of_w"])))
rd_, nd = tile_flop_ratio(args.height // 8, args.width // 8, p["lmin_h"], p["lmin_w"], int(p["lmin_h"] * (1 - p["of_h"])), int(p["lmin_w"] * (1 - p["of_w"])))
res["tiles"] = {"encode": ne, "decode": nd, "flop_ratio_encode": re_, "flop_ratio_decode": rd_}
if "tiled_loop" in res and "tiled" in res:
    res["loop_vs_batched_bit_identical"] = all(bool(torch.equal(outs[m][0], outs["tiled_loop"][0]) and torch.equal(outs[m][1], outs["tiled_loop"][1]))
                                               for m in outs if m.startswith("tiled"))
if "untiled" in res and "tiled" in res:
    res["tiled_vs_ideal"] = {"encode": res["untiled"]["encode_ms"] * re_ / res["tiled"]["encode_ms"],
                             "decode": res["untiled"]["decode_ms"] * rd_ / res["tiled"]["decode_ms"],
                             "vae": (res["untiled"]["encode_ms"] * re_ + res["untiled"]["decode_ms"] * rd_) / res["tiled"]["vae_ms"]}
    a, b = outs["untiled"][1].float(), outs["tiled"][1].float()
    res["psnr_decode_tiled_vs_untiled_db"] = float(10 * torch.log10(1.0 / (((a - b) ** 2).mean() + 1e-12)))
print(json.dumps(res))
