"""bench.py -- SR frames/s of the one-step DOVE path on synthetic 33x720x1280 clips (BASELINE.json configs[1]).

One "step" = one `process_video` call on one [1,3,33,720,1280] clip already resident in HBM: VAE encode ->
posterior sample -> 42-layer DiT at t=399 -> get_velocity -> VAE decode -> [0,1] range map, all through the C-ABI
HIP operators (full CogVideoX1.5-5B architecture, deterministic random-init weights, bf16 storage / fp32
accumulate).  N GPUs = N independent clips (the reference's chunk farm: no data-path collective; weak scaling).  With N > 1 the
same line also carries `single_clip`: ONE clip sharded over the N ranks (BASELINE configs[2]: halo-exact VAE with RCCL send/recv of
the temporal-conv borders + sequence/head-parallel DiT), timed in the same run, with halo bytes per rank and the strong-scaling
efficiency against this run's own one-GPU clip time.

`python bench.py --gpus N` launches its own N ranks (re-exec under torch.distributed.run on 127.0.0.1) when it was not
started by a launcher already; either way every rank asserts WORLD_SIZE == --gpus and that it owns a distinct GPU.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel = implicit-GEMM conv on MFMA, achieved from HIP events
recorded around every launch - in the timed region when the VAE runs one stream, in a one-stream pass of the same process right
behind it when the timed region runs the product's two VAE streams, where an event pair would also span the other stream's
kernels; the line says which) and `cpu_baseline` (the torch-CPU oracle timed
on the host cores on a bounded sample = the full 42-layer model on BASELINE configs[0]'s 9x256x256 clip; rank 0,
N=1 only), plus the PSNR of the HIP path against that oracle run.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from dove_amd import config, flops, ops  # noqa: E402
from dove_amd.inference import process_video  # noqa: E402
from dove_amd.pipeline import CogVideoXPipeline  # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0   # /opt/skills/guides/MI355X_MICROARCH.md: dense bf16 MFMA peak ~2.5 PF


def synth_lr_clip(F=33, H=180, W=320, seed=42, device="cuda"):
    """Synthetic LR clip (uint8-valued): moving low-frequency sinusoids + noise, image-like statistics."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    vid = torch.full((F, 3, H, W), 127.5)
    for c in range(3):
        for _ in range(8):
            fx, fy, ph, amp = torch.rand(4, generator=g)
            for f in range(F):
                vid[f, c] += 40 * amp * torch.sin(6.2832 * (fx * 6 * (xx + f) / W + fy * 6 * yy / H + ph))
    vid += 4 * torch.randn(F, 3, H, W, generator=g)
    return vid.clamp(0, 255).round().to(device)


def prepare_clip(lr, upscale=4):
    """Script-level pre-processing (ref :672-679): bilinear xN, /255*2-1, [1,3,F,H,W].  Outside the timed region."""
    F, C, H, W = lr.shape
    up = torch.nn.functional.interpolate(lr, size=(H * upscale, W * upscale), mode="bilinear", align_corners=False)
    return (up / 255.0 * 2.0 - 1.0).permute(1, 0, 2, 3)[None].contiguous()


def frame_checksums(x):
    """[1, 3, f, H, W] -> int64 [f]: two independently position-weighted sums of each frame's raw bits, folded into one word (wrapping int64
    arithmetic; one frame at a time: ~60 MB of temporaries instead of 1.5 GB for the clip).  Equal checksums on two ranks <=> the same frame
    bits for every practical purpose - a pair of opposite differences that cancels under one weighting (its weights repeat every 65 521
    elements) does not under the other (a multiplicative hash of the position); 8 bytes per frame cross the wire instead of 5.5 MB."""
    f = x.shape[2]
    out = torch.zeros(f, dtype=torch.int64, device=x.device)
    if f == 0:
        return out
    n = x.shape[1] * x.shape[3] * x.shape[4]
    idx = torch.arange(n, device=x.device, dtype=torch.int64)
    w1 = idx % 65521 + 1
    w2 = ((idx * 2654435761) & 0x7FFFFFFF) | 1
    itype = torch.int16 if x.element_size() == 2 else torch.int32
    for i in range(f):
        bits = x[0, :, i].contiguous().view(itype).reshape(-1).to(torch.int64)
        out[i] = (bits * w1).sum() * 1000003 + (bits * w2).sum()
    return out


def cpu_baseline(text, v, t, s, seed, dev):
    """Oracle (clean-room port of the reference's diffusers CPU path) on a bounded sample, all host cores, fp32: the FULL
    model (42 DiT layers, same deterministic weights as the timed GPU run, read tensor-by-tensor from the GPU generator so
    the 22 GB fp32 state dict is never resident) on BASELINE configs[0]'s clip size 9x256x256."""
    from dove_amd import weights
    from oracle import dit as odit
    from oracle.vae import OracleVAE

    cores = min(os.cpu_count() or 1, 64)     # torch-CPU conv3d stops scaling (and oversubscribes) beyond ~64 threads
    torch.set_num_threads(cores)
    F, H, W = 9, 256, 256
    wv = weights.LazyStateDict(weights.vae_param_shapes(v), seed, dev, to="cpu")
    wt = weights.LazyStateDict(weights.dit_param_shapes(t), seed, dev, to="cpu")
    g = torch.Generator().manual_seed(0)
    video = prepare_clip(synth_lr_clip(F, H // 4, W // 4, seed=43, device="cpu"), 4)
    noise = torch.randn(1, 16, 3, H // 8, W // 8, generator=g)
    vae, dit = OracleVAE(v, wv), odit.OracleDiT(t, wt)
    t0 = time.time()
    trace = {}
    ref = odit.process_video(vae, dit, s, video, text.float()[None], noise, trace=trace)
    dt = time.time() - t0
    fl = flops.clip_macs(v, t, F, H, W)["flop"]
    per_frame = flops.clip_macs(v, t, 33, 720, 1280)["flop"] / 33
    base = {"value": fl / dt / per_frame, "unit": "SR frames/s (headline-equivalent: sample TFLOP/s / 35.78 TFLOP per 720p frame)",
            "cores": cores, "kind": "port", "seconds": dt, "tflops": fl / dt / 1e12,
            "sample_frames_per_s": F / dt,
            "sample": f"oracle fp32 process_video on one {F}x{H}x{W} clip (BASELINE configs[0] size), full CogVideoX1.5 VAE + "
                      f"{t['num_layers']}-layer DiT, {fl/1e12:.2f} TFLOP"}
    return base, (video, noise, ref, trace)


def relaunch_if_needed(args):
    """`python bench.py --gpus N` (how the driver calls it) with no launcher: become N ranks, one per GPU."""
    if args.gpus <= 1 or "RANK" in os.environ:
        return
    import socket
    n = torch.cuda.device_count()
    if n < args.gpus and not args.oversubscribe:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but only {n} GPU(s) are visible")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execvp(cmd[0], cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=33)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--layers", type=int, default=None, help="debug only: fewer DiT layers (result marked invalid)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dit-linear", choices=["bf16", "mxfp8"], default="bf16",
                    help="BASELINE configs[4] variant: the DiT's big linears in MXFP8 (block-scaled fp8 MFMA).  A SEPARATE line, "
                         "never the headline: the line says so in `dtype` and `config.variant`")
    ap.add_argument("--dit-attention", choices=["bf16", "mxfp8"], default="bf16",
                    help="same configs[4] variant: attention products on the block-scaled fp8 MFMA; never the headline")
    ap.add_argument("--vae-streams", type=int, choices=[1, 2, 3, 4], default=None,
                    help="HIP streams the VAE's frame-batches alternate on (default: the product's, 2); 1 for the A/B")
    ap.add_argument("--timed-region-events", action="store_true",
                    help="also record the per-launch HIP events inside the two-stream timed region (roofline.avg_launch_ms_timed_region); by "
                         "default the events are recorded only in the one-stream pass behind it, which is where the roofline is taken from")
    ap.add_argument("--no-variants", action="store_true", help="skip the extra (never headline) MXFP8 measurement of the N=1 line")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="debug only: play the N ranks as N processes on GPU 0 over gloo (RCCL refuses two ranks on one device) to "
                         "exercise the multi-rank code path on a one-GPU box; the line is marked invalid")
    ap.add_argument("--single-clip", action="store_true",
                    help="strong scaling: ONE clip sharded over all ranks (halo-exact VAE + Ulysses DiT, dove_amd.dist."
                         "process_video_sharded) instead of one clip per rank; not the driver's default")
    args = ap.parse_args()
    relaunch_if_needed(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or "RANK" in os.environ          # launched by torch.distributed.run: one rank per GPU over RCCL
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # a peer that left must not make the communicator's watchdog abort this process: rank 0 still has the line to print (see the
        # guard around the single-clip mode below)
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")
        if args.oversubscribe:
            local = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    if world != args.gpus:
        raise SystemExit(f"bench.py: launched with WORLD_SIZE={world} but --gpus {args.gpus}")
    if torch.cuda.device_count() < (local + 1):
        raise SystemExit(f"bench.py: rank {rank} has no GPU (LOCAL_RANK {local}, {torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from safetensors.torch import load_file
    text = load_file(os.path.join(ROOT, "tests", "golden", "empty_prompt_embedding.safetensors"))["prompt_embedding"]

    v, t, s = config.default_configs()
    if args.layers is not None:
        t["num_layers"] = args.layers
    t_build = time.time()
    pipe = CogVideoXPipeline.from_config(v, t, s, seed=1234, device=dev, init_device=dev, dit_linear_precision=args.dit_linear,
                                         dit_attention_precision=args.dit_attention)
    torch.cuda.synchronize()
    t_build = time.time() - t_build
    if args.vae_streams is not None:
        pipe.vae.n_streams = args.vae_streams

    up = 4
    strong = args.single_clip and use_dist
    video = prepare_clip(synth_lr_clip(args.frames, args.height // up, args.width // up, seed=42 + (0 if strong else rank),
                                       device=dev), up)
    T = 1 + (args.frames - 1) // 4
    noise = torch.randn(1, 16, T, args.height // 8, args.width // 8, device=dev, generator=torch.Generator(device=dev).manual_seed(7))

    def step_sharded(clip):
        from dove_amd.dist import process_video_sharded
        # every rank keeps the frames it decoded (no gather of the 183 MB clip: each rank would write its own frames)
        return process_video_sharded(pipe, clip, empty_prompt_embedding=text, posterior_noise=noise, gather="none")

    def step():
        if strong:
            return step_sharded(video)
        return process_video(pipe, video, empty_prompt_embedding=text, posterior_noise=noise)

    def barrier():
        if use_dist:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    records = []
    two_stream_region = pipe.vae.n_streams >= 2 and not strong
    if args.timed_region_events or not two_stream_region:
        ops.set_profiler(records)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    ops.set_profiler(None)
    # The product runs the VAE's frame-batches on two HIP streams: an event pair around a launch then also spans the time the kernel shared the
    # chip with the other stream's kernels.  The roofline's per-launch durations therefore come from a ONE-stream pass of the same process,
    # right behind the timed region (same clip, same weights, same launches); the line carries both and says which is which.
    prof_records, prof_steps, prof_elapsed, prof_note = records, args.steps, elapsed, "the timed region (one HIP stream)"
    if two_stream_region:
        keep_streams, pipe.vae.n_streams = pipe.vae.n_streams, 1
        prof_records, prof_steps = [], max(1, min(args.steps, 3))
        step()
        ops.set_profiler(prof_records)
        torch.cuda.synchronize()
        tp0 = time.perf_counter()
        for _ in range(prof_steps):
            step()
        torch.cuda.synchronize()
        prof_elapsed = time.perf_counter() - tp0
        ops.set_profiler(None)
        pipe.vae.n_streams = keep_streams
        prof_note = (f"a ONE-stream pass of {prof_steps} step(s) of this process right behind the timed region ({prof_elapsed / prof_steps * 1e3:.1f} ms per clip "
                     f"against {elapsed / args.steps * 1e3:.1f} in the two-stream timed region, where an event pair also spans co-scheduled kernels of the other stream)")
    per_rank = [elapsed]
    observed_world, gpu_ids = 1, [torch.cuda.get_device_properties(dev).name + f" #{local}"]
    if use_dist:
        import torch.distributed as dist
        observed_world = dist.get_world_size()                    # what RCCL's communicator actually spans
        gdev = "cpu" if args.oversubscribe else dev            # gloo gathers host tensors
        mine = torch.tensor([elapsed, float(local)], device=gdev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(observed_world)]
        dist.all_gather(allr, mine)
        per_rank = [float(x[0]) for x in allr]
        gpu_ids = [f"cuda:{int(x[1])}" for x in allr]
        assert observed_world == args.gpus and (args.oversubscribe or len(set(gpu_ids)) == observed_world), (gpu_ids, observed_world, args.gpus)
        elapsed = max(per_rank)
    if strong:
        assert out is None or (out.shape[0:2] == (1, 3) and out.shape[3:] == (args.height, args.width) and bool(torch.isfinite(out).all()))
    else:
        assert out.shape == (1, 3, args.frames, args.height, args.width) and bool(torch.isfinite(out).all())

    def measure_single_clip():
        fault = os.environ.get("DOVE_BENCH_STRONG_FAULT", "")      # debug only (tests of the guard below): "raise:<rank>" / "hang:<rank>" / "corrupt:<rank>"
        if fault == f"raise:{rank}":
            raise RuntimeError("injected fault (DOVE_BENCH_STRONG_FAULT)")
        if fault == f"hang:{rank}":
            time.sleep(1e6)
        import torch.distributed as dist
        clip0 = video if rank == 0 else prepare_clip(synth_lr_clip(args.frames, args.height // up, args.width // up, seed=42, device=dev), up)
        for _ in range(max(1, args.warmup)):            # the first sharded call records the halo plan (blocking receives)
            step_sharded(clip0)
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            mine = step_sharded(clip0)
        barrier()
        el1 = time.perf_counter() - t1
        info = torch.tensor([el1, 0.0 if mine is None else float(mine.shape[2]), float(getattr(pipe.vae, "last_halo_bytes_encode", 0)),
                             float(getattr(pipe.vae, "last_halo_bytes_decode", 0))], device=dev, dtype=torch.float64)
        info = info.to(gdev)
        infos = [torch.zeros_like(info) for _ in range(observed_world)]
        dist.all_gather(infos, info)
        el1 = max(float(x[0]) for x in infos)
        assert mine is None or bool(torch.isfinite(mine).all())
        assert sum(int(x[1]) for x in infos) == args.frames, "the ranks' decoded frames do not add up to the clip"
        # SELF-VALIDATION of the wires: the frames every rank decoded in this mode against the frames of the ONE-GPU result of the same clip
        # (rank 0's `out` of the weak region above: clip seed 42, same posterior noise, same weights) - per-frame checksums of the raw bits
        # travel, not the tensor.  The mode is bit-identical to the one-GPU operator by construction (tests/test_dist_gpu.py), so anything
        # but equality means the transport delivered other bytes than the kernels sent
        cs = torch.full((args.frames,), -1, dtype=torch.int64, device=dev)
        if fault == f"corrupt:{rank}" and mine is not None:       # debug only: one flipped bit in this rank's last frame must be caught
            mine.view(torch.int16 if mine.element_size() == 2 else torch.int32)[0, 1, -1, 3, 5] ^= 1
        if mine is not None and mine.shape[2] > 0:
            cs[:mine.shape[2]] = frame_checksums(mine)
        cs = cs.to(gdev)
        css = [torch.zeros_like(cs) for _ in range(observed_world)]
        dist.all_gather(css, cs)
        bit_identical, mismatched = None, None
        if rank == 0:
            want = frame_checksums(out).cpu()
            got = torch.cat([c.cpu()[:int(x[1])] for c, x in zip(css, infos)])
            bad = (got != want).nonzero().flatten().tolist()
            bit_identical, mismatched = not bad, bad
        macs1 = flops.clip_macs(v, t, args.frames, args.height, args.width)
        n_tok = macs1["tokens"]
        return {
            "what": "ONE clip sharded over all ranks (BASELINE configs[2]): halo-exact VAE (frame-batches / paired pieces per rank, "
                    "temporal-conv borders by send/recv rank -> rank+1) + sequence/head-parallel DiT (one all_to_all each way per layer); "
                    "bit-identical to the one-GPU result (tests/test_dist_gpu.py)",
            "scaling": "strong", "world_size_observed": observed_world, "steps": args.steps,
            "value": args.steps * args.frames / el1, "unit": "frames/s", "ms_per_clip": el1 / args.steps * 1e3,
            "frames_decoded_per_rank": [int(x[1]) for x in infos],
            "vae_halo_bytes_sent_per_rank": {"encode": [int(x[2]) for x in infos], "decode": [int(x[3]) for x in infos]},
            "dit_all_to_all_bytes_per_rank_per_layer": int(4 * n_tok * t["num_attention_heads"] * t["attention_head_dim"] * 2 / observed_world),
            # strong-scaling efficiency against THIS run's one-GPU clip time (the weak region above: one clip per GPU)
            "one_gpu_ms_per_clip_this_run": elapsed / args.steps * 1e3,
            "efficiency_vs_n1": (elapsed / args.steps) / (observed_world * (el1 / args.steps)),
            "transport": "gloo through host memory (--oversubscribe debug run)" if args.oversubscribe else "RCCL (backend nccl) over xGMI",
            # which process groups carried this rank's halos (dove_amd.dist._comm_label): a rank's receives (from rank-1) and sends (to
            # rank+1) never share a communicator - the neighbour pair (r-1, r) talks on link r % 2 - so pre-posted receives cannot hold sends back
            "halo_communicators_rank0": {"encode": getattr(pipe.vae, "last_halo_stats_encode", {}).get("communicators"),
                                         "decode": getattr(pipe.vae, "last_halo_stats_decode", {}).get("communicators")},
            "bit_identical": bit_identical, "mismatched_frames": mismatched,
            "validation": "per-frame checksums of every rank's decoded frames in THIS run against the frames of rank 0's one-GPU result of the same "
                          "clip (`bit_identical`); the mode's bit-identity with the one-GPU operator is a tested property of the kernels "
                          "(tests/test_dist_gpu.py: 2 / 4 / 8 ranks with gloo-staged wires, and 8 ranks at 33x720x1280) - before this "
                          "measurement the RCCL transport had run with ONE rank only, so a False here points at the wires",
        }

    if rank == 0:
        macs = flops.clip_macs(v, t, args.frames, args.height, args.width)
        ms_step = elapsed / args.steps * 1e3
        value = (1 if strong else world) * args.steps * args.frames / elapsed
        # dominant kernel = conv3x3_halo4x_kernel (VAE 3x3x3 / up-sampling 3x3 convs, ~half of the step).  achieved = sum(algorithmic FLOP)
        # / sum(launch duration) over ITS launches inside the timed region (HIP events on the launch stream).
        def agg(recs, col=5):
            # col 5 = FLOPs the kernel actually issued, col 1 = the reference's algorithmic count (larger where the first-frame / sub-pixel
            # weight sums skip duplicate taps): the roofline is priced on the ISSUED work, the algorithmic rate is reported next to it
            fl = sum(r[col] for r in recs)
            ms = sum(r[2].elapsed_time(r[3]) for r in recs)
            return fl, ms
        DOM = "conv3x3_halo4x_kernel"
        dom = [r for r in prof_records if r[4] == DOM]
        dom_fl, dom_ms = agg(dom)
        tot_fl, tot_ms = agg(prof_records)
        dom_fl_alg, _ = agg(dom, 1)
        dom_timed = [r for r in records if r[4] == DOM]
        dom_ms_timed = agg(dom_timed)[1]
        by = {}
        for key, _fl_alg, e0, e1, var, fl in prof_records:
            k = f"{var}:cin{key[0]}_cout{key[1]}_taps{key[2]}"
            a = by.setdefault(k, [0.0, 0.0, 0])
            a[0] += fl
            a[1] += e0.elapsed_time(e1)
            a[2] += 1
        top = sorted(by.items(), key=lambda kv: -kv[1][1])[:8]
        achieved = dom_fl / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
        all_igemm = tot_fl / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0
        # PMC-derived fields are NOT measured in this run: they are replayed from the committed summary of a separate
        # `rocprofv3 --pmc` pass over this same command (tools/runs/gpu_pmc_bench.sh) and labelled as such
        traffic = None
        pmc_busy = None
        pmc_src = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            with open(pmc) as f:
                pj = json.load(f)
            # the static summary is only replayed when it describes THIS build's launches: the dominant kernel must be in it under the
            # name the library reports now, with the launch count per clip this run just observed (a kernel that was renamed, split or
            # re-dispatched since the PMC pass makes the numbers stale - then the fields stay null and say why)
            pk = pj.get("per_kernel", {}).get(DOM)
            seen = len(dom) // max(prof_steps, 1)
            from dove_amd.lib import kernel_source_sha256
            tree = kernel_source_sha256()
            if pj.get("kernel_source_sha256") != tree:
                # the summary was measured on OTHER kernel code than this tree's (any edit of csrc/*.hip, csrc/*.h or the ABI header)
                pmc_src = (f"profiles/pmc_traffic.json NOT replayed: it was collected on kernel sources {str(pj.get('kernel_source_sha256'))[:16]}, "
                           f"this tree is {tree[:16]} - re-run tools/runs/gpu_pmc_bench.sh")
                pj = None
            elif pk is None or int(pk.get("launches", -1)) != seen:
                pmc_src = (f"profiles/pmc_traffic.json NOT replayed: it holds {None if pk is None else pk.get('launches')} launches of {DOM} per clip, "
                           f"this run made {seen} - re-run tools/runs/gpu_pmc_bench.sh")
                pj = None
        if os.path.exists(pmc) and pj is not None:
            traffic = pj.get("per_kernel", {}).get(DOM, {}).get("hbm_bytes_per_launch")
            pmc_busy = {"whole_step": pj.get("whole_step_mfma_busy_frac"), "per_kernel": pj.get("mfma_busy_frac_per_kernel"),
                        "source": "profiles/pmc_traffic.json (static: separate rocprofv3 --pmc pass, not this run; same kernel sources: "
                                  f"sha256 {tree[:16]})", "note": pj.get("note")}
            pmc_src = "profiles/pmc_traffic.json (static: separate rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE passes, not this run)"
        fp8_parts = (["qkv/out/ff linears"] if args.dit_linear == "mxfp8" else []) + (["attention"] if args.dit_attention == "mxfp8" else [])
        headline = not fp8_parts
        res = {
            "metric": f"SR frames/s ({args.frames}x{args.height}x{args.width} 4x one-step, whole job)", "value": value, "unit": "frames/s",
            "n_gpus": observed_world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "bf16" if headline else "mxfp8 (DiT " + " + ".join(fp8_parts) + ": e4m3 + E8M0 block scales) + bf16 (everything else)",
            "data": "synthetic",
            "config": {"workload": f"synthetic {args.frames}x{args.height}x{args.width} HR clip (LR {args.height//up}x{args.width//up}, 4x), "
                                   f"one-step t=399, CogVideoX1.5-5B VAE + {t['num_layers']}-layer DiT random-init, " + ("ONE clip sharded over all GPUs (BASELINE configs[2])" if strong else "1 clip per GPU (BASELINE configs[1])"),
                       "tokens": macs["tokens"], "pflop_per_clip": macs["flop"] / 1e15,
                       # the arithmetic that was timed: pack-time weight sums ON = first-frame temporal taps, sub-pixel upsample convs and frame pairs
                       # multiply by fp32 sums of the bf16 taps rounded once (fewer MACs; inside every parity gate); OFF = the reference's per-tap
                       # arithmetic (pipe.vae.weight_sums = False / DOVE_OPT_WEIGHT_SUMS), timed below as a variant
                       "weight_sums": bool(pipe.vae.weight_sums), "vae_streams": int(pipe.vae.n_streams),
                       "variant": "headline (bf16)" if headline else "BASELINE configs[4]: fp8 DiT " + " + ".join(fp8_parts) + ", NOT the headline dtype"},
            "frames_per_s_per_gpu": value / world,
            "ranks": {"world_size_observed": observed_world, "gpus": gpu_ids,
                      "busy_s_per_rank": per_rank, "slowest_over_fastest": max(per_rank) / min(per_rank)},
            "whole_path_tflops_per_gpu": macs["flop"] * args.steps / elapsed / 1e12,
            # the same with the taps the weight-summed conv forms skip taken out (what the matrix pipes were actually asked to do)
            "whole_path_tflops_issued_per_gpu": (macs["flop"] - sum(r[1] - r[5] for r in prof_records) / max(prof_steps, 1)) * args.steps / elapsed / 1e12,
            "roofline": {"bound": "mfma", "kernel": DOM + " (persistent LDS-halo implicit-GEMM 3x3(x3) conv, bf16 MFMA 16x16x32)",
                         "achieved": achieved, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / MFMA_BF16_PEAK_TFLOPS,
                         "traffic": traffic, "traffic_source": pmc_src, "launches": len(dom) // max(prof_steps, 1), "avg_launch_ms": dom_ms / max(len(dom), 1),
                         "launch_is": "one dove_conv_igemm_bf16 call = ONE kernel launch at every shape of the untiled clip (dove_conv_partial_launches "
                                      "is 0 for them: profiles/pmc_traffic.json kernel_launches == launches); only the tiled variant's 240 x 360 "
                                      "tile class adds a second launch (conv3x3_halo4x_kernel<..., 1>) for its partial last tile column",
                         "durations_from": prof_note,
                         "avg_launch_ms_timed_region": (dom_ms_timed / len(dom_timed)) if dom_timed else None,
                         "avg_launch_gflop": dom_fl / max(len(dom), 1) / 1e9,
                         "flops_counted": "MFMA work actually issued; the reference's formulation of the same launches is "
                                          f"{dom_fl_alg / max(dom_fl, 1):.4f} x that (first-frame temporal sums, sub-pixel upsample convs skip duplicate taps)",
                         "achieved_algorithmic": dom_fl_alg / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0,
                         "share_of_step_time": dom_ms / (prof_elapsed * 1e3),
                         # SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs) from profiles/pmc_traffic.json
                         # (separate rocprofv3 --pmc pass over this command, tools/runs/gpu_pmc_bench.sh)
                         "mfma_pipe_busy_frac_pmc": pmc_busy,
                         "all_igemm_kernels": {"achieved": all_igemm, "frac": all_igemm / MFMA_BF16_PEAK_TFLOPS,
                                               "share_of_step_time": tot_ms / (prof_elapsed * 1e3), "launches": len(prof_records) // max(prof_steps, 1)},
                         "top_classes": {k: {"ms": a[1] / prof_steps, "tflops": a[0] / (a[1] * 1e-3) / 1e12, "launches": a[2] // prof_steps}
                                         for k, a in top}},
            "model_build_s": t_build,
            # peak of torch's allocator over the run so far (weights 11.6 GB + the clip's activations; two VAE streams = two pools)
            "hbm_peak_allocated_gb": torch.cuda.max_memory_allocated(dev) / 1e9, "hbm_peak_reserved_gb": torch.cuda.max_memory_reserved(dev) / 1e9,
        }
        if (args.frames, args.height, args.width) != (33, 720, 1280):
            res["note"] = "not BASELINE.json's headline clip size (33x720x1280): a side measurement"
        if args.oversubscribe:
            res["invalid"] = "debug run: all ranks share GPU 0 over gloo (--oversubscribe)"
        if strong:
            res["halo_exchange"] = {"vae_halo_bytes_sent_rank0_last_stage": int(getattr(pipe.vae, "last_halo_bytes", 0)),
                                    "mode": "isend + pre-posted irecv per causal conv, GroupNorm pair sums on a side communicator",
                                    "communicators": getattr(pipe.vae, "last_halo_stats_decode", {}).get("communicators")}
        if args.layers is not None:
            res["invalid"] = "debug run with a truncated DiT"
        if world == 1 and not args.no_cpu_baseline:
            base, (svid, snoise, sref, strace) = cpu_baseline(text, v, t, s, 1234, dev)
            res["cpu_baseline"] = base
            # PSNR of the SAME pipeline object the timed region ran (full depth, same weights) vs the fp32 oracle on that sample
            stages = {}
            got = process_video(pipe, svid.to(dev), empty_prompt_embedding=text, posterior_noise=snoise.to(dev), _stages=stages).float().cpu()

            def rms_rel(a, b):
                a, b = a.float().cpu(), b.float().cpu()
                return float(((a - b) ** 2).mean().sqrt() / (b ** 2).mean().sqrt().clamp_min(1e-20))
            # the un-clamped stages of the same run against the oracle's trace: what the image PSNR cannot show on a saturated picture.
            # Gates = the test suite's (tests/test_parity_gpu.py::test_e2e_256_full_model_stagewise)
            se = {k: rms_rel(stages[k], strace[k]) for k in ("moments", "latent", "v", "x0")}
            gates = {"moments": 2e-2, "latent": 2e-2, "v": 6e-2, "x0": 4e-2}
            res["stage_parity"] = {"what": "rms-relative error of the un-clamped stages vs the fp32 oracle on the 9x256x256 sample (full 42-layer model)",
                                   "rms_rel": se, "gates": gates, "passed": all(se[k] <= gates[k] for k in se)}
            mse = ((got - sref) ** 2).flatten(3).mean(-1)
            res["psnr_vs_oracle_db"] = float((10 * torch.log10(1.0 / (mse + 1e-8))).mean())
            # the gated number: only pixels the oracle leaves strictly inside (0, 1) - a clamped pixel contributes zero error and
            # random-init weights clamp about a third of this image, which flatters a plain PSNR
            inside = ((sref > 0) & (sref < 1)).float()
            mse_in = (((got - sref) ** 2) * inside).flatten(3).sum(-1) / inside.flatten(3).sum(-1).clamp_min(1)
            psnr_in = float((10 * torch.log10(1.0 / (mse_in + 1e-8))).mean())
            res["psnr_unsaturated_pixels_db"] = psnr_in
            res["parity_gate"] = {"what": "PSNR over the oracle's un-saturated pixels of the 9x256x256 sample, full 42-layer model, vs the fp32 oracle",
                                  "threshold_db": 35.0, "passed": psnr_in >= 35.0}
            res["psnr_note"] = ("9x256x256 clip, full 42-layer model; random-init weights saturate "
                                f"{100 * float(((sref <= 0) | (sref >= 1)).float().mean()):.0f} % of the reference pixels; the 0.05 dB "
                                "north-star gate against the bf16-emulated reference lives in tests/test_parity_gpu.py")
            if not res["parity_gate"]["passed"]:
                res["invalid"] = f"parity gate failed: {psnr_in:.2f} dB over un-saturated pixels (< 35 dB)"
            if not res["stage_parity"]["passed"]:
                res["invalid"] = f"stage parity gate failed: {se}"
        if world == 1 and headline and not args.no_variants:
            res["variants"] = []
            vsteps = max(1, min(args.steps, 5))

            def timed_variant():
                torch.cuda.empty_cache()        # a variant must not run into the pools the previous configuration left cached (one pool per stream)
                step()
                barrier()
                tv0 = time.perf_counter()
                for _ in range(vsteps):
                    o = step()
                barrier()
                return time.perf_counter() - tv0, o

            # (a) the attention fast path: every head starts on the no-shift pipelined kernel (csrc/attention_pipe.hip); a head whose row sums
            # leave that kernel's window is recomputed with the running maximum inside the same call.  Score bounds <= 80 cannot (random-init
            # weights: ~12); above, it depends on the real scores.  Report which heads did what in this run, the same clip with the q / k
            # LayerNorm gains x 3 each (scores x 9, bound ~108: the regime where the window is CHECKED, not guaranteed), and the clip with the
            # running maximum in every head (norm2 = NULL): the floor
            tr = pipe.transformer

            def path_shares():
                tr.attn_bound_trace, tr.attn_path_trace = [], []
                step()
                torch.cuda.synchronize()
                b_ = torch.stack([1.01 * (n2[:, 0] * n2[:, 1]).sqrt() for n2 in tr.attn_bound_trace]).float().cpu()
                on_pipe = torch.stack([torch.isfinite(n2[:, 0] * n2[:, 1]) for n2 in tr.attn_path_trace]).float().cpu()
                tr.attn_bound_trace = None
                return b_, float(on_pipe.mean())

            b_all, share = path_shares()
            res["attention"] = {"no_shift_heads_frac": share, "score_bound_max": float(b_all.max()),
                                "score_bound_median": float(b_all.median()), "guaranteed_below": 80.0, "heads_x_layers": int(b_all.numel()),
                                "note": "share of (layer, head) pairs the no-shift pipelined kernel finished in this run; a head is handed to the "
                                        "running maximum only when one of its un-shifted row sums leaves [2^-80, 2^100]"}
            qk = [blk[nm][j] for blk in tr.blocks for nm in ("nq", "nk") for j in (0, 1)]
            keep = [w_.clone() for w_ in qk]
            for w_ in qk:
                w_.mul_(3.0)
            b3, share3 = path_shares()
            tv3, _ = timed_variant()
            for w_, k_ in zip(qk, keep):
                w_.copy_(k_)
            del keep
            res["variants"].append({
                "name": "q / k LayerNorm gains and biases x 3 each (every score x 9; score bound median "
                        f"{float(b3.median()):.0f}, max {float(b3.max()):.0f} - above the 80 up to which the no-shift kernel needs no check)",
                "dtype": "bf16", "value": vsteps * args.frames / tv3, "unit": "frames/s", "steps": vsteps, "ms_per_step": tv3 / vsteps * 1e3,
                "speedup_vs_headline_this_run": (vsteps * args.frames / tv3) / value, "no_shift_heads_frac": share3})
            tr.attn_score_bound = False
            tv, o_rm = timed_variant()
            tr.attn_score_bound = True
            res["variants"].append({
                "name": "attention with the running maximum in every head (no score bound handed to dove_attention_fwd_bf16): the weight-independent "
                        "floor of the headline", "dtype": "bf16", "value": vsteps * args.frames / tv, "unit": "frames/s", "steps": vsteps,
                "ms_per_step": tv / vsteps * 1e3, "speedup_vs_headline_this_run": (vsteps * args.frames / tv) / value,
                "psnr_vs_headline_output_db": float(10 * torch.log10(1.0 / (((o_rm.float() - out.float()) ** 2).mean() + 1e-12))),
                "psnr_note": "same function, other summation order in the softmax; 42 random-init layers amplify last-bit differences (the 2-layer "
                             "agreement gate is tests/test_parity_gpu.py::test_dit_mixed_softmax_paths_wide_qk_gains)"})
            # (a') the cost of exactness: no pack-time weight sum anywhere - every conv launch computes the reference's per-tap arithmetic
            pipe.vae.weight_sums = False
            tvw, o_w = timed_variant()
            pipe.vae.weight_sums = True
            res["variants"].append({
                "name": "pipe.vae.weight_sums = False: the reference's per-tap conv arithmetic (no first-frame / sub-pixel / frame-pair weight sums)",
                "dtype": "bf16", "value": vsteps * args.frames / tvw, "unit": "frames/s", "steps": vsteps, "ms_per_step": tvw / vsteps * 1e3,
                "speedup_vs_headline_this_run": (vsteps * args.frames / tvw) / value,
                "psnr_vs_headline_output_db": float(10 * torch.log10(1.0 / (((o_w.float() - out.float()) ** 2).mean() + 1e-12)))})
            del o_w
            # (b) the reference's PUBLISHED configuration: --is_vae_st = pipe.vae.enable_slicing() + enable_tiling() (inference.sh:8,
            # inference_script.py:642-645): 4 x 5 overlapping 240x360-px tiles per VAE stage, blended.  Tiling recomputes the overlaps
            # (FLOP ratio below), so "ideal" = the untiled clip time with the VAE share scaled by that ratio
            vrecs = []
            pipe.vae.enable_slicing()
            pipe.vae.enable_tiling()
            ops.set_profiler(vrecs)
            tv, o_t = timed_variant()
            ops.set_profiler(None)
            pipe.vae.disable_tiling()
            pipe.vae.disable_slicing()
            tp = pipe.vae._tiling_params()

            def cover(n, tile, stride):
                return sum(min(tile, n - i) for i in range(0, n, stride)) / n
            ratio = (cover(args.height, tp["smin_h"], int(tp["smin_h"] * (1 - tp["of_h"]))) *
                     cover(args.width, tp["smin_w"], int(tp["smin_w"] * (1 - tp["of_w"]))))
            vae_fl = 2.0 * (macs["encode"] + macs["decode"])
            by_t = {}
            for key, _fl_alg, e0, e1, var, fl in vrecs[len(vrecs) // (vsteps + 1):]:       # drop the warm-up step's records
                a = by_t.setdefault(var, [0.0, 0.0, 0])
                a[0] += fl
                a[1] += e0.elapsed_time(e1)
                a[2] += 1
            ent = {
                "name": "vae_tiling: the reference's published configuration (--is_vae_st: enable_slicing + enable_tiling, 240x360-px tiles with "
                        "1/6 and 1/5 overlaps, blended); all tiles of one shape run as one batch (dove_conv_desc.nb), edge classes on a second stream",
                "dtype": "bf16", "value": vsteps * args.frames / tv, "unit": "frames/s", "steps": vsteps, "ms_per_step": tv / vsteps * 1e3,
                "speedup_vs_headline_this_run": (vsteps * args.frames / tv) / value,
                "vae_flop_ratio_tiled_over_untiled": ratio,
                "kernels": {k: {"ms": a[1] / vsteps, "tflops": a[0] / (a[1] * 1e-3) / 1e12, "launches": a[2] // vsteps} for k, a in by_t.items()},
                "kernels_note": "HIP-event durations per launch, summed; the edge tile classes run on a second stream, so the sums include co-scheduling "
                                "waits and add up to more than the step (one-stream rocprofv3 summary: profiles/r04_tiled_batched_1stream_kernel_stats.csv)",
                "psnr_vs_untiled_output_db": float(10 * torch.log10(1.0 / (((o_t.float() - out.float()) ** 2).mean() + 1e-12))),
                "psnr_note": "tiling is a DIFFERENT function of the clip (GroupNorm statistics per tile, blended seams) - diffusers' too; parity of "
                             "the tiled path is gated against the oracle's tiled restatement (tests/test_e2e_gpu.py::test_vae_tiling*)",
            }
            whole_ratio = (vae_fl * ratio + macs["flop"] - vae_fl) / macs["flop"]
            ent["clip_flop_ratio_tiled_over_untiled"] = whole_ratio
            ent["throughput_vs_untiled_times_flop_ratio"] = (vsteps * args.frames / tv) / (value / whole_ratio)
            res["variants"].append(ent)
        if world == 1 and headline and not args.no_variants and args.layers is None:
            # BASELINE configs[4] measured in the SAME run on the same clip (never the headline): the DiT rebuilt with MXFP8 linears +
            # attention (same seed), the VAE object shared.  PSNR gates of this variant: tests/test_parity_gpu.py::test_mxfp8_dit_psnr_gate
            del pipe.transformer
            torch.cuda.empty_cache()
            from dove_amd import weights as W_
            from dove_amd.transformer import CogVideoXTransformer3DModel
            pipe.transformer = CogVideoXTransformer3DModel(t, W_.LazyStateDict(W_.dit_param_shapes(t), 1234, dev), dev, torch.bfloat16,
                                                           "mxfp8", "mxfp8")
            ref_out = out
            out8 = step()
            barrier()
            tv = time.perf_counter()
            for _ in range(vsteps):
                out8 = step()
            barrier()
            tv = time.perf_counter() - tv
            mse = ((out8.float() - ref_out.float()) ** 2).flatten(3).mean(-1)
            res["variants"].append({
                "name": "BASELINE configs[4]: DiT linears + attention in MXFP8 (e4m3 + E8M0 block scales, v_mfma_scale_f32_32x32x64_f8f6f4), "
                        "everything else bf16 - NOT the headline dtype",
                "dtype": "mxfp8 + bf16", "value": vsteps * args.frames / tv, "unit": "frames/s", "steps": vsteps, "ms_per_step": tv / vsteps * 1e3,
                "speedup_vs_headline_this_run": (vsteps * args.frames / tv) / value,
                "psnr_vs_bf16_path_db_this_clip": float((10 * torch.log10(1.0 / (mse + 1e-8))).mean()),
                "psnr_note": "full-size clip, random-init weights (saturated output); the un-saturated 42-layer gate is in tests/test_parity_gpu.py"})
    # ---- N > 1: the same ranks now run ONE clip together (BASELINE configs[2]); same barrier / max-over-ranks timing.  This mode
    # has run on RCCL with one rank only (no multi-GPU box was available to the builder; tests: gloo, R = 2 / 4 / 8 processes on one GPU), so it
    # must not be able to take the weak-scaling line - measured and assembled above - down with it: an exception on any rank, or a
    # collective that never returns, ends in `single_clip: {"error": ...}` on the ONE line rank 0 prints, and every rank leaves with 0 ----
    if use_dist and world > 1 and not strong:
        import threading
        limit = float(os.environ.get("DOVE_BENCH_STRONG_TIMEOUT", 180.0 + 6.0 * elapsed * (1.0 + args.warmup / max(args.steps, 1))))

        def leave(err):
            if rank == 0:
                res["single_clip"] = {"error": err, "note": "the weak-scaling fields of this line were measured before this mode ran and are unaffected"}
                res["single_clip_failed"] = True               # a reader of the line must not take the missing numbers for a pass
                print(json.dumps(res), flush=True)
            sys.stdout.flush()
            os._exit(0)                                        # not sys.exit: a communicator with a dead peer may hang in its destructor

        dog = threading.Timer(limit, leave, args=(f"single-clip mode did not finish within {limit:.0f} s (a rank failed or a collective hung)",))
        dog.daemon = True
        dog.start()
        try:
            single = measure_single_clip()
        except BaseException as e:                             # noqa: BLE001 - report, never crash the line
            leave(f"{type(e).__name__}: {e}")
        dog.cancel()
        if rank == 0:
            res["single_clip"] = single
            if single.get("bit_identical") is not True:
                res["single_clip_failed"] = True           # the sharded mode did not reproduce the one-GPU frames: its numbers are not a result
    if rank == 0:
        print(json.dumps(res), flush=True)
    if use_dist:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
