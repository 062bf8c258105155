"""Multi-GPU execution of the one-step SR path: one process per GPU, torch.distributed ("nccl" = RCCL over
xGMI on the GPU box, "gloo" in CPU tests).  SURVEY.md 8(e).

(A) Chunk farm -- the reference's own semantics: every (time-chunk x spatial-tile) of
    /root/reference/inference_script.py:690-703 is an independent ``process_video`` call, so work items are dealt
    round-robin to ranks; NO data-path collective, only the final stitch (a sum of disjoint pieces).  This is what
    ``bench.py --gpus N`` scales (one clip per GPU, weak scaling).

(B) Halo-exact single clip -- not in the reference; the distributed form of diffusers' ``conv_cache``.  The VAE's
    frame-batches (8 px-frames / 2 latent frames) are split into contiguous groups, one
    group per rank; every k_t=3 CausalConv3d of the first local batch receives its 2-frame temporal halo (the last
    two INPUT frames of that conv in the previous rank's last batch) with a point-to-point recv from rank-1, and
    the last local batch sends its own to rank+1 -- neighbour traffic on ONE xGMI link, in layer order, so ranks
    run as a wavefront skewed by one layer.  GroupNorm statistics are per frame-batch; with at most as many ranks as
    frame-batches batches are never split, no reduction is needed and results are bit-identical to the single-GPU path.
    With more ranks (8 GPUs on a 33-frame clip: BASELINE's "frame-chunk = 4") a batch is split into two pieces on a rank
    pair which combines its GroupNorm sums with one send/recv per norm (``plan_pieces``).
    The DiT attends over all tokens of the clip (not frame-separable): moments are all-gathered and the DiT runs either
    replicated or (C) sequence/head-parallel.

(C) Ulysses-style DiT for the single-clip mode: the [N, 3072] residual stream is sharded by ROWS (tokens; text rows
    first) for everything that is row-local - LayerNormZero, the QKV / out / FFN linears with their gated residuals,
    QK-LayerNorm + RoPE - and by HEADS for attention (48 heads / R ranks): one all_to_all moves Q', K', V^T from
    "my rows, all heads" to "all rows, my heads" before the flash-attention kernel and one moves its output back, per
    layer (4 x N x 3072 x 2 B / R per rank and layer over xGMI).  Every row and every head sees exactly the arithmetic of
    the single-GPU path, so the result is bit-identical (gloo test with 2 and 3 ranks); `process_video_sharded` chains
    (B) encode -> (C) DiT -> (B) decode.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import tiling
from .inference import process_video
from .vae import DiagonalGaussianDistribution, frame_batches


# ---- wires ---------------------------------------------------------------------------------------------
# RCCL ("nccl") moves device tensors directly.  Any other backend (gloo: the CPU tests, and the GPU test that plays R ranks as R
# processes on ONE device, tests/test_dist_gpu.py - RCCL refuses two ranks on one GPU) gets bytes, and device tensors are staged
# through host memory explicitly instead of relying on the backend's optional device support.  Same messages, same order.
def _direct(group):
    return dist.get_backend(group) == "nccl"


def _bytes(t):
    return t.view(torch.uint8) if t.dtype == torch.bfloat16 else t


class _StagedRecv:
    """``irecv`` into host memory on behalf of a device buffer: ``wait()`` completes the receive, then uploads."""

    def __init__(self, buf, src, group):
        self._buf, self._host = buf, torch.empty(buf.shape, dtype=buf.dtype, device="cpu")
        self._work = dist.irecv(_bytes(self._host), src=src, group=group)

    def wait(self):
        self._work.wait()
        self._buf.copy_(self._host)


_wire_log = None     # tests: a list -> every point-to-point call appends (communicator label, "send" | "recv" | "swap", global peer rank)


def _comm_label(group):
    """Name of a communicator that means the same on every rank: "link0" / "link1" (``_links``), "side" (``_side_group``), else "stage"
    (the group the sharded call was given).  The serials of ``_group_key`` are per process."""
    for ls in _link_groups.values():
        for i, g in enumerate(ls):
            if g is group:
                return f"link{i}"
    if any(g is group for g in _side_groups.values()):
        return "side"
    return "stage"


def _log_wire(group, op, peer):
    if _wire_log is not None:
        _wire_log.append((_comm_label(group), op, int(peer)))


def _isend(t, dst, group, log=True):
    """-> (tensor to keep alive, work).  ``dst`` is a global rank."""
    if log:
        _log_wire(group, "send", dst)
    if _direct(group):
        return t, dist.isend(t, dst=dst, group=group)
    h = t.cpu() if t.is_cuda else t                    # .cpu() waits for the producing stream
    return h, dist.isend(_bytes(h), dst=dst, group=group)


def _irecv(buf, src, group, log=True):
    if log:
        _log_wire(group, "recv", src)
    if _direct(group):
        return dist.irecv(buf, src=src, group=group)
    if buf.is_cuda:
        return _StagedRecv(buf, src, group)
    return dist.irecv(_bytes(buf), src=src, group=group)


def _send(t, dst, group):
    _isend(t, dst, group)[1].wait()


def _exchange(mine, theirs, peer, group):
    """Symmetric swap with one peer, both directions in flight at once.  On RCCL the pair must be ONE grouped call: two ranks that
    each enqueue recv-then-send separately wait on each other's send forever."""
    _log_wire(group, "swap", peer)
    if _direct(group):
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, mine, peer, group), dist.P2POp(dist.irecv, theirs, peer, group)]):
            w.wait()
        return
    rw = _irecv(theirs, peer, group, log=False)                 # (the pair of calls is this ONE swap on the log)
    keep, sw = _isend(mine, peer, group, log=False)
    rw.wait()
    sw.wait()


def _recv(buf, src, group):
    _irecv(buf, src, group).wait()


def _all_gather(outs, t, group):
    if _direct(group) or not t.is_cuda:
        dist.all_gather([_bytes(o) for o in outs], _bytes(t), group=group)
        return
    hs = [torch.empty(o.shape, dtype=o.dtype) for o in outs]
    dist.all_gather([_bytes(h) for h in hs], _bytes(t.cpu()), group=group)
    for o, h in zip(outs, hs):
        o.copy_(h)


def _all_to_all(out, inp, out_splits, in_splits, group):
    """all_to_all_single over flat bf16 buffers; splits in ELEMENTS."""
    ob, ib = [2 * v for v in out_splits], [2 * v for v in in_splits]
    if _direct(group) or not inp.is_cuda:
        dist.all_to_all_single(out.view(torch.uint8), inp.view(torch.uint8), ob, ib, group=group)
        return
    h = torch.empty(out.shape, dtype=out.dtype)
    dist.all_to_all_single(h.view(torch.uint8), inp.cpu().view(torch.uint8), ob, ib, group=group)
    out.copy_(h)


def _broadcast(t, src, group):
    """In place on ``t`` (contiguous)."""
    if _direct(group) or not t.is_cuda:
        dist.broadcast(_bytes(t), src=src, group=group)
        return
    h = t.cpu()
    dist.broadcast(_bytes(h), src=src, group=group)
    t.copy_(h)


_group_serial: "weakref.WeakKeyDictionary" = None
_group_count = [0]


def _group_key(group):
    """Identity of a process group for plan caches and the poison list: its rank tuple and backend (``id(group)`` alone can be reused
    after a group is destroyed) plus a serial number handed out the first time THIS group object is seen - a group re-created over the
    same ranks (what the poison message asks for) is a new object, gets a new serial, and starts with a clean record."""
    global _group_serial
    import weakref
    if _group_serial is None:
        _group_serial = weakref.WeakKeyDictionary()
    g = group if group is not None else dist.group.WORLD
    try:
        n = _group_serial.get(g)
        if n is None:
            _group_count[0] += 1
            n = _group_serial[g] = _group_count[0]
    except TypeError:                                     # a group type that cannot be weakly referenced: fall back to the object id
        n = id(g)
    return (tuple(dist.get_process_group_ranks(g)), dist.get_backend(g), n)


_poisoned: set = set()          # groups on which a sharded call failed part-way (peers may be mismatched from then on)


# ---- (A) chunk farm ---------------------------------------------------------------------------------
def owns(i: int, n: int, rank: int, world: int) -> bool:
    """Round-robin ownership of work item i of n."""
    return i % world == rank


@torch.no_grad()
def run_clip_distributed(pipe, video, *, group=None, chunk_len=0, overlap_t=8, tile_size_hw=(0, 0), overlap_hw=(32, 32),
                         empty_prompt_embedding=None, sr_noise_step=399, seeds=None):
    """Reference chunk/tile loop with the work items sharded over the ranks of ``group``; every rank returns the full
    stitched [1,3,F,H,W] fp32 host tensor (pieces are disjoint, so the merge is an all-reduce SUM) and the write
    count, on which the reference's exact-once coverage check is applied."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    items = tiling.plan(video.shape, chunk_len, overlap_t, tile_size_hw, overlap_hw)
    out = torch.zeros(video.shape, dtype=torch.float32)
    wc = torch.zeros(video.shape, dtype=torch.int32)
    for i, ((t0, t1, h0, h1, w0, w1), region) in enumerate(items):
        if not owns(i, len(items), rank, world):
            continue
        gen = None
        if seeds is not None:   # per-item generator so the result does not depend on which rank ran the item
            gen = torch.Generator(device=pipe.vae.device).manual_seed(int(seeds) + i)
        piece = process_video(pipe, video[:, :, t0:t1, h0:h1, w0:w1], sr_noise_step=sr_noise_step,
                              empty_prompt_embedding=empty_prompt_embedding, generator=gen)
        tiling.stitch(out, wc, piece.float().cpu(), region)
    backend = dist.get_backend(group)
    if backend == "nccl":
        dev = pipe.vae.device
        o, w = out.to(dev), wc.to(dev)
        dist.all_reduce(o, group=group)
        dist.all_reduce(w, group=group)
        out, wc = o.cpu(), w.cpu()
    else:
        dist.all_reduce(out, group=group)
        dist.all_reduce(wc, group=group)
    tiling.check_coverage(wc)
    return out, wc


# ---- (B) halo-exact VAE -------------------------------------------------------------------------------
def split_batches(batches, world):
    """Contiguous groups of frame-batches, one per rank (earlier ranks get the extras)."""
    n = len(batches)
    base, extra = divmod(n, world)
    out, s = [], 0
    for r in range(world):
        k = base + (1 if r < extra else 0)
        out.append(batches[s:s + k])
        s += k
    return out


class HaloCache(dict):
    """conv_cache dict whose misses on the first local batch are filled by a receive from rank-1 and whose final entries
    are sent to rank+1.  ``phase`` is set by the VAE driver loop: 'first' / 'last' / 'both' / 'mid'.

    The exchange overlaps with compute (~75 ms of halo traffic vs ~118 ms of compute per rank at 8 GPUs):
      * sends are ``isend``: on RCCL the transfer runs on the communicator's stream behind an event recorded at the call, so
        the producing stream goes straight on to the conv; the handles (and tensors) are kept until ``finish()``;
      * receives are PRE-POSTED: the order and shapes of a stage's causal convs are a function of (stage, input shape)
        only, so the first run with a given shape records the plan (blocking receives) and every later run posts all of
        its ``irecv``s up front, in conv order, into preallocated buffers - conv k's halo streams in while convs < k run,
        and ``fetch`` merely waits on its handle (a stream-side wait on RCCL).
    Order and bytes on the wire are those of the blocking version; results are bit-identical (gloo tests).

    COMMUNICATORS.  A rank's receives (from rank-1) and its sends (to rank+1) must never share a communicator: RCCL executes the
    point-to-point operations of one communicator in issue order on one stream (torch with eager init - ``device_id=`` - says so itself:
    "unbatched P2P ops are ... serialized with all other ops on this ProcessGroup"), so a stage's pre-posted receives, queued first, would
    hold every send of the rank back until its LAST halo had arrived - rank r+1 would start when rank r-1 had finished the stage and the
    one-layer-skewed wavefront would degrade to ~R/2 stage times.  ``links`` = two extra world-spanning groups (``_link_groups``); the
    neighbour pair (r-1, r) talks on ``links[r % 2]``: rank r receives on ``links[r % 2]`` and sends on ``links[(r+1) % 2]`` - on either
    communicator a rank only ever receives or only ever sends (tests/test_dist_cpu.py::test_halo_wavefront_communicators).  Without
    links (a sub-group: ``new_group`` is collective over the WORLD, its non-members never get here) nothing is pre-posted: each halo is
    received right where it is consumed, so receives and sends interleave in layer order on the one communicator."""

    _plans: dict = {}                                   # (stage key) -> [(name, shape)] of the first local batch

    def __init__(self, group, rank, world, plan_key=None, links=None):
        super().__init__()
        self.group, self.rank, self.world = group, rank, world
        self.recv_group = links[rank % 2] if links else group
        self.send_group = links[(rank + 1) % 2] if links else group
        self.can_prepost = bool(links)
        self.phase = "mid"
        self.bytes_sent = 0
        self.plan_key = plan_key
        self._record = []
        self._posted = {}                               # name -> (buffer, work)
        self._sends = []
        self._dev = None
        self.recv_blocking = 0                          # halos this stage fetched with a blocking receive (the recording pass)
        self.recv_preposted = 0                         # ... and from a receive posted before the stage's first kernel
        self.msg_bytes = []                             # size of every halo message sent, in conv order

    def _peer(self, r):
        return dist.get_global_rank(self.group, r) if self.group else r

    def prepost(self, device):
        """Post every receive of the first local batch when this (stage, shape) has been seen before."""
        self._dev = device
        plan = HaloCache._plans.get(self.plan_key)
        if plan is None or self.rank == 0 or not self.can_prepost:
            return
        for name, shape in plan:
            buf = torch.empty(shape, dtype=torch.bfloat16, device=device)
            self._posted[name] = (buf, _irecv(buf, self._peer(self.rank - 1), self.recv_group))

    def fetch(self, name, like_shape, device):
        """Halo for conv ``name`` of the first local batch (None on rank 0: replicate-first-frame padding)."""
        if self.phase in ("first", "both") and self.rank > 0:
            if name in self._posted:
                buf, work = self._posted.pop(name)
                assert tuple(buf.shape) == tuple(like_shape), (name, buf.shape, like_shape)
                work.wait()
                self.recv_preposted += 1
                return buf
            buf = torch.empty(like_shape, dtype=torch.bfloat16, device=device)
            _recv(buf, self._peer(self.rank - 1), self.recv_group)
            self._record.append((name, tuple(like_shape)))
            self.recv_blocking += 1
            return buf
        return self.get(name)

    def publish(self, name, new):
        self[name] = new
        if self.phase in ("last", "both") and self.rank < self.world - 1:
            self._sends.append(_isend(new.contiguous(), self._peer(self.rank + 1), self.send_group))
            self.bytes_sent += new.numel() * 2
            self.msg_bytes.append(new.numel() * 2)

    def stats(self):
        return dict(bytes_sent=self.bytes_sent, messages=list(self.msg_bytes), recv_blocking=self.recv_blocking, recv_preposted=self.recv_preposted,
                    communicators=dict(recv=_comm_label(self.recv_group), send=_comm_label(self.send_group), stage=_comm_label(self.group)))

    def finish(self):
        """End of the stage: every send has left, every posted receive was consumed, the plan is remembered."""
        for _, work in self._sends:
            work.wait()
        self._sends.clear()
        assert not self._posted, f"pre-posted halos never consumed: {list(self._posted)}"
        if self._record and self.plan_key is not None:
            HaloCache._plans[self.plan_key] = list(self._record)

    def abandon(self):
        """A stage failed part-way: forget the plan and refuse the group from now on (``_run_sharded``) instead of computing on stale
        buffers - receives that are still posted cannot be recalled and would pair with the NEXT call's halos."""
        HaloCache._plans.pop(self.plan_key, None)
        # ANY failure inside a multi-rank stage can leave the peers mismatched (a blocking receive of the recording pass that never
        # returned, sends still in flight, receives still posted): the next sharded call could pair its halos with stale messages
        if self.world > 1:
            _poisoned.add(_group_key(self.group))
        self._posted.clear()
        self._sends.clear()


def plan_pieces(batches, world, kind):
    """Work list per rank for the halo-exact VAE.  With at most as many ranks as frame-batches every rank gets a contiguous
    group of whole batches (``split_batches``).  With more ranks, batches are split in two PIECES handed to consecutive
    ranks (a rank pair): the conv halos flow rank -> rank+1 exactly as between batches, the GroupNorm statistics of the
    batch are combined across the pair, and Upsample3D's first-frame rule is told which piece starts the batch.
    Split points keep every temporal 2:1 pooling pair inside one piece: px-frame batches (``kind == "enc"``) of 8k(+1)
    frames split at 4k(+1); latent batches (``kind == "dec"``) of 2 or 3 frames split before the last frame.
    Returns (per-rank list of dicts {s, e, role, partner, odd}, number of active ranks)."""
    nb = len(batches)
    if world <= nb:
        groups = split_batches(batches, world)
        return [[dict(s=s, e=e, role=None, partner=None) for s, e in g_] for g_ in groups], sum(1 for g_ in groups if g_)
    out, r = [[] for _ in range(world)], 0
    for i, (s, e) in enumerate(batches):
        n = e - s
        spare = (world - r) - (nb - i)               # ranks we can still spend on splitting
        if kind == "enc":
            head = (n % 2) + 4 * ((n - n % 2) // 8)
            can = n - (n % 2) >= 8 and (n - n % 2) % 8 == 0
        else:
            head = n - 1
            can = n >= 2
        if spare >= 1 and can:
            odd = n % 2 == 1
            out[r].append(dict(s=s, e=s + head, role="head" if odd else "tail", partner=r + 1, lower=True))
            out[r + 1].append(dict(s=s + head, e=e, role="tail", partner=r, lower=False))
            r += 2
        else:
            out[r].append(dict(s=s, e=e, role=None, partner=None))
            r += 1
    return out, r


_side_groups: dict = {}
_link_groups: dict = {}


def _links(group):
    """Two more communicators over the world for the halo wavefront (HaloCache, COMMUNICATORS): created collectively on first use for the
    default group - every rank reaches ``_run_sharded`` with the same arguments - and None for a sub-group."""
    if group is not None and group is not dist.group.WORLD:
        return None
    key = _group_key(None)
    if key not in _link_groups:
        _link_groups.clear()                            # groups of a destroyed world are gone with it
        be = dist.get_backend()
        _link_groups[key] = (dist.new_group(backend=be), dist.new_group(backend=be))
    return _link_groups[key]


def _side_group(group):
    """A second communicator over the same ranks for the GroupNorm pair exchange of split frame-batches.  Point-to-point
    messages between two ranks are matched strictly in order per communicator (RCCL has no tags), and pre-posted halo
    receives would otherwise pair up with the partner's GroupNorm sums sent in between.  Created collectively on first use
    for the default group; for a sub-group (whose non-members never get here) there is none and receives stay un-posted."""
    if group is not None and group is not dist.group.WORLD:
        return None
    key = _group_key(None)
    if key not in _side_groups:
        _side_groups.clear()
        _side_groups[key] = dist.new_group(backend=dist.get_backend())
    return _side_groups[key]


def _run_sharded(vae, x_cl, batches, world, rank, group, fn, kind="enc"):
    """Run ``fn(tensor, cache)`` over this rank's frame-batches or pieces of frame-batches (``plan_pieces``)."""
    from . import ops
    plan, active = plan_pieces(batches, world, kind)
    mine = plan[rank]
    paired = any(pc["partner"] is not None for r in plan for pc in r)
    gn_group = _side_group(group) if paired else group
    links = _links(group)                               # both collective on first use: before anything that can differ between ranks
    gkey = _group_key(group)
    if gkey in _poisoned:
        raise RuntimeError("dove_amd.dist: an earlier sharded call on this process group failed part-way (its peers may hold unmatched halo messages); "
                           "destroy and re-create the process group before the next sharded call")
    cache = HaloCache(group, rank, active, plan_key=(kind, tuple(x_cl.shape), world, rank, gkey), links=links)
    if not paired or gn_group is not None:
        cache.prepost(x_cl.device)
    if gn_group is None:
        gn_group = group
    outs = []

    def gsrc(r):
        return dist.get_global_rank(group, r) if group else r      # the side group spans the same ranks in the same order

    try:
        _run_pieces(vae, x_cl, mine, cache, fn, outs, gsrc, gn_group, ops)
    except BaseException:
        cache.abandon()
        raise
    cache.finish()
    return outs, cache


def _run_pieces(vae, x_cl, mine, cache, fn, outs, gsrc, gn_group, ops):
    for i, pc in enumerate(mine):
        first, last = i == 0, i == len(mine) - 1
        cache.phase = "both" if first and last else ("first" if first else ("last" if last else "mid"))
        if pc["partner"] is not None:
            partner, lower = pc["partner"], pc["lower"]

            def hook(x, partner=partner, lower=lower):
                # whole-batch GroupNorm statistics = my piece's (sum, sumsq, count) + the partner's, same fp64 finalize
                sums = ops.groupnorm_sums_of(x)          # from the producing conv's epilogue rows when it wrote any
                cnt = float(x.numel() // 32)
                mine_msg = torch.cat([sums.reshape(-1), torch.tensor([cnt], dtype=torch.float64, device=sums.device)])
                theirs = torch.empty_like(mine_msg)
                _exchange(mine_msg, theirs, gsrc(partner), gn_group)     # 65 doubles each way, ONE round trip per norm
                a, b = (mine_msg, theirs) if lower else (theirs, mine_msg)      # same summation order on both ranks
                return ops.groupnorm_from_sums(a + b, None, vae.eps)     # element count = the 65th double: no host read-back

            vae._gn_hook, vae._piece_role = hook, pc["role"]
        try:
            outs.append(fn(x_cl[pc["s"]:pc["e"]], cache))
        finally:
            vae._gn_hook, vae._piece_role = None, None


def _gather_time(parts, group, world, device, to="all"):
    """Gather variable-length [T_r, ...] tensors along T.  ``to="all"``: every rank gets the whole tensor (all_gather);
    ``to="writer"``: only rank 0 does (the others send their frames once and return None) - the decoded clip is 183 MB and
    nobody but the writer needs it."""
    local = torch.cat(parts, dim=0) if parts else None
    if to == "writer":
        rank = dist.get_rank(group)
        peer = (lambda r: dist.get_global_rank(group, r)) if group else (lambda r: r)
        meta = torch.tensor(list(local.shape) if local is not None else [0, 0, 0, 0], dtype=torch.int64, device=device)
        metas = [torch.zeros_like(meta) for _ in range(world)]
        _all_gather(metas, meta, group)
        if rank != 0:
            if local is not None:
                _send(local.contiguous(), peer(0), group)
            return None
        pieces = []
        for r, m in enumerate(metas):
            if int(m[0]) == 0:
                continue
            if r == 0:
                pieces.append(local)
                continue
            buf = torch.empty(tuple(int(v) for v in m), dtype=torch.bfloat16, device=device)
            _recv(buf, peer(r), group)
            pieces.append(buf)
        return torch.cat(pieces, dim=0)
    shape = [torch.zeros(4, dtype=torch.int64, device=device) for _ in range(world)]
    mine = torch.tensor(list(local.shape) if local is not None else [0, 0, 0, 0], dtype=torch.int64, device=device)
    _all_gather(shape, mine, group)
    tail = next(tuple(int(v) for v in s[1:]) for s in shape if int(s[0]) > 0)
    tmax = max(int(s[0]) for s in shape)
    pad = torch.zeros((tmax,) + tail[:-1] + (tail[-1] * 2,), dtype=torch.uint8, device=device)   # bf16 as bytes on the wire
    if local is not None:
        pad[: local.shape[0]] = local.contiguous().view(torch.uint8)
    bufs = [torch.empty_like(pad) for _ in range(world)]
    _all_gather(bufs, pad, group)
    return torch.cat([b[: int(s[0])] for b, s in zip(bufs, shape)], dim=0).view(torch.bfloat16)


def _bcast_from_first(t, group):
    """Every rank ends up with rank 0's tensor (bf16 moves as bytes: gloo has no bf16 wire type)."""
    src = dist.get_global_rank(group, 0) if group else 0
    wire = t.contiguous()
    if dist.get_rank(group) != 0 and wire.data_ptr() == t.data_ptr():
        wire = wire.clone()                     # the caller's tensor (e.g. its posterior_noise) is not overwritten with rank 0's
    _broadcast(wire, src, group)
    return wire


class _SharedPosterior(DiagonalGaussianDistribution):
    """Posterior of the sharded encode.  Every rank patchifies only ITS rows of the sampled latent, so all ranks must
    hold the same sample: a draw from each rank's own device RNG (what ``sample()`` does by default, like diffusers) would
    silently mix shards of different samples.  The noise - injected, drawn from ``generator`` or from the global RNG - is
    therefore always rank 0's, broadcast over the group."""

    def __init__(self, moments_cl, latent_channels, dtype, group):
        super().__init__(moments_cl, latent_channels, dtype)
        self._group = group

    def sample(self, generator=None, noise=None):
        T, h, w, _ = self._m[0].shape
        if noise is None:
            noise = torch.randn(len(self._m), self._L, T, h, w, generator=generator, device=self._m[0].device, dtype=self._dtype)
        noise = _bcast_from_first(noise.to(self._m[0].device), self._group)
        return super().sample(noise=noise)


class _SharedScheduler:
    """``pipe.scheduler`` for the sharded path: the `--noise_step` pre-noising draws eps with torch.randn_like on every
    rank (ref :449-457); rank 0's draw is the one all ranks use."""

    def __init__(self, sched, group):
        self._sched, self._group, self.config = sched, group, sched.config

    def get_velocity(self, *a, **k):
        return self._sched.get_velocity(*a, **k)

    def add_noise(self, original_samples, noise, timesteps):
        return self._sched.add_noise(original_samples, _bcast_from_first(noise, self._group), timesteps)


@torch.no_grad()
def encode_sharded(vae, x, group=None):
    """vae.encode with frame-batches sharded over ranks + temporal halo exchange.  Every rank returns the full
    posterior (moments all-gathered), bit-identical to ``vae.encode(x)``."""
    from . import ops
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    assert x.shape[0] == 1, "sharded path runs B = 1 (the reference's batch)"
    x = x.to(vae.device).contiguous()
    if getattr(vae, "_conv_in_split", False):                   # same form of conv_in as the single-GPU encode (bit-identity)
        x_cl = ops.cl_im2col3x3_from_ncthw(x[0], vae.pc["encoder.conv_in.taps"].cin_pad)
        enc = lambda xb, cache: vae._encoder(xb, cache, split_in=True)      # noqa: E731
    else:
        x_cl = ops.cl_from_ncthw(x[0], vae.pc["encoder.conv_in"].cin_pad)
        enc = vae._encoder
    outs, cache = _run_sharded(vae, x_cl, frame_batches(x_cl.shape[0], vae.enc_batch), world, rank, group, enc)
    moments = _gather_time(outs, group, world, vae.device)
    vae.last_halo_bytes = vae.last_halo_bytes_encode = cache.bytes_sent
    vae.last_halo_stats_encode = cache.stats()
    return _SharedPosterior([moments], vae.lat, vae.dtype, group)


@torch.no_grad()
def decode_sharded(vae, z, group=None, _range01=False, _prescale=1.0, gather="all"):
    """vae.decode with latent frame-batches sharded over ranks.  ``gather``: "all" - every rank returns the full [1,3,F,H,W]
    video; "writer" - rank 0 does, the others return None; "none" - every rank returns only ITS frames [1,3,F_r,H,W]
    (no data-path collective at all: each rank writes its own frames, like the chunk farm)."""
    from . import ops
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    assert z.shape[0] == 1
    z = z.to(vae.device).contiguous()
    z_cl = ops.cl_from_ncthw(z[0], vae.pc["decoder.conv_in"].cin_pad, scale=_prescale)
    split = getattr(vae, "_conv_out_split", False)              # the single-GPU decode runs conv_out tap-split: same arithmetic here
    dec = (lambda zb, cache: vae._decoder(zb, cache, split_out=True)) if split else vae._decoder
    outs, cache = _run_sharded(vae, z_cl, frame_batches(z_cl.shape[0], vae.dec_batch), world, rank, group, dec, "dec")
    if split:      # partial planes -> conv_out's bf16 output, channels-last [T,H,W,4] (the range map comes after the gather)
        cc = vae.config["out_channels"]
        outs = [torch.nn.functional.pad(ops.conv_out_gather(o, cc, vae.conv_out_bias, torch.bfloat16).permute(1, 2, 3, 0), (0, 4 - cc)).contiguous()
                for o in outs]
    post = dict(scale=0.5, shift=0.5, lo=0.0, hi=1.0) if _range01 else {}
    vae.last_halo_bytes = vae.last_halo_bytes_decode = cache.bytes_sent
    vae.last_halo_stats_decode = cache.stats()
    if gather == "none":
        if not outs:
            return None
        local = torch.cat(outs, dim=0) if len(outs) > 1 else outs[0]
        return ops.ncthw_from_cl(local.contiguous(), vae.config["out_channels"], vae.dtype, **post)[None]
    full = _gather_time(outs, group, world, vae.device, to=gather)          # [F,H,W,4] channels-last
    if full is None:
        return None
    return ops.ncthw_from_cl(full, vae.config["out_channels"], vae.dtype, **post)[None]


# ---- (C) sequence / head parallel DiT -------------------------------------------------------------------
def _row_bounds(n, world):
    return [(i * n) // world for i in range(world + 1)]


@torch.no_grad()
def dit_forward_ulysses(tr, hidden, text, t, rope, group=None):
    """One sample of CogVideoXTransformer3DModel.forward (same arguments as ``tr._forward_one``) with rows sharded over the
    ranks of ``group`` and attention head-parallel.  Every rank returns the full [T,C,H,W] prediction."""
    import math

    from . import ops
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    D, heads = tr.D, tr.heads
    if getattr(tr, "linear_precision", "bf16") != "bf16" or getattr(tr, "attention_precision", "bf16") != "bf16":
        raise NotImplementedError("the sequence/head-parallel DiT runs the bf16 operators only (the MXFP8 variant of BASELINE "
                                  "configs[4] is single-GPU / one clip per GPU)")
    if heads % world:
        raise ValueError(f"{heads} attention heads do not split over {world} ranks")
    hloc = heads // world
    dev = tr.device
    T, Cc, H, W = hidden.shape
    p, pt = tr.p, tr.pt
    hidden = hidden.to(dev).contiguous()
    text = text.to(dev, torch.bfloat16).contiguous()
    cos, sin = (r.to(dev, torch.float32).contiguous() for r in rope)
    Lt = text.shape[0]
    nv = (T // pt) * (H // p) * (W // p)
    N = Lt + nv
    npad = (N + 127) // 128 * 128
    bounds = _row_bounds(N, world)
    r0, r1 = bounds[rank], bounds[rank + 1]
    counts = [bounds[i + 1] - bounds[i] for i in range(world)]
    nloc = r1 - r0
    assert nloc > 0, "more ranks than tokens"
    lt_loc = max(0, min(r1, Lt) - r0)                          # local rows that are text rows (they come first)
    v0, v1 = max(r0, Lt) - Lt, max(r1, Lt) - Lt                # local video rows [v0, v1)
    blocks_mod, final_mod = tr._modulation(t)

    hs = torch.empty(nloc, D, dtype=torch.bfloat16, device=dev)
    if lt_loc:
        ops.linear(text[r0:r0 + lt_loc].contiguous(), tr.pe_text, out=hs[:lt_loc])
    if v1 > v0:
        tok = ops.patchify(hidden, pt, p, tr.pe_proj.cin_pad)
        ops.linear(tok[v0:v1].contiguous(), tr.pe_proj, out=hs[lt_loc:])
    cos_l, sin_l = cos[v0:v1].contiguous(), sin[v0:v1].contiguous()
    if v1 == v0:                                               # a text-only shard still hands a valid table to the kernel
        cos_l, sin_l = cos[:1].contiguous(), sin[:1].contiguous()
    # Buffers are allocated ONCE per call and reused by all layers.  Rank-local head-major operands are packed with row
    # stride nloc (no pad rows), so "my rows of rank j's heads" is one contiguous chunk and Ql / Kl / Vl ARE the all-to-all
    # send buffers; the receive buffers are persistent too and the only per-layer data movement besides the collective is
    # the re-interleave of the received [source rank][head][rows] blocks into the kernel's [head][all rows] layout (one launch).
    z = lambda *shp: torch.zeros(*shp, dtype=torch.bfloat16, device=dev)   # noqa: E731
    e = lambda *shp: torch.empty(*shp, dtype=torch.bfloat16, device=dev)   # noqa: E731
    # one extra row per head (column for V^T) behind the nloc rows: the K row carries this rank's score-bound pair of the head to the rank
    # that owns the head (dove_ulysses_place_bf16), so the bound needs no collective of its own
    Ql, Kl, Vl = z(heads, nloc + 1, 64), z(heads, nloc + 1, 64), z(heads, 64, nloc + 1)
    Qh, Kh, Vt = z(hloc, npad, 64), z(hloc, npad, 64), z(hloc, 64, npad)     # pad rows / columns stay zero across layers
    qscale = (tr.hd ** -0.5) * math.log2(math.e)
    blk_in = [(c + 1) * hloc * 64 for c in counts]              # elements I receive from each source rank (rows + the extra row)
    blk_out = [(nloc + 1) * hloc * 64] * world
    ret_in, ret_out = [c * hloc * 64 for c in counts], [nloc * hloc * 64] * world     # the attention output's way back: rows only
    rq, rk, rv = e((N + world) * hloc * 64), e((N + world) * hloc * 64), e((N + world) * hloc * 64)
    att = e(N, hloc * 64)
    back = e(nloc * hloc * 64 * world)
    att_loc = e(nloc, D)
    # per-head score bound of the attention kernel (max squared q / k row norms): every rank takes the maximum over ITS rows of all heads;
    # the element-wise maximum over the ranks is what one GPU computes over all rows (bit for bit).  The pairs travel in the K blocks of
    # the all-to-all and the receive side reduces them for its heads
    norm2 = torch.empty(heads, 2, dtype=torch.float32, device=dev)
    norm2_mine = torch.empty(hloc, 2, dtype=torch.float32, device=dev)
    k_extra = Kl[:, nloc, :4]                                   # [heads, 4] bf16 slots = two floats per head

    def a2a(out, inp, out_splits, in_splits):
        _all_to_all(out, inp, out_splits, in_splits, group)

    for blk, md in zip(tr.blocks, blocks_mod):
        n1 = ops.layernorm_modulate(hs, blk["ln1"][0], blk["ln1"][1], tr.eps, md["m1"], lt_loc)
        qkv = ops.linear(n1, blk["qkv"])
        ops.qkv_post(qkv, nloc, nloc + 1, heads, lt_loc, blk["nq"][0], blk["nq"][1], blk["nk"][0], blk["nk"][1], cos_l, sin_l,
                     qscale, 1e-6, Ql, Kl, Vl, v_order=0, norm2=norm2)   # natural key order: the pieces are assembled below
        k_extra.view(torch.float32).copy_(norm2)
        a2a(rq, Ql.view(-1), blk_in, blk_out)
        a2a(rk, Kl.view(-1), blk_in, blk_out)
        a2a(rv, Vl.view(-1), blk_in, blk_out)
        # [source rank][hloc][rows of that rank][64] (V^T: [hloc][64][rows]) -> the kernel's [hloc][all rows][64] / quad-swapped
        # [hloc][64][all rows] with zero pad columns: ONE launch (was 3 x world slice copies + a pad clear + an in-place swap)
        ops.ulysses_place(rq, rk, rv, counts, hloc, N, npad, Qh, Kh, Vt, norm2_out=norm2_mine)
        ops.attention(Qh, Kh, Vt, N, npad, hloc, att, norm2=norm2_mine)
        # heads -> rows: rank j gets rows [bounds[j], bounds[j+1]) of my heads; I get my rows of every head group
        a2a(back, att.view(-1), ret_out, ret_in)
        # [source rank = head group][my rows][hloc*64] -> [my rows][all heads]: one strided copy
        att_loc.view(nloc, world, hloc * 64).copy_(back.view(world, nloc, hloc * 64).permute(1, 0, 2))
        ops.linear(att_loc, blk["out"], resid=hs, gate=md["gate1"], gate_split=lt_loc, out=hs)
        n2 = ops.layernorm_modulate(hs, blk["ln2"][0], blk["ln2"][1], tr.eps, md["m2"], lt_loc, out=n1)
        f1 = ops.linear(n2, blk["ff1"], act=1)
        ops.linear(f1, blk["ff2"], resid=hs, gate=md["gate2"], gate_split=lt_loc, out=hs)

    width = tr.proj_out.cout_store
    o_loc = torch.zeros(0, width, dtype=torch.bfloat16, device=dev)
    if v1 > v0:
        xv = hs[lt_loc:].contiguous()
        xv = ops.layernorm_modulate(xv, tr.norm_final[0], tr.norm_final[1], tr.eps)
        xv = ops.layernorm_modulate(xv, tr.norm_out[0], tr.norm_out[1], tr.eps, final_mod, 0)
        o_loc = ops.linear(xv, tr.proj_out)
    vcounts = [max(bounds[i + 1], Lt) - max(bounds[i], Lt) for i in range(world)]
    vmax = max(vcounts)
    pad = torch.zeros(vmax, width, dtype=torch.bfloat16, device=dev)
    pad[: o_loc.shape[0]] = o_loc
    bufs = [torch.empty(vmax, width * 2, dtype=torch.uint8, device=dev) for _ in range(world)]
    _all_gather(bufs, pad.view(torch.uint8), group)
    o = torch.cat([b[:c] for b, c in zip(bufs, vcounts)], dim=0).view(torch.bfloat16)
    return ops.unpatchify(o.contiguous(), T, Cc, H, W, pt, p, tr.dtype)


class _ShardedVAE:
    def __init__(self, vae, group):
        self._vae, self._group = vae, group
        self.device, self.dtype, self.config = vae.device, vae.dtype, vae.config

    def encode(self, x):
        class _O:
            latent_dist = encode_sharded(self._vae, x, self._group)
        return _O()


class _ShardedTransformer:
    def __init__(self, tr, group):
        self._tr, self._group, self.config = tr, group, tr.config

    def __call__(self, hidden_states, encoder_hidden_states, timestep, image_rotary_emb=None, return_dict=False, **_):
        ts = [int(v) for v in timestep.reshape(-1).tolist()]
        outs = [dit_forward_ulysses(self._tr, hidden_states[b], encoder_hidden_states[b], ts[b if len(ts) > 1 else 0],
                                    image_rotary_emb, self._group) for b in range(hidden_states.shape[0])]
        return (torch.stack(outs),)


class _ShardedPipe:
    """Duck-typed ``pipe`` whose VAE and transformer run sharded over ``group`` (what process_video touches, nothing more)."""

    def __init__(self, pipe, group, gather="all"):
        self._pipe, self._group, self._gather = pipe, group, gather
        self.vae = _ShardedVAE(pipe.vae, group)
        self.transformer = _ShardedTransformer(pipe.transformer, group)
        self.scheduler = _SharedScheduler(pipe.scheduler, group)
        self.tokenizer, self.text_encoder = pipe.tokenizer, pipe.text_encoder

    def decode_latents(self, latents, _range01=False):
        z = latents.permute(0, 2, 1, 3, 4).contiguous()
        return decode_sharded(self._pipe.vae, z, self._group, _range01=_range01,
                              _prescale=1.0 / float(self._pipe.vae.config["scaling_factor"]), gather=self._gather)


@torch.no_grad()
def process_video_sharded(pipe, video, *, group=None, gather="all", **kw):
    """``process_video`` on ONE clip with every stage sharded over the ranks of ``group``: halo-exact VAE (B) and
    sequence/head-parallel DiT (C).  All ranks must pass the same clip; random draws (posterior sample, optional
    pre-noising) are rank 0's, broadcast, so the ranks stay consistent whatever their RNG states.  All ranks return the
    full SR clip (``gather="all"``), only rank 0 does (``"writer"``), or every rank keeps just the frames it decoded
    (``"none"``) - bit-identical to the single-GPU result given rank 0's noise.  With ONE rank there is nothing to shard: the call is
    ``process_video`` itself (the sharding machinery - head-major repacking for the all-to-all, the gather of the decoded frames - used to
    cost 2.4 % before any wire existed)."""
    if dist.get_world_size(group) == 1:
        return process_video(pipe, video, **kw)
    return process_video(_ShardedPipe(pipe, group, gather), video, **kw)
