"""Data-parallel training: one process per GPU, torch.distributed over RCCL/xGMI (backend "nccl").

The reference wraps the model in apex DistributedDataParallel (bin/vc_train.py:423-431): an averaged gradient all-reduce
per optimiser step, overlapped with the backward pass through per-bucket hooks, plus a parameter broadcast at wrap time;
BatchNorm statistics stay rank-local and rank 0's buffers are what a checkpoint holds (trainers/base.py:98-101).

Here the gradients already live in ONE flat fp32 buffer (optim.FlatAdam) that the weight-gradient kernels write into
directly -- autograd never sees them, so there are no per-parameter hooks to hang a bucket on.  Instead the model names a
few *gradient cuts* (ops.functional.cut_point) and a stage plan (`model.dp_plan()`): the backward pass runs stage by
stage, every stage finishes the gradients of one contiguous range of the flat buffer, and that range's all-reduce is
started (asynchronously, on RCCL's stream) before the next stage is launched:

    VTN / TTS   [decoder + heads + postnet] -> [encoder layers + after-norm] -> [input layer]
                3 buckets: 62.8 | 42.6 | 16.5 MB fp32 (VTN vc1) -- only the last one travels with nothing to hide behind
    AAS-VC      [all decoder layers + heads + postnet, and -- rooted on the auxiliary stream -- aligner + duration
                predictor] -> [encoder]                                      2 buckets: 573 | 57 MB fp32 (vc2)
                (models/*.dp_plan() is the authority; these are the shipped recipes' numbers)

xGMI is point-to-point (7 links x ~153 GB/s per GPU), so large messages are what reaches link bandwidth: a bucket is one
contiguous slice (>= 28 MB), split into 128 MiB collectives only above that.  `payload="bf16"` halves the bytes on the links:
the slice is cast into a bf16 staging buffer, summed over the ranks in bf16 and cast back (the fp32 gradients of a rank are
rounded once; the sum of 8 ranks then carries bf16 rounding, ~3 significant digits, which Adam's normalisation tolerates;
fp32 is the parity setting -- the reference's DDP all-reduces fp32 -- and the default of EVERY trainer; `dp_grad_payload: bf16` is a
per-recipe opt-in, worth it for AAS-VC: 630 MB fp32 per step is 7 ms of ring time on one 153 GB/s link against a 12 ms step).  The 1/world of the mean rides on the loss, so the collectives
are plain sums.  `collective="rs_ag"` runs every bucket as reduce-scatter + all-gather (each rank sums 1/world of the slice, then
the shards are gathered) instead of one all-reduce -- the same bytes per link for a ring, but two half-size collectives that
RCCL can place on different channels, and the shape a sharded optimiser step would need; same results (a sum is a sum).
"""
import os

import torch

from ..ops import functional as Fn


def init_from_env(backend=None):
    """env:// initialisation used by the launcher (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return None, 0, 1
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if not dist.is_initialized():
        dist.init_process_group(backend)
    return dist, dist.get_rank(), dist.get_world_size()


def allreduce_mean_(flat, dist, world, chunk_numel=32 * 1024 * 1024, group=None, force=False, stage_bf16=None):
    """In-place mean all-reduce of a flat buffer in large chunks (works on cuda/RCCL and cpu/gloo).
    force: issue the collectives even at world size 1 (exercises the backend on a single device).
    stage_bf16: a bf16 buffer of the same size -> the exchange runs on a bf16 copy (half the bytes on the links)."""
    if world <= 1 and not force:
        return flat
    buf = flat
    if stage_bf16 is not None:
        from ..ops import kernels as K
        K.cast(flat, torch.bfloat16, out=stage_bf16)
        buf = stage_bf16
    n = buf.numel()
    handles = []
    for o in range(0, n, chunk_numel):
        handles.append(dist.all_reduce(buf[o:o + chunk_numel], op=dist.ReduceOp.SUM, group=group, async_op=True))
    for h in handles:
        h.wait()
    if stage_bf16 is not None:
        from ..ops import kernels as K
        K.cast(stage_bf16, torch.float32, out=flat)
    flat.mul_(1.0 / world)
    return flat


def allreduce_sum_begin(flat, dist, world, chunk_numel=32 * 1024 * 1024, group=None, force=False):
    """Start the in-place SUM all-reduce of a flat buffer (or a slice of one) and return the pending handles: the
    collectives run on the backend's own stream while the caller keeps launching compute; allreduce_end() joins."""
    if world <= 1 and not force:
        return []
    return [dist.all_reduce(flat[o:o + chunk_numel], op=dist.ReduceOp.SUM, group=group, async_op=True)
            for o in range(0, flat.numel(), chunk_numel)]


def allreduce_end(handles):
    for h in handles:
        h.wait()


class _RsAg:
    """One bucket as reduce-scatter + all-gather, in chunks of at most `chunk_numel` elements: begin (the constructor) starts the
    reduce-scatter of every (padded) chunk into this rank's shard and, chained behind it on the backend's stream, the ASYNCHRONOUS
    all-gather of the shards -- both halves overlap with the backward stages that follow; finish() only waits and copies a padded
    tail back.  Backends without reduce_scatter_tensor (gloo -- chosen from dist.get_backend(), an asynchronous failure could not be
    caught) run an all-reduce of the chunk, from which every rank keeps its shard: the data path the test suite runs on CPU."""

    def __init__(self, buf, dist, world, group=None, chunk_numel=None):
        self.buf, self.dist, self.world, self.group = buf, dist, world, group
        self.emulated = str(dist.get_backend(group)).lower() not in ("nccl", "rccl")
        n = buf.numel()
        chunk = n if not chunk_numel else max(world, (int(chunk_numel) // world) * world)
        self.parts = []
        for o in range(0, n, chunk):
            piece = buf[o:o + chunk]
            m = piece.numel()
            shard_n = (m + world - 1) // world
            padded = None
            src = piece
            if shard_n * world != m:
                padded = torch.zeros(shard_n * world, dtype=buf.dtype, device=buf.device)
                padded[:m].copy_(piece)
                src = padded
            shard = torch.empty(shard_n, dtype=buf.dtype, device=buf.device)
            if self.emulated:
                h = dist.all_reduce(src, op=dist.ReduceOp.SUM, group=group, async_op=True)
                self.parts.append([piece, padded, src, shard, shard_n, h, None])
            else:
                h = dist.reduce_scatter_tensor(shard, src, op=dist.ReduceOp.SUM, group=group, async_op=True)
                # same process group -> same communication stream: the gather is ordered behind the scatter without a host wait
                g = dist.all_gather_into_tensor(src, shard, group=group, async_op=True)
                self.parts.append([piece, padded, src, shard, shard_n, h, g])

    def finish(self):
        for part in self.parts:
            piece, padded, src, shard, shard_n, h, g = part
            h.wait()
            if self.emulated:
                r = self.dist.get_rank(self.group)
                shard.copy_(src[r * shard_n:(r + 1) * shard_n])
                g = self.dist.all_gather_into_tensor(src, shard, group=self.group, async_op=True)
            g.wait()
            if padded is not None:
                piece.copy_(padded[: piece.numel()])
        self.parts = []


class _Waiter:
    def __init__(self, h):
        self.h = h

    def finish(self):
        self.h.wait()


def broadcast_(flat, dist, world, src=0, group=None):
    """Initial parameter broadcast from rank 0 (what DDP does at wrap time)."""
    if world > 1:
        dist.broadcast(flat, src=src, group=group)
    return flat


def broadcast_model_(model, optimizer, dist, world, src=0, group=None):
    """Rank `src`'s parameters and buffers to everyone (apex DDP's wrap-time broadcast, bin/vc_train.py:431): the flat
    parameter buffer in one collective (optim.FlatAdam) or tensor by tensor (torch optimisers), then the buffers
    (BatchNorm running statistics).  Afterwards the bf16 shadows are refreshed."""
    if world <= 1:
        return
    flat = getattr(optimizer, "flat_p", None)
    if flat is not None:
        dist.broadcast(flat, src=src, group=group)
        listed = {id(p) for p in optimizer.params}
        rest = [p for p in model.parameters() if id(p) not in listed]       # frozen parameters live outside the flat buffer
    else:
        rest = list(model.parameters())
    for t in rest + list(model.buffers()):
        dist.broadcast(t.data, src=src, group=group)
    if hasattr(optimizer, "refresh_shadow"):
        optimizer.refresh_shadow()


class OverlappedBackward:
    """The backward pass of one data-parallel step, stage by stage, with the gradient exchange of every finished stage in
    flight while the next one runs (see the module docstring).

        ob = OverlappedBackward(model, optimizer, dist, world)
        with ob.forward_context():                 # activates the model's gradient cuts
            out = model(...); losses = {"loss": l1 + bce}          # keys as named by model.dp_plan()
        ob.backward(losses)                        # stages + asynchronous all-reduces, joined before it returns
        optimizer.step()

    For hipGraph replay the pieces are exposed separately: `run_stage(i, losses)` (capturable: launches kernels only),
    `begin_reduce(i)` (RCCL calls, issued between graph replays) and `finish()`.
    """

    def __init__(self, model, optimizer, dist, world, payload="fp32", chunk_numel=32 * 1024 * 1024, group=None, force=False,
                 plan=None, collective="allreduce"):
        if not hasattr(optimizer, "flat_g"):
            raise TypeError("OverlappedBackward needs optim.FlatAdam (gradients in one flat buffer)")
        if payload not in ("fp32", "bf16"):
            raise ValueError("payload must be 'fp32' or 'bf16'")
        if collective not in ("allreduce", "rs_ag"):
            raise ValueError("collective must be 'allreduce' or 'rs_ag'")
        self.collective = collective
        self.model, self.opt, self.dist, self.world, self.group = model, optimizer, dist, max(1, int(world)), group
        self.force, self.chunk, self.payload = force, chunk_numel, payload
        self.plan = plan if plan is not None else model.dp_plan()
        roots = [r for st in self.plan for r in ([st["root"]] if isinstance(st["root"], str) else st["root"])]
        roots += [st["branch_root"] for st in self.plan if st.get("branch_root")]
        self.cuts = Fn.GradCuts([r[4:] for r in roots if r.startswith("cut:")])
        self.ranges = [optimizer.param_ranges(st["modules"]) for st in self.plan]
        covered = sorted(r for rs in self.ranges for r in rs)
        pos = 0
        for lo, hi in covered:
            if lo != pos:
                break
            pos = hi
        if pos != optimizer.numel:
            raise RuntimeError("the model's dp_plan() does not cover every trainable parameter exactly once "
                               f"(covered up to {pos} of {optimizer.numel} elements)")
        self.scale = 1.0 / self.world
        self.stage_buf = torch.empty(optimizer.numel, dtype=torch.bfloat16, device=optimizer.flat_g.device) if payload == "bf16" else None
        self.handles, self.pending_bf16 = [], []
        self.marks = {}             # stage -> ops.kernels.GraphMark recorded behind the stage (mark())
        self.comm_stream = None     # the stream the exchanges are issued from when they wait for marks

    # -- pieces ---------------------------------------------------------------------------------------------
    def forward_context(self):
        return Fn.grad_cuts(self.cuts)

    def active(self):
        return self.world > 1 or self.force

    def run_stage(self, i, losses, scale=None):
        """Backward of stage i: from a loss (`loss:<key>`) or from below a cut (`cut:<name>`) -- or from several such roots, in
        the order given -- then the join of the parameter-gradient side work, so that the stage's slices of the flat gradient
        buffer are final on the current stream."""
        if i == 0 and hasattr(self.opt, "join_prologue"):
            self.opt.join_prologue()                             # zero-filled gradients + refreshed weight copies (FlatAdam.begin_step)
        return self._run_stage(i, losses, scale)

    def _run_stage(self, i, losses, scale=None):
        roots = self.plan[i]["root"]
        branch_root = self.plan[i].get("branch_root")           # a loss (or a cut) whose sub-network ran on the auxiliary stream
        fork = None
        if branch_root is not None and torch.cuda.is_available():
            fork = torch.cuda.Event()
            fork.record()                                        # before this stage queues anything: the branch starts here
        s = self.scale if scale is None else scale
        if branch_root is not None and branch_root.startswith("cut:"):      # the rest of a branch below a cut of an earlier stage
            if fork is not None:
                Fn.branch_resume(self.cuts, branch_root[4:], fork)
            else:
                self.cuts.resume(branch_root[4:])
        elif branch_root is not None:                            # first: the calling stream is still empty
            loss = losses.get(branch_root[5:])
            if loss is not None and loss.requires_grad:
                if fork is not None:
                    Fn.branch_backward(loss, fork, retain_graph=False, scale=s)
                else:
                    Fn.root_backward(loss, s, retain_graph=False)
        for root in ([roots] if isinstance(roots, str) else list(roots)):
            if root.startswith("loss:"):
                loss = losses.get(root[5:])
                if loss is not None and loss.requires_grad:
                    Fn.root_backward(loss, s, retain_graph=False)
            else:
                self.cuts.resume(root[4:])
        Fn.side_join()
        if fork is not None:
            Fn.branch_wait()

    def mark(self, i):
        """Record "stage i's gradients are final" on the current stream.  Inside a capture this is an event-record NODE of the
        graph (ops.kernels.GraphMark): the stages of a backward pass can then share ONE graph and the exchange of stage i still
        starts when stage i is done, while the same graph runs stage i + 1 (`begin_reduce(i, after_mark=True)` after the replay)
        -- the granularity of the exchange is no longer paid for in graph boundaries (apex DDP's per-bucket
        hooks, bin/vc_train.py:423-431, are what this stands in for)."""
        if not self.active() or not self.opt.flat_g.is_cuda:
            return
        from ..ops import kernels as K
        m = self.marks.get(i)
        if m is None:
            m = self.marks[i] = K.GraphMark()
        m.record()

    def begin_reduce(self, i, after_mark=False):
        """after_mark: issue the exchange from the communication stream, behind mark i (the current stream may already hold the
        rest of the step: a graph that contains later stages); finish() joins as usual."""
        if not self.active():
            return
        if after_mark and i in self.marks:
            if self.comm_stream is None:
                self.comm_stream = Fn.distinct_stream()
            self.marks[i].wait(self.comm_stream)
            with torch.cuda.stream(self.comm_stream):
                self._begin_reduce(i)
            return
        self._begin_reduce(i)

    def _begin_reduce(self, i):
        from ..ops import kernels as K
        for lo, hi in self.ranges[i]:
            if self.stage_buf is not None:
                K.cast(self.opt.flat_g[lo:hi], torch.bfloat16, out=self.stage_buf[lo:hi])
                buf = self.stage_buf[lo:hi]
                self.pending_bf16.append((lo, hi))
            else:
                buf = self.opt.flat_g[lo:hi]
            if self.collective == "rs_ag":
                self.handles.append(_RsAg(buf, self.dist, self.world, self.group, self.chunk))
            else:
                self.handles += [_Waiter(h) for h in allreduce_sum_begin(buf, self.dist, self.world, self.chunk, self.group, force=True)]

    def finish(self):
        for h in self.handles:
            h.finish()
        self.handles = []
        if self.pending_bf16:
            from ..ops import kernels as K
            for lo, hi in self.pending_bf16:
                K.cast(self.stage_buf[lo:hi], torch.float32, out=self.opt.flat_g[lo:hi])
            self.pending_bf16 = []

    # -- the whole thing (eager; the trainers) -----------------------------------------------------------------
    def backward(self, losses, reduce=True, scale=None):
        """reduce=False: a gradient-accumulation micro-step (no exchange; the buffer keeps accumulating).
        scale: factor on the loss (default 1/world; with gradient accumulation 1/(world * accumulate steps))."""
        for i in range(len(self.plan)):
            self.run_stage(i, losses, scale)
            if reduce:
                self.begin_reduce(i)
        if reduce:
            self.finish()
        self.cuts.clear()

    def bucket_bytes(self):
        e = 2 if self.payload == "bf16" else 4
        return [sum(hi - lo for lo, hi in rs) * e for rs in self.ranges]


class FlushExchange:
    """The gradient exchange at the granularity of GRADIENT-BATCH FLUSHES of an UNCUT backward pass (round 6).

    OverlappedBackward cuts the backward pass into stages; every cut is a join of the parameter-gradient work (and of the branch on
    the auxiliary stream), and those joins -- not the graph boundaries: one graph with event-record nodes costs the same as one
    graph per stage -- are what a finer plan pays for (AAS-VC with a stage per decoder layer: 12.7 vs 10.8 ms).  Here nothing is
    cut.  The parameter-gradient closures of a backward pass are flushed in batches anyway (ops.functional._side_run); behind every
    flush a mark is recorded on the flushing stream (ops.kernels.GraphMark: an event-record node when the step is captured), and the
    slices of the flat gradient buffer that are FINAL at a flush -- learned once, from an instrumented eager pass that snapshots the
    buffer behind every flush -- are all-reduced from the communication stream behind that mark, while the same graph keeps running
    the rest of the backward pass.  What apex DDP's per-bucket hooks (bin/vc_train.py:423-431) do, at the batch granularity the
    schedule already has, with no new synchronisation in the backward pass.

        fx = FlushExchange(optimizer, dist, world)
        fx.learn(run)                  # run(): zero-fill + forward + backward + side_join, eager, on the REAL shapes
        ... capture or run `run()` with fx.recording(): marks behind every flush, fx.mark_end() behind the join ...
        fx.issue(); fx.finish()        # after the graph's launch (or the eager pass): exchanges behind their marks, then the join

    The plan (which ranges travel behind which flush) is the same on every rank -- a parameter counts as final behind the latest
    flush that changed it on ANY rank (a MAX all-reduce of the learned table) -- so every rank issues the same collectives in the
    same order whatever its data.  A change of the schedule (batch size, streams, model) needs a new learn()."""

    def __init__(self, optimizer, dist, world, payload="fp32", chunk_numel=32 * 1024 * 1024, group=None, force=False,
                 min_bucket_numel=4 * 1024 * 1024):
        if not hasattr(optimizer, "flat_g"):
            raise TypeError("FlushExchange needs optim.FlatAdam (gradients in one flat buffer)")
        if payload not in ("fp32", "bf16"):
            raise ValueError("payload must be 'fp32' or 'bf16'")
        self.opt, self.dist, self.world, self.group = optimizer, dist, max(1, int(world)), group
        self.payload, self.chunk, self.force, self.min_bucket = payload, chunk_numel, force, int(min_bucket_numel)
        self.scale = 1.0 / self.world
        self.stage_buf = torch.empty(optimizer.numel, dtype=torch.bfloat16, device=optimizer.flat_g.device) if payload == "bf16" else None
        self.plan = None            # [(flush ordinal or None = behind the join, [(lo, hi), ...])]
        self.plans, self.key = {}, None
        self.marks, self.mark_streams, self.end_mark = [], [], None
        self.n = 0                  # flush ordinal inside the current pass
        self.mode = None            # "learn" | "mark"
        self.snap, self.last_change = None, None
        self.comm_stream = None
        self.handles, self.pending_bf16 = [], []

    def active(self):
        return self.world > 1 or self.force

    # -- the hook behind every flush -------------------------------------------------------------------------------------------
    def _on_flush(self):
        if self.mode == "learn":
            torch.cuda.synchronize()
            cur = self.opt.flat_g
            changed = (cur != self.snap)
            csum = torch.cumsum(changed.to(torch.int32), 0)
            hi = csum[self.p_hi - 1]
            lo = torch.where(self.p_lo > 0, csum[(self.p_lo - 1).clamp_min(0)], torch.zeros_like(hi))
            touched = (hi - lo) > 0
            self.last_change[touched] = self.n
            self.snap = cur.clone()
        elif self.mode == "mark":
            from ..ops import kernels as K
            if self.n >= len(self.marks):
                self.marks.append(K.GraphMark())
                self.mark_streams.append(None)
            self.marks[self.n].record()
            self.mark_streams[self.n] = torch.cuda.current_stream().cuda_stream
        self.n += 1

    def recording(self):
        """Context: flushes of the backward pass inside it leave marks."""
        fx = self

        class _Ctx:
            def __enter__(self_):
                fx.mode, fx.n = "mark", 0
                Fn._Side.on_flush = fx._on_flush

            def __exit__(self_, *exc):
                Fn._Side.on_flush = None
                fx.mode = None
                if exc[0] is None and fx.plan is not None and fx.n != fx.n_flushes:
                    raise RuntimeError(f"FlushExchange: this pass flushed {fx.n} gradient batches, the learned plan has {fx.n_flushes} "
                                       "(the schedule changed: learn() again)")
        return _Ctx()

    def mark_end(self):
        """Behind the join of the backward pass (ops.functional.side_join) on the current stream: everything is final."""
        if not self.active():
            return
        from ..ops import kernels as K
        if self.end_mark is None:
            self.end_mark = K.GraphMark()
        self.end_mark.record()

    # -- the plan -------------------------------------------------------------------------------------------------------------------
    def learn(self, run):
        """run(): one eager training pass up to and including side_join (zero-filled gradients at its start)."""
        self.learn_begin()
        try:
            run()
        finally:
            Fn._Side.on_flush = None
            self.mode = None
        return self.learn_end()

    def select(self, key):
        """Several schedules in one run (a trainer whose step changes with the training regime: the duration loss switching on):
        one plan per key; -> True if the plan of `key` is known."""
        if key != self.key:
            self.plans[self.key] = (self.plan, getattr(self, "n_flushes", 0))
            self.key = key
            self.plan, self.n_flushes = self.plans.get(key, (None, 0))
        return self.plan is not None

    def learn_begin(self):
        """The backward pass that follows (eager, from zero-filled gradients) is instrumented; learn_end() behind its side_join."""
        opt = self.opt
        order = sorted(zip(opt.offsets, opt.params), key=lambda t: t[0])
        los = [o for o, _ in order]
        his = los[1:] + [opt.numel]                                # up to the next parameter (absorbs the alignment padding)
        dev = opt.flat_g.device
        self.p_lo = torch.tensor(los, dtype=torch.int64, device=dev)
        self.p_hi = torch.tensor(his, dtype=torch.int64, device=dev)
        self.last_change = torch.full((len(los),), -1, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        self.snap = opt.flat_g.clone()          # zeros, or what earlier micro-steps of an accumulation window left
        self.mode, self.n = "learn", 0
        Fn._Side.on_flush = self._on_flush
        self._los, self._his = los, his

    def learn_end(self):
        Fn._Side.on_flush = None
        self.mode = None
        los, his = self._los, self._his
        torch.cuda.synchronize()
        self.n_flushes = self.n
        if self.dist is not None and self.world > 1:
            # ONE plan for every rank (identical collectives in identical order), safe for every rank's data: a parameter is final
            # behind the LATEST flush that changed it on ANY rank (a contribution that happens to be exactly zero on one rank -- dead
            # units, an absent branch -- must not let that rank's view release the bucket early)
            info = torch.cat([self.last_change, torch.tensor([self.n_flushes, -self.n_flushes], dtype=torch.int64, device=self.last_change.device)])
            self.dist.all_reduce(info, op=self.dist.ReduceOp.MAX, group=self.group)
            nmax, nmin = int(info[-2]), -int(info[-1])
            if nmax != self.n_flushes or nmin != self.n_flushes:
                raise RuntimeError(f"FlushExchange: the ranks flush different numbers of gradient batches ({nmin} .. {nmax})")
            self.last_change = info[:-2]
        last = self.last_change.tolist()
        self.snap = self.last_change = None
        # ranges per flush ordinal; parameters no flush touched (or touched behind the last flush) travel behind the join
        by_flush = {}
        for (lo, hi), n in zip(zip(los, his), last):
            by_flush.setdefault(n if n >= 0 else None, []).append((lo, hi))
        plan, held, held_n = [], [], 0
        for n in sorted(k for k in by_flush if k is not None):
            held += by_flush[n]
            held_n += sum(hi - lo for lo, hi in by_flush[n])
            if held_n >= self.min_bucket:
                plan.append((n, self._merge(held)))
                held, held_n = [], 0
        tail = held + by_flush.get(None, [])
        if tail:
            plan.append((None, self._merge(tail)))
        self.plan = plan
        return plan

    @staticmethod
    def _merge(ranges):
        out = []
        for lo, hi in sorted(ranges):
            if out and out[-1][1] == lo:
                out[-1][1] = hi
            else:
                out.append([lo, hi])
        return [tuple(r) for r in out]

    def bucket_bytes(self):
        e = 2 if self.payload == "bf16" else 4
        return [sum(hi - lo for lo, hi in rs) * e for _, rs in (self.plan or [])]

    # -- after the graph's launch (or the eager pass) -----------------------------------------------------------------------------------
    def issue(self):
        if not self.active():
            return
        if self.plan is None:
            raise RuntimeError("FlushExchange.issue() before learn()")
        from ..ops import kernels as K
        if self.comm_stream is None:
            self.comm_stream = Fn.distinct_stream()
        comm = self.comm_stream
        for n, ranges in self.plan:
            if n is None:
                self.end_mark.wait(comm)
            else:
                seen = set()
                for k in range(n, -1, -1):                       # the latest mark at or before flush n of every flushing stream
                    sh = self.mark_streams[k]
                    if sh not in seen:
                        seen.add(sh)
                        self.marks[k].wait(comm)
            with torch.cuda.stream(comm):
                for lo, hi in ranges:
                    if self.stage_buf is not None:
                        K.cast(self.opt.flat_g[lo:hi], torch.bfloat16, out=self.stage_buf[lo:hi])
                        buf = self.stage_buf[lo:hi]
                        self.pending_bf16.append((lo, hi))
                    else:
                        buf = self.opt.flat_g[lo:hi]
                    self.handles += [_Waiter(h) for h in allreduce_sum_begin(buf, self.dist, self.world, self.chunk, self.group, force=True)]

    def finish(self):
        """The current stream waits for every exchange (and, with the bf16 payload, converts the buckets back)."""
        for h in self.handles:
            h.finish()
        self.handles = []
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        if self.pending_bf16:
            from ..ops import kernels as K
            for lo, hi in self.pending_bf16:
                K.cast(self.stage_buf[lo:hi], torch.float32, out=self.opt.flat_g[lo:hi])
            self.pending_bf16 = []


def allreduce_grads_(params, dist, world, group=None):
    """Mean all-reduce of `p.grad` for torch optimisers (no flat buffer): one coalesced collective."""
    if world <= 1:
        return
    # every trainable parameter takes part, with zeros where this rank produced no gradient (a branch that did not run here,
    # e.g. a duration predictor before dp_train_start_steps): the buffer layout must not depend on which gradients exist, or
    # the ranks' collectives mismatch in size.  A has-gradient flag per parameter travels at the end of the same buffer: a
    # parameter NO rank produced a gradient for keeps `grad = None` afterwards, as in a single-GPU run -- torch optimisers skip it
    # (no Adam state, no step count, no weight decay), so world = 1 and world > 1 stay in step when the branch switches on later.
    ps = [p for p in params if p.requires_grad]
    if not ps:
        return
    dev, dt = ps[0].device, ps[0].dtype
    for p in ps[1:]:                 # the buffer takes the WIDEST dtype of the set: fp32 gradients are never rounded through a bf16 /
        dt = torch.promote_types(dt, p.dtype)     # fp16 neighbour's dtype on the way (each parameter gets its own dtype back below)
    has = torch.tensor([0.0 if p.grad is None else 1.0 for p in ps], dtype=dt, device=dev)
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).to(dt) for p in ps] + [has])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    any_grad = (flat[-len(ps):] > 0).tolist()
    flat.mul_(1.0 / world)
    o = 0
    for p, live in zip(ps, any_grad):
        n = p.numel()
        if not live:
            p.grad = None
        elif p.grad is None:
            p.grad = flat[o:o + n].view_as(p).to(p.dtype).clone()
        else:
            p.grad.copy_(flat[o:o + n].view_as(p))
        o += n
