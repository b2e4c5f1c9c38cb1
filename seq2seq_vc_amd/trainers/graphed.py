"""Captured training steps for the trainers (config["hip_graph"]).

The reference has no counterpart (its step is a stream of eager torch calls, trainers/ar_vc.py:59-112, trainers/aas_vc.py:56-164);
here the eager step is bound by the host (VTN vc1: 13.5 ms eager, 4.6 ms replayed), so the trainers can replay the step
from hipGraphs.  What a graph bakes in is kept out of it:

  * shapes -- a batch is copied into static device buffers whose time axes are rounded up to a multiple of
    config["graph_length_quantum"] (default 64 frames); one set of graphs per (regime, batch size, rounded lengths);
  * lengths -- they are DATA of the graph (modules.LensBank): every length vector a kernel reads is a slot of one device buffer
    that is recomputed on the host from the new batch's lengths and shipped with one copy before the replay.  That includes the
    length the reference CROPS the batch to before it computes (its longest utterance: models/vtn.py:208-214, 269-271): the frames
    between it and the padded length are absent for every kernel that mixes along time or over the batch (Lens.ext / crop(),
    include/s2svc_hip.h "absent rows"), so a captured step computes what the reference computes on the cropped batch -- losses,
    gradients, parameters after the optimiser step and BatchNorm buffers against the oracle:
    tests/gpu_model_check.py captured_steps_on_short_batches_vs_oracle;
  * the host bookkeeping of a step (step counters, end of training) is replayed on the host.

First sighting of a key: the step runs eagerly on the static buffers ("traced": same padding, same length handling) -- that
also builds everything lazily built for these shapes.  Second sighting: capture, then replay.  Later: replay.
With data parallelism the backward pass is captured stage by stage (distributed.OverlappedBackward) and the all-reduce of a
finished stage is issued between the replays, as bench.py does; the optimiser step is its own graph behind the join.
config["hip_graph"] = "trace" never captures (the eager reference of the tests for exactly this data path).

Gradient accumulation (trainers/aas_vc.py:141-149 of the reference; AASVCTrainer.GRAPH_ACCUMULATE): a MICRO-step is what gets
captured, and its ROLE is part of the key -- (gradients cleared at its start?, optimiser step at its end?) as
`Trainer._graph_regime()` reports it before the step runs.  An accumulation window of k micro-steps replays the "accumulate"
graph (forward + backward into the flat gradient buffer, no zero-fill, no exchange: staged data-parallel passes run their stages
inside this one graph) k - 1 times and the "last" graph(s) (forward + backward [+ the staged exchange] + clip / Adam / WarmupLR
+ the zero-fill for the next window) once; the host-side counters (`backward_steps`, `steps`) advance by the deltas recorded at
capture time.  Every role keeps its own static buffers and length bank.

The set of captured shapes is BOUNDED: at most config["graph_cache_size"] (default 16) keys keep their static buffers, length
bank and graphs; reaching a new key beyond that evicts the least recently used one (its graphs return their blocks to the shared
pool), and a key is only captured on its config["graph_capture_after"]-th sighting (default 2: the first one runs eagerly and
builds everything that is built lazily for the shape).  With lengths rounded to 64 frames a corpus such as ARCTIC yields a
handful of keys per model; a sampler that buckets utterances by length keeps the hit rate high on corpora with a wide spread.
"""
import gc
import os
import logging
from collections import OrderedDict

import torch

from .. import modules as Mo


def _round_up(n, q):
    return ((int(n) + q - 1) // q) * q


class _Entry:
    def __init__(self):
        self.static, self.lens_src, self.caps = {}, {}, {}
        self.bank = None
        self.graphs = []          # [(graph, stage index or None)]
        self.deltas = None
        self.zero_due_after = None    # optimizer._zero_due as the captured step left it (host state a replay must reproduce)
        self.sightings = 0


class GraphedStep:
    def __init__(self, trainer):
        self.t = trainer
        self.quantum = int(trainer.config.get("graph_length_quantum", 64))
        self.trace_only = trainer.config.get("hip_graph") == "trace"
        self.entries = OrderedDict()                    # key -> _Entry, least recently used first
        self.max_entries = max(1, int(trainer.config.get("graph_cache_size", 16)))
        self.capture_after = max(2, int(trainer.config.get("graph_capture_after", 2)))
        self.evictions = 0
        self.stream = torch.cuda.Stream(trainer.device)      # replaced by a distinct one at capture time if it aliases a stream in use
        # host batches reach the static buffers through two alternating device staging buffers filled on a COPY stream: the
        # host-to-device copy of batch n + 1 (5 MB for VTN vc1: ~0.2 ms of the step's stream when it is issued there) runs beside
        # the graphs of batch n, the step's stream only does a device-to-device copy
        self.copy_stream = "lazy"
        self._stage = {}
        self.pool = torch.cuda.graph_pool_handle()     # one memory pool for the graphs of all shapes
        if trainer.gradient_accumulate_steps != 1 and not trainer.GRAPH_ACCUMULATE:
            raise NotImplementedError(f'config["hip_graph"]: {type(trainer).__name__} has no captured micro-steps '
                                      "(gradient_accumulate_steps must be 1)")
        from ..schedulers import FusedWarmupLR
        if trainer.scheduler is not None and not isinstance(trainer.scheduler, FusedWarmupLR):
            raise NotImplementedError('config["hip_graph"]: the learning-rate schedule must live in the fused optimiser step '
                                      "(schedulers.FusedWarmupLR); a host-side scheduler would not run during replays")
        if trainer.dist is not None and trainer.dp is None and getattr(trainer, "fx", None) is None:
            raise NotImplementedError('config["hip_graph"] with config["distributed"] needs a model with dp_plan() (staged backward)')

    def invalidate(self):
        """Forget every captured step (the set of trainable parameters changed: Trainer.freeze_modules)."""
        for e in self.entries.values():
            e.graphs, e.static, e.bank = [], {}, None
        self.entries.clear()
        self._stage.clear()

    # -- batch -> static buffers ------------------------------------------------------------------
    def _fields(self, batch):
        spec = self.t.GRAPH_BATCH
        if spec is None:
            raise NotImplementedError(f'{type(self.t).__name__} has no captured step (config["hip_graph"])')
        return spec

    def _key(self, batch, spec):
        shapes = tuple((name, int(batch[name].shape[0]), _round_up(batch[name].shape[1], self.quantum)) + tuple(batch[name].shape[2:])
                       for name in spec)
        return (self.t._graph_regime(), shapes)

    def _load(self, e, batch, spec):
        dev = self.t.device
        for name, (lname, pad) in spec.items():
            src = batch[name]
            T = src.shape[1]
            Tb = _round_up(T, self.quantum)
            st = e.static.get(name)
            if st is None:
                st = torch.empty((src.shape[0], Tb) + tuple(src.shape[2:]), dtype=src.dtype, device=dev)
                e.static[name] = st
                e.caps.setdefault(lname, Tb)
            if self.copy_stream is not None and not src.is_cuda:
                self._staged_copy(name, st, src, T)
            else:
                st[:, :T].copy_(src, non_blocking=True)
            if Tb > T:
                st[:, T:].fill_(pad)
            if lname not in e.lens_src:
                e.lens_src[lname] = torch.zeros(src.shape[0], dtype=torch.long)
            e.lens_src[lname].copy_(torch.as_tensor(batch[lname]).to(torch.long))
        return {**{k: v for k, v in batch.items() if k not in e.static and k not in e.lens_src}, **e.static, **e.lens_src}

    def _staged_copy(self, name, st, src, T):
        key = (name, tuple(st.shape), st.dtype)
        slot = self._stage.get(key)
        if self.copy_stream == "lazy":                 # (torch hands out stream objects round-robin: take one nothing else uses)
            from ..ops import functional as Fn
            self.copy_stream = Fn.distinct_stream(Fn._taken_streams() | {self.stream.cuda_stream})
        if slot is None:
            slot = self._stage[key] = {"buf": [torch.empty_like(st), torch.empty_like(st)], "free": [None, None], "i": 0}
            # the buffers were allocated on the step's stream: a block the caching allocator has just recycled may still be in
            # use by kernels queued there -- the copy stream's first write waits for them once, and the allocator learns that
            # the copy stream uses the blocks (so they are not handed out again while a copy is in flight)
            self.copy_stream.wait_stream(torch.cuda.current_stream())
            for b in slot["buf"]:
                b.record_stream(self.copy_stream)
        i = slot["i"]
        slot["i"] ^= 1
        buf, cs = slot["buf"][i], self.copy_stream
        if slot["free"][i] is not None:
            cs.wait_event(slot["free"][i])            # the device copy that last read this staging buffer
        with torch.cuda.stream(cs):
            buf[:, :T].copy_(src, non_blocking=True)
            landed = torch.cuda.Event()
            landed.record(cs)
        main = torch.cuda.current_stream()
        main.wait_event(landed)
        st[:, :T].copy_(buf[:, :T])
        done = torch.cuda.Event()
        done.record(main)
        slot["free"][i] = done

    def _roots(self, e, bank):
        for lname, src in e.lens_src.items():
            bank.root(lname, src, src.tolist(), e.caps[lname])

    # -- one step -----------------------------------------------------------------------------------
    def step(self, batch):
        t = self.t
        spec = self._fields(batch)
        batch = t._batch_dict(batch)
        key = self._key(batch, spec)
        e = self.entries.get(key)
        if e is None:
            while len(self.entries) >= self.max_entries:          # bounded: drop the least recently used shape
                old_key, old = self.entries.popitem(last=False)
                gone = {(n, tuple(v.shape), v.dtype) for n, v in old.static.items()}
                old.graphs, old.static, old.bank = [], {}, None       # graphs first: their blocks go back to the shared pool
                # the staging buffers (2 x the batch per field and shape) of shapes no remaining entry uses go with it
                live = {(n, tuple(v.shape), v.dtype) for ent in self.entries.values() for n, v in ent.static.items()}
                for k in gone - live:
                    self._stage.pop(k, None)
                self.evictions += 1
                logging.info(f"hip_graph: evicted the captured step of {old_key[1]} ({self.evictions} evictions so far; "
                             f'config["graph_cache_size"] = {self.max_entries})')
            e = self.entries[key] = _Entry()
        else:
            self.entries.move_to_end(key)
        e.sightings += 1
        static_batch = self._load(e, batch, spec)
        if self.trace_only or e.sightings < self.capture_after:
            bank = Mo.LensBank(t.device)
            with Mo.lens_bank(bank):
                self._roots(e, bank)
                bank.upload()
                t._train_step(static_batch)
            return
        if not e.graphs:
            self._capture(e, static_batch)
            e.bank.upload()
            logging.info(f"hip_graph: captured the step for {key[1]} ({len(self.entries)} of at most {self.max_entries} shapes cached)")
        else:
            e.bank.refresh({lname: src.tolist() for lname, src in e.lens_src.items()})
            t.steps += e.deltas[0]
            t.backward_steps += e.deltas[1]
            if e.zero_due_after is not None:
                # begin_step() / zero_grad(defer=True) ran as Python only at capture time: a replay must leave the flag the way the
                # captured step did, or the next window's role key (Trainer._graph_regime) picks an accumulate graph without the
                # zero-fill (no path sets the flag since the prologue overlap was removed)
                t.optimizer._zero_due = e.zero_due_after
            t._check_train_finish()
            t.optimizer._touch()        # the weights change without a Python-side optimizer.step(): cached decode sessions etc. go stale
        self._replay(e)

    def _replay(self, e):
        dp, fx = self.t.dp, getattr(self.t, "fx", None)
        for g, stage in e.graphs:
            if stage == "opt" and dp is not None:
                dp.finish()
            if stage == "opt" and fx is not None and any(s == "fx" for _, s in e.graphs):
                fx.finish()
            g.replay()
            if isinstance(stage, int) and dp is not None:
                dp.begin_reduce(stage)
            if stage == "fx" and fx is not None:
                fx.select(e.fx_key)
                fx.issue()                       # every bucket behind the mark of the flush that finished it (nodes of the graph just launched)

    # -- capture ------------------------------------------------------------------------------------
    def _capture(self, e, static_batch):
        t = self.t
        e.bank = Mo.LensBank(t.device)
        before = (t.steps, t.backward_steps)
        mode = "thread_local" if t.dist is not None else "global"
        torch.cuda.synchronize()
        from ..ops import functional as Fn
        if self.stream.cuda_stream in Fn._taken_streams():          # an alias out of torch's round-robin stream pool
            self.stream = Fn.distinct_stream()
        self.stream.wait_stream(torch.cuda.current_stream())
        cap = _Capture(self, e, mode)
        t._capture = cap if (t.dp is not None or getattr(t, "fx", None) is not None) else None
        # no cyclic garbage collection while the stream captures: the destructor of a CUDAGraph / stream of an earlier life calls
        # the runtime, which refuses during a capture (seen as an abort from the autograd thread)
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            with torch.cuda.stream(self.stream), Mo.lens_bank(e.bank):
                self._roots(e, e.bank)
                cap.begin(None)
                t._train_step(static_batch)
                cap.end()
        except BaseException:
            import sys
            import traceback
            traceback.print_exc()                   # shown even if tearing the capture down takes the process with it
            sys.stderr.flush()
            if cap.cur is not None:                 # leave capture mode before the exception travels (a CUDAGraph destroyed
                try:                                # while its stream is capturing aborts the process)
                    cap.cur[0].capture_end()
                except Exception:  # noqa: BLE001
                    pass
            e.graphs = []
            raise
        finally:
            t._capture = None
            if gc_was_on:
                gc.enable()
        torch.cuda.current_stream().wait_stream(self.stream)
        e.bank.closed = True
        e.deltas = (t.steps - before[0], t.backward_steps - before[1])
        if hasattr(t.optimizer, "_zero_due"):
            e.zero_due_after = bool(t.optimizer._zero_due)


class _Capture:
    """The sequence of graphs of one step: Trainer._backward cuts it at the stage borders when the backward pass is staged."""

    def __init__(self, owner, entry, mode):
        self.owner, self.entry, self.mode = owner, entry, mode
        self.cur = None

    def begin(self, stage):
        g = torch.cuda.CUDAGraph()
        g.capture_begin(pool=self.owner.pool, capture_error_mode=self.mode)
        self.cur = (g, stage)

    def end(self):
        g, stage = self.cur
        from ..ops import functional as Fn
        main = torch.cuda.current_stream()
        for st in [Fn._Branch.stream] + list(Fn._Side.streams):      # belt and braces: nothing may still be forked off
            if st is None:
                continue
            with torch.cuda.stream(st):
                forked = torch.cuda.is_current_stream_capturing()
            if forked:
                main.wait_stream(st)
        g.capture_end()
        self.entry.graphs.append((g, stage))
        self.cur = None

    def flush_backward(self, fx, total, scale):
        """The uncut backward pass with a mark (an event-record node) behind every flushed gradient batch closes the first graph;
        what follows (the optimiser) is its own graph, replayed behind the exchange."""
        from ..ops import functional as Fn
        self.cur = (self.cur[0], "fx")
        self.entry.fx_key = fx.key           # the plan (training regime) this graph's marks belong to
        with fx.recording():
            Fn.root_backward(total, scale)
            Fn.side_join()
        fx.mark_end()
        self.end()
        self.begin("opt")

    def staged_backward(self, dp, parts):
        """forward + stage 0 close the first graph; every further stage is its own graph; what follows (the optimiser) too."""
        n = len(dp.plan)
        for i in range(n):
            if i > 0:
                self.begin(i)
            else:
                self.cur = (self.cur[0], 0)
            dp.run_stage(i, parts)
            self.end()
        dp.cuts.clear()
        self.begin("opt")
