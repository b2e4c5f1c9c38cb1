"""Differentiable ops of the hot path: torch.autograd.Function wrappers whose forward AND backward
are hand-written HIP kernels (seq2seq_vc_amd/csrc) reached through the C ABI.

Conventions
  * activations are (B, T, D) channel-last, contiguous, in the compute dtype (fp32 = parity mode,
    bf16 = benchmark mode); parameters stay fp32 in the reference's torch layouts;
  * masks are never materialised: ops take int32 length vectors that live on the device;
  * weight gradients are fp32.  If a parameter carries `_s2s_grad` (a view into the model's flat
    gradient buffer, see optim.FlatAdam) the wgrad kernel accumulates straight into it and autograd
    gets no tensor for that input.
"""
import math
import os

import torch
from torch.autograd import Function

from . import kernels as K
from . import kernels_attn as KAT

_STATE = {"dtype": torch.float32}


def set_compute_dtype(dtype):
    if dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("compute dtype must be float32 or bfloat16")
    _STATE["dtype"] = dtype


def compute_dtype():
    return _STATE["dtype"]


def to_compute(x):
    """Cast an input tensor to the compute dtype with the HIP cast kernel."""
    return K.cast(x.contiguous(), _STATE["dtype"])


def _c(x):
    return x if x.is_contiguous() else K.dense_rows(x)


def _wcast(w, dtype):
    """fp32 master weight -> compute-dtype operand (uses the optimiser's bf16 shadow when present)."""
    if dtype == torch.float32:
        return w
    sh = getattr(w, "_s2s_bf16", None)
    if sh is not None:
        return sh
    return K.cast(w.detach(), dtype)


def _dgrad_operand(weight, w, n_out, n_in, dtype):
    """The weight as B operand of dX[M, n_in] = dY[M, n_out] . W[n_out, n_in]  (B(row = k', red = n') = W[n', k']).
    With the optimiser's transposed bf16 shadow (FlatAdam, `_s2s_bf16_t` = W^T (n_in, n_out)) that operand is
    K-contiguous and the GEMM runs on the all-DMA kernel; otherwise W is read row-contiguous (register transposes)."""
    wt = getattr(weight, "_s2s_bf16_t", None) if dtype == torch.bfloat16 else None
    if wt is not None:
        weight._s2s_perm_registry.sync()       # the transposed shadow is refreshed by the step prologue (optim.FlatAdam.begin_step);
        #                                        every tensor that carries `_s2s_bf16_t` carries the registry (optim.FlatAdam)
        return K.operand(wt, n_out)
    return K.operand(w, n_in, layout=K.RC)


def _emit_wgrad(param, shape, writer):
    """Run `writer(out, accumulate)` into the flat-grad slot of `param` if it has one (returns None so
    autograd skips it), else into a fresh fp32 tensor (returned to autograd)."""
    slot = getattr(param, "_s2s_grad", None)
    if slot is not None:
        writer(slot.view(shape), True)
        return None
    out = torch.empty(shape, dtype=torch.float32, device=param.device)
    writer(out, False)
    return out


def _emit_vgrad(param, value):
    """Same for small vector gradients already computed in fp32 (`value`)."""
    slot = getattr(param, "_s2s_grad", None)
    if slot is not None:
        K.axpby(1.0, slot.view(-1), 1.0, value.reshape(-1), out=slot.view(-1))
        return None
    return value.view(param.shape)


def _emit_permuted(param, dwp, n, A, Bn):
    """The weight gradient `dwp` (n, A, Bn) as its GEMM left it -> the parameter's (n, Bn, A) layout, accumulated into the flat-gradient
    slot if the parameter has one: ONE launch through LDS (K.permute_inner) instead of an element-wise gather + an axpby."""
    if K.permute_inner_ok(A, Bn):
        slot = getattr(param, "_s2s_grad", None)
        if slot is not None:
            K.permute_inner(dwp, n, A, Bn, out=slot, accumulate=True)
            return None
        return K.permute_inner(dwp, n, A, Bn).view(param.shape)
    return _emit_vgrad(param, K.gather3(dwp, (n, Bn, A), (A * Bn, 1, Bn), 0, torch.float32))


# ------------------------------------------------------------------------------------------------
# Gradient cuts (data-parallel overlap).  A model's forward marks a few tensors with cut_point(x, name).  Normally that is
# the identity.  Inside `with grad_cuts(GradCuts([...]))` the autograd graph is cut there: the consumer sees a detached leaf,
# so loss.backward() stops at it, and `cuts.resume(name)` later continues the backward pass below the cut.  Everything
# above a cut has then finished its parameter gradients and their all-reduce can travel while the rest of the backward
# pass runs (distributed.OverlappedBackward); each piece can also be captured as its own hipGraph with the collectives
# issued between the replays.  Values and gradients are exactly those of the uncut graph.
# ------------------------------------------------------------------------------------------------


class GradCuts:
    def __init__(self, names):
        self.names = set(names)
        self.points = {}          # name -> (tensor above the cut [graph side of the producer], detached leaf the consumers see)

    def cut(self, x, name):
        """x: a tensor, or a tuple of tensors crossing the cut together (e.g. the residual stream of a pre-norm layer stack
        and the feed-forward output still to be added to it); None members pass through."""
        if name not in self.names or not torch.is_grad_enabled():
            return x
        xs = x if isinstance(x, tuple) else (x,)
        if not any(t is not None and t.requires_grad for t in xs):
            return x
        if name in self.points:
            raise RuntimeError(f"gradient cut '{name}' was reached twice in one forward pass")
        inner = tuple(t.detach().requires_grad_(True) if (t is not None and t.requires_grad) else t for t in xs)
        self.points[name] = (xs, inner)
        return inner if isinstance(x, tuple) else inner[0]

    def resume(self, name):
        """Continue the backward pass below the cut (no-op if the forward pass never reached it or nothing above it
        needed the gradient)."""
        outer, inner = self.points.get(name, (None, None))
        if outer is None:
            return
        pairs = [(o, i.grad) for o, i in zip(outer, inner) if o is not None and o.requires_grad and i is not o and i.grad is not None]
        if pairs:
            torch.autograd.backward([o for o, _ in pairs], [g for _, g in pairs], retain_graph=False)

    def clear(self):
        self.points.clear()


_CUTS = {"active": None}


class grad_cuts:
    def __init__(self, cuts):
        self.cuts = cuts

    def __enter__(self):
        self.prev, _CUTS["active"] = _CUTS["active"], self.cuts
        if self.cuts is not None:
            self.cuts.clear()
        return self.cuts

    def __exit__(self, *exc):
        _CUTS["active"] = self.prev
        return False


def cut_point(x, name):
    c = _CUTS["active"]
    return x if c is None else c.cut(x, name)


# ------------------------------------------------------------------------------------------------
# Side streams for parameter-gradient work.  Weight / bias / LayerNorm gradients are only consumed by
# the optimiser at the end of the step, so (when they land in flat-gradient slots) their kernels are
# issued on a small pool of side HIP streams and overlap with the data-gradient chain on the main
# stream; `side_join()` must be called after backward() and before the optimiser step.  Under hipGraph
# capture the cross-stream waits become graph edges, i.e. parallel branches of the captured step.
# ------------------------------------------------------------------------------------------------
class _Side:
    enabled = False
    streams = []
    idx = 0
    pending = []     # tensors that must stay alive until the join (their memory is in use on a side stream)
    queue = []       # closures waiting for the next fork point
    origins = []     # the stream each queued closure was issued from
    inline = False   # no side streams, but batched (see enable_side_streams)
    inline_q = {}    # stream handle -> (stream, [closures])
    batch = 16
    grouped = []     # weight-gradient GEMM descriptors of the batch being flushed
    grouped_cr = []  # column reductions of the batch being flushed
    on_flush = None  # callable(): called on the flushing stream behind every flushed batch (distributed.FlushExchange)


def _taken_streams():
    h = {st.cuda_stream for st in _Side.streams}
    if _Branch.stream is not None:
        h.add(_Branch.stream.cuda_stream)
    if torch.cuda.is_available():
        h.add(torch.cuda.current_stream().cuda_stream)
    return h


def distinct_stream(taken=None):
    """A torch.cuda.Stream whose underlying stream is none of `taken` (default: the side streams, the auxiliary stream and the
    current stream).  torch hands out Stream objects round-robin from a pool of 32 per device: a process that has created many
    (every trainer / test / decode session creates a few) gets ALIASES of earlier ones, and two roles that the scheduling here
    keeps apart -- e.g. the auxiliary stream of a branch and a side stream of the gradient work -- end up on one stream, which a
    capture with forks and joins between them does not survive (seen as a segmentation fault in hipStreamEndCapture)."""
    taken = _taken_streams() if taken is None else set(taken)
    for _ in range(96):
        st = torch.cuda.Stream()
        if st.cuda_stream not in taken:
            return st
    raise RuntimeError("no distinct stream left in torch's stream pool")


def enable_side_streams(n=4, inline_batches=False, batch=None):
    """batch: closures per gradient batch (default: 16 forked / 64 inline).  Round 4 re-measured both with
    the 8-wave weight-gradient kernel that takes 40 problems per launch: VTN 12 -> 16: 3.95 -> 3.86 ms, AAS-VC 12 -> 48 ... 160:
    11.85 -> 11.6-11.7 ms (larger grids, fewer ragged last rounds; the operands stay alive a little longer).
    n > 0: parameter-gradient work is forked to n side streams (small, latency-bound models: VTN).
    n == 0 and inline_batches: the work stays on the stream that issued it but is still queued and run in batches, so
    that the dense weight-gradient GEMMs of a batch become one grouped launch -- for models whose kernels fill the chip
    anyway (AAS-VC: d = 1536) the forks cost more than the overlap gives (19.3 vs 20.9 ms/step).  Both need side_join()
    between backward and the optimiser step; n == 0 without inline_batches runs everything immediately."""
    K.set_wgrad_cap(64 if n > 0 else 0)
    _Side.enabled = n > 0
    _Side.inline = (n == 0) and inline_batches
    _Side.batch = int(batch) if batch else (64 if _Side.inline else 16)
    old, others = _Side.streams, _taken_streams() - {st.cuda_stream for st in _Side.streams}
    if n > 0 and len(old) == n and len({st.cuda_stream for st in old}) == n and not ({st.cuda_stream for st in old} & others):
        pass                                     # keep the ones we have: every new Stream object eats a slot of torch's pool
    else:
        _Side.streams = []
        for _ in range(n):
            _Side.streams.append(distinct_stream())
    _Side.idx = 0


def _side_run(fn, keep=(), solo=False):
    """Parameter-gradient work off the data-gradient chain: `fn` is queued and runs on a side stream in batches of
    `_Side.batch` closures -- one fork point (cross-stream edge of the captured graph) per batch instead of one per call.
    The stream the caller runs on is remembered: backward nodes of a branch (branch_run) execute on the branch's stream.
    solo=True: `fn` is forked NOW as a batch of its own (long kernels at the end of the backward pass: whatever shares their
    batch runs behind them on the same stream)."""
    if solo and _Side.enabled and not _Side.inline:
        held, held_o = _Side.queue, _Side.origins
        _Side.queue, _Side.origins = [fn], [torch.cuda.current_stream()]
        _Side.pending.append(keep)
        _side_flush()
        _Side.queue, _Side.origins = held, held_o
        return
    if _Side.inline:
        cur = torch.cuda.current_stream()
        q = _Side.inline_q.setdefault(cur.cuda_stream, (cur, []))[1]
        q.append(fn)
        _Side.pending.append(keep)
        if len(q) >= _Side.batch:
            _inline_flush(cur.cuda_stream)
        return
    if not _Side.enabled:
        fn()
        return
    _Side.queue.append(fn)
    _Side.origins.append(torch.cuda.current_stream())
    _Side.pending.append(keep)
    if len(_Side.queue) >= _Side.batch:
        _side_flush()


def _run_batch(closures):
    """Run a batch of gradient closures on the current stream: their dense weight-gradient GEMMs become ONE grouped launch
    (no split-K, no reduction passes), their column reductions into gradient slots (LayerNorm / BatchNorm / bias vectors)
    two grouped launches; everything else they launch (conv weight gradients, ...) runs as issued."""
    with K.record_grouped(_Side.grouped), K.record_colreduce(_Side.grouped_cr):
        for fn in closures:
            fn()
    if _Side.grouped:
        K.flush_grouped(_Side.grouped)
    if _Side.grouped_cr:
        K.flush_colreduce(_Side.grouped_cr)


def _inline_flush(key):
    """Run the closures queued from one stream ON that stream (no fork), the dense weight gradients as a grouped launch."""
    st, q = _Side.inline_q.pop(key)
    with torch.cuda.stream(st):
        _run_batch(q)
        if _Side.on_flush is not None:
            _Side.on_flush()
    return st


def _side_flush():
    if _Side.inline:
        return
    if not _Side.queue:
        return
    k = _Side.idx % len(_Side.streams)
    st = _Side.streams[k]
    cur = torch.cuda.current_stream().cuda_stream
    if st.cuda_stream == cur or (_Branch.stream is not None and st.cuda_stream == _Branch.stream.cuda_stream):
        st = _Side.streams[k] = distinct_stream()        # an alias (see distinct_stream): e.g. the capture stream of a later graph
    _Side.idx += 1
    seen = set()
    for origin in _Side.origins + [torch.cuda.current_stream()]:     # every stream that produced an input of the batch
        if origin.cuda_stream not in seen:
            seen.add(origin.cuda_stream)
            st.wait_stream(origin)
    _Side.origins = []
    with torch.cuda.stream(st):
        _run_batch(_Side.queue)
        if _Side.on_flush is not None:
            _Side.on_flush()
    _Side.queue = []


# ------------------------------------------------------------------------------------------------
# Independent sub-networks (e.g. the duration predictor next to the length regulator + decoder of AAS-VC) can run on
# an auxiliary stream: hundreds of small dependent kernels then fill the gaps of the other branch instead of extending
# the chain.  torch's autograd engine runs every backward node on the stream of its forward op, so the backward pass of
# the branch overlaps as well; under hipGraph capture fork and join are two graph edges.
# ------------------------------------------------------------------------------------------------
_NO_BRANCH = os.environ.get("S2SVC_NO_BRANCH", "0") == "1"      # diagnostic: branches run in line on the issuing stream


class _Branch:
    stream = None
    active = False
    dirty = False        # work was put on the auxiliary stream since the last join of a backward pass (side_join)


def branch_run(fn, uses=()):
    """Run fn() on the auxiliary stream, after everything queued on the current stream.  Call branch_join() on the
    current stream before anything there consumes the results.
    `uses`: the tensors of the CURRENT stream that fn reads (or saves for its backward pass).  They are recorded on the auxiliary
    stream: the caching allocator hands a block back to the stream that allocated it the moment the last reference goes, and
    the last reference to these is dropped by the branch (autograd releasing what its nodes saved) while the kernel that
    reads them may not have run yet -- a kernel of the allocating stream queued after that point could then overwrite the block
    under the reader.  Eager launches hide this (the reader was launched first and the GPU keeps up); in a captured graph
    only dependencies order the two streams (seen as wrong duration-predictor gradients in the AAS-VC stage graphs: the
    alignment search's durations `ds`, saved by the flow's backward pass, held another tensor by the time it read them)."""
    if not torch.cuda.is_available() or _NO_BRANCH:
        return fn()
    main = torch.cuda.current_stream()
    if _Branch.stream is None or _Branch.stream.cuda_stream == main.cuda_stream:
        _Branch.stream = distinct_stream()
    _Branch.stream.wait_stream(main)
    for t in uses:
        if isinstance(t, torch.Tensor) and t.is_cuda:
            t.record_stream(_Branch.stream)
    with torch.cuda.stream(_Branch.stream):
        out = fn()
    _Branch.active = True
    _Branch.dirty = True
    return out


def branch_join(*results):
    """Make the current stream wait for the auxiliary stream; `results` are tensors produced there that the current
    stream goes on to use (their memory must not return to the auxiliary stream's pool while that use is pending)."""
    if _Branch.active:
        main = torch.cuda.current_stream()
        main.wait_stream(_Branch.stream)
        for t in results:
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(main)
        _Branch.active = False


def branch_backward(loss, fork_event, retain_graph=False, scale=1.0):
    """loss.backward() rooted on the auxiliary stream, which starts from `fork_event` (recorded on the calling stream at the
    start of a backward stage).  Call it BEFORE the stage's other roots and branch_wait() after them: the nodes of a
    sub-network that ran under branch_run() go to the auxiliary stream, the loss arithmetic and whatever else of this root ran on
    the calling stream goes there while it is still empty, and the other roots then queue beside the branch.  (A root processed
    on the calling stream after other work puts its first nodes -- and with them the whole branch -- behind that work; one
    processed there before it stalls the caller at the first node that consumes a result of the branch.)"""
    if _Branch.stream is None or _NO_BRANCH:
        root_backward(loss, scale, retain_graph)
        return
    _Branch.stream.wait_event(fork_event)
    _Branch.dirty = True
    with torch.cuda.stream(_Branch.stream):
        root_backward(loss, scale, retain_graph)


def branch_resume(cuts, name, fork_event):
    """cuts.resume(name) rooted on the auxiliary stream (see branch_backward): the part of a branch below a gradient cut, run one
    stage after the part above it."""
    if _Branch.stream is None or _NO_BRANCH:
        cuts.resume(name)
        return
    _Branch.stream.wait_event(fork_event)
    _Branch.dirty = True
    with torch.cuda.stream(_Branch.stream):
        cuts.resume(name)


def branch_wait():
    """The current stream waits for what has been queued on the auxiliary stream (end of a stage that used branch_backward)."""
    if _Branch.stream is not None and not _NO_BRANCH:
        torch.cuda.current_stream().wait_stream(_Branch.stream)


def side_join():
    """Run what is still queued and make the current stream wait for all side-stream gradient work (call between
    backward and optimiser)."""
    if _Side.inline:
        main = torch.cuda.current_stream()
        for key in list(_Side.inline_q):
            st = _inline_flush(key)
            if st.cuda_stream != main.cuda_stream:
                main.wait_stream(st)
    if _Side.enabled:
        # the batch flushed HERE runs behind the end of the data-gradient chain: there is nothing left to protect from a
        # chip-filling weight-gradient grid, so its launches are not capped (ops.kernels.set_wgrad_cap)
        cap = K.get_wgrad_cap()
        K.set_wgrad_cap(0)
        try:
            _side_flush()
        finally:
            K.set_wgrad_cap(cap)
        main = torch.cuda.current_stream()
        for st in _Side.streams:
            main.wait_stream(st)
    _join_branch_stream()
    K.audit_reset()                  # (writer audit, ops.kernels._Audit: a join -- every slot may change hands)
    _Side.pending.clear()
    _Side.idx = 0


def _join_branch_stream():
    """Backward nodes of a sub-network that ran under branch_run() execute on the auxiliary stream; nothing guarantees that the
    last of them is followed by work another stream waits for (autograd only joins streams that ran gradient-accumulation
    nodes; the parameter gradients here go straight to their slots).  So the join after a backward pass (side_join) also joins
    the auxiliary stream -- if a branch was started since the last join (a later stage graph of a staged backward pass has no
    part of the branch in it, and must not wait for an event of another capture)."""
    if _Branch.stream is not None and _Branch.dirty:
        torch.cuda.current_stream().wait_stream(_Branch.stream)
    _Branch.dirty = False


def _slotted(*params):
    return all(p is None or getattr(p, "_s2s_grad", None) is not None for p in params)


def _bias_sink(bias, n):
    """Where a wgrad GEMM should put the fused bias gradient: (buffer, accumulate, value_for_autograd)."""
    if bias is None or not bias.requires_grad:
        return None, False, None
    slot = getattr(bias, "_s2s_grad", None)
    if slot is not None:
        return slot.view(-1), True, None
    buf = torch.empty(n, dtype=torch.float32, device=bias.device)
    return buf, False, buf.view(bias.shape)


def _reduce_to(p_sum, p_dot, mode, dy, x=None, mean=None, rstd=None):
    """Column reduction whose results are the gradients of `p_sum` (sum_r dy) and `p_dot` (sum_r dy*xhat): written
    (accumulated) straight into their flat-gradient slots when they have them -> (None, None) for autograd."""
    s_slot = getattr(p_sum, "_s2s_grad", None)
    d_slot = getattr(p_dot, "_s2s_grad", None) if p_dot is not None else None
    if s_slot is not None and (p_dot is None or d_slot is not None):
        K.colreduce(mode, dy, x, mean, rstd, want_dot=p_dot is not None, out_sum=s_slot.view(-1),
                    out_dot=None if d_slot is None else d_slot.view(-1), accumulate=True)
        return None, None
    s, d = K.colreduce(mode, dy, x, mean, rstd, want_dot=p_dot is not None)
    return s.view(p_sum.shape), (None if d is None else d.view(p_dot.shape))


# ================================================================================================
# Linear (+bias, +activation)          reference: torch.nn.Linear call sites of the hot path
# ================================================================================================


class _Linear(Function):
    """y = act(x W^T + b).  passthrough=True additionally returns an alias of x: a post-LN residual that takes x from there
    sends its gradient back through this node, where it rides in the epilogue of the data-gradient GEMM (dX = dY W + g_pass)
    instead of costing an element-wise add on the main chain."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, passthrough=False):
        dtype = x.dtype
        x2 = _c(x).view(-1, x.shape[-1])
        M, Kd = x2.shape
        N = weight.shape[0]
        w = _wcast(weight, dtype)
        y = torch.empty((M, N), dtype=dtype, device=x.device)
        K.gemm(K.operand(x2, Kd), K.operand(w, Kd), M, N, Kd, y, in_dtype=dtype, bias=bias, act=act)
        ctx.act = act
        ctx.has_bias = bias is not None
        ctx.params = (weight, bias)
        ctx.save_for_backward(x2, w, y if act else None)
        ctx.xshape = x.shape
        if passthrough:
            ctx.set_materialize_grads(False)
            return y.view(*x.shape[:-1], N), x.view_as(x)
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy, g_pass=None):
        if dy is None:                 # only the pass-through output was used
            return g_pass, None, None, None, None
        x2, w, y = ctx.saved_tensors
        weight, bias = ctx.params
        M, Kd = x2.shape
        N = w.shape[0]
        dtype = x2.dtype
        dy2 = _c(dy).view(M, N)
        if ctx.act:
            dy2 = K.act_dropout_bwd(dy2, y, act=ctx.act)
        ldy, zp, dy_rows = N, False, dy2
        if dtype == torch.float32 and N % 4 and dy2.is_cuda:
            # rows of N fp32 values with N % 4 != 0 (the 29 spline parameters of a ConvFlow) keep both gradient GEMMs on the
            # element-wise fallback kernel (19 + 37 us per flow): a zero-padded copy with rows of a whole number of 16-byte vectors
            # puts them on the vectorised kernels (the pad columns contribute zeros to the reductions and are never stored)
            ldy, zp = (N + 3) // 4 * 4, True
            dy2 = K.pad_cols(dy2, ldy)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((M, Kd), dtype=dtype, device=dy.device)
            res = _c(g_pass).view(M, Kd).to(dtype) if g_pass is not None else None
            K.gemm(K.operand(dy2, ldy, zero_padded=zp), _dgrad_operand(weight, w, N, Kd, dtype), M, Kd, N, dx, in_dtype=dtype, res=res)
            dx = dx.view(ctx.xshape)
        dw = db = None
        if weight.requires_grad:
            tile, sk = K.plan_gemm(N, Kd, M, dtype=dtype)
            rs, racc, db = _bias_sink(bias, N)     # bias gradient = row sums of dY^T, fused into the wgrad GEMM

            def wr(out, acc):
                K.gemm(K.operand(dy2, ldy, layout=K.RC, zero_padded=zp), K.operand(x2, Kd, layout=K.RC), N, Kd, M, out, in_dtype=dtype,
                       splitk=sk, tile=tile, accumulate=acc, a_rowsum=rs, a_rowsum_accumulate=racc)
            if _slotted(weight, bias if ctx.has_bias else None):
                _side_run(lambda: _emit_wgrad(weight, (N, Kd), wr), keep=(dy2, x2))
            else:
                dw = _emit_wgrad(weight, (N, Kd), wr)
                if dw is not None:
                    dw = dw.view(weight.shape)  # 1x1 Conv1d weights (N, K, 1) are accepted as Linear weights
        elif ctx.has_bias and bias.requires_grad:
            db, _ = _reduce_to(bias, None, 0, dy_rows)
        return dx, dw, db, None, None


def linear(x, weight, bias=None, act=None, passthrough=False):
    """passthrough=True -> (y, x_alias), see _Linear."""
    return _Linear.apply(x, weight, bias, act, passthrough)


class _FFNRelu(Function):
    """w_2(dropout(act(w_1 x)))  (positionwise_feed_forward.py:30-32), act = relu (Transformer) or swish (Conformer), as two
    GEMMs forward and four backward: activation and dropout mask ride in the first GEMM's epilogue, and the dgrad GEMM
    through w_2 applies dropmask * act' in its epilogue -- relu: (h > 0 <=> active and kept) read off the stored output;
    swish: swish'(u) of the pre-activation u, which the first GEMM writes as a second output -- so no element-wise pass
    over the (rows, hidden) tensor remains."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, p, passthrough=False, act="relu"):
        dtype = x.dtype
        x2 = _c(x).view(-1, x.shape[-1])
        M, Kd = x2.shape
        Hd, N = w1.shape[0], w2.shape[0]
        w1c, w2c = _wcast(w1, dtype), _wcast(w2, dtype)
        seed = K.new_seed(x.device) if p > 0.0 else (None, 0)
        h = torch.empty((M, Hd), dtype=dtype, device=x.device)
        u = torch.empty((M, Hd), dtype=dtype, device=x.device) if act != "relu" else None
        K.gemm(K.operand(x2, Kd), K.operand(w1c, Kd), M, Hd, Kd, h, in_dtype=dtype, bias=b1, act=act, drop_p=p, seed=seed, pre_out=u)
        y = torch.empty((M, N), dtype=dtype, device=x.device)
        K.gemm(K.operand(h, Hd), K.operand(w2c, Hd), M, N, Hd, y, in_dtype=dtype, bias=b2)
        ctx.params = (w1, b1, w2, b2)
        ctx.meta = (p, seed, x.shape)
        ctx.save_for_backward(x2, h, w1c, w2c, u)
        if passthrough:                # (y, alias of x): see _Linear
            ctx.set_materialize_grads(False)
            return y.view(*x.shape[:-1], N), x.view_as(x)
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy, g_pass=None):
        if dy is None:
            return g_pass, None, None, None, None, None, None, None
        x2, h, w1c, w2c, u = ctx.saved_tensors
        w1, b1, w2, b2 = ctx.params
        p, seed, xshape = ctx.meta
        M, Kd = x2.shape
        Hd, N = w1c.shape[0], w2c.shape[0]
        dtype = x2.dtype
        dy2 = _c(dy).view(M, N)
        # du = (dY W2) * dropmask * relu'(u): gradient at the pre-activation, straight out of the GEMM
        du = torch.empty((M, Hd), dtype=dtype, device=dy.device)
        K.gemm(K.operand(dy2, N), _dgrad_operand(w2, w2c, N, Hd, dtype), M, Hd, N, du, in_dtype=dtype, emask=h if u is None else u,
               emask_mode=0 if u is None else 1, drop_p=p, seed=seed)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((M, Kd), dtype=dtype, device=dy.device)
            res = _c(g_pass).view(M, Kd).to(dtype) if g_pass is not None else None
            K.gemm(K.operand(du, Hd), _dgrad_operand(w1, w1c, Hd, Kd, dtype), M, Kd, Hd, dx, in_dtype=dtype, res=res)
            dx = dx.view(xshape)

        def wgrad(weight, bias, g, a, n_out, n_in):
            tile, sk = K.plan_gemm(n_out, n_in, M, dtype=dtype)
            rs, racc, db = _bias_sink(bias, n_out)

            def wr(out, acc):
                K.gemm(K.operand(g, n_out, layout=K.RC), K.operand(a, n_in, layout=K.RC), n_out, n_in, M, out, in_dtype=dtype,
                       splitk=sk, tile=tile, accumulate=acc, a_rowsum=rs, a_rowsum_accumulate=racc)
            if _slotted(weight, bias):
                _side_run(lambda: _emit_wgrad(weight, (n_out, n_in), wr), keep=(g, a))
                return None, None
            dw = _emit_wgrad(weight, (n_out, n_in), wr)
            return (dw.view(weight.shape) if dw is not None else None), db
        dw2, db2 = wgrad(w2, b2, dy2, h, N, Hd) if w2.requires_grad else (None, None)
        dw1, db1 = wgrad(w1, b1, du, x2, Hd, Kd) if w1.requires_grad else (None, None)
        return dx, dw1, db1, dw2, db2, None, None, None


def ffn_relu(x, w1, b1, w2, b2, p=0.0, passthrough=False):
    return _FFNRelu.apply(x, w1, b1, w2, b2, p, passthrough, "relu")


def ffn_act(x, w1, b1, w2, b2, act, p=0.0, passthrough=False):
    """The same block with `act` in ("relu", "swish")."""
    return _FFNRelu.apply(x, w1, b1, w2, b2, p, passthrough, act)


class _Embedding(Function):
    """Token embedding lookup (models/transformer_tts.py:63-77) in the compute dtype; deterministic weight gradient."""

    @staticmethod
    def forward(ctx, idx, weight, padding_idx):
        idx = idx.contiguous()
        ctx.weight, ctx.padding_idx = weight, padding_idx
        ctx.save_for_backward(idx)
        return K.embedding_fwd(idx, weight.detach(), compute_dtype())

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        weight = ctx.weight
        dw = None
        if weight.requires_grad:
            dw = _emit_wgrad(weight, tuple(weight.shape),
                             lambda out, acc: K.embedding_bwd(idx, _c(dy), weight.shape[0], ctx.padding_idx, out=out, accumulate=acc))
        return None, dw, None


def embedding(idx, weight, padding_idx=None):
    return _Embedding.apply(idx, weight, padding_idx)


# ================================================================================================
# LayerNorm, optionally fused with "s = res + hscale * dropout(h)"
# reference: modules/transformer/layer_norm.py:12-42 + residual/dropout lines of the layer classes
# ================================================================================================
class _AddLayerNorm(Function):
    @staticmethod
    def forward(ctx, h, res, gamma, beta, eps, p, hscale):
        seed = K.new_seed(h.device) if p > 0.0 else (None, 0)
        h = _c(h)
        res = _c(res) if res is not None else None
        y, s, mean, rstd = K.layernorm_fwd(h, gamma, beta, eps, res=res, p=p, seed=seed, hscale=hscale)
        ctx.meta = (eps, p, seed, hscale, res is not None)
        ctx.params = (gamma, beta)
        ctx.save_for_backward(s if res is not None else h, mean, rstd)
        ctx.set_materialize_grads(False)
        if res is None:
            return y
        return y, s

    @staticmethod
    def backward(ctx, dy, ds_direct=None):
        s, mean, rstd = ctx.saved_tensors
        gamma, beta = ctx.params
        eps, p, seed, hscale, fused = ctx.meta
        dy = _c(dy) if dy is not None else torch.zeros_like(s)
        extra = _c(ds_direct) if (fused and ds_direct is not None) else None
        slotted = gamma.requires_grad and _slotted(gamma, beta)
        ds, dh, part = K.layernorm_bwd(dy, s, mean, rstd, gamma, ds_extra=extra, p=p, seed=seed,
                                       want_dh=fused and (p > 0.0 or hscale != 1.0), hscale=hscale, want_partials=True, partials_ok=slotted)
        dgamma = dbeta = None
        if gamma.requires_grad:
            if part is not None:
                # the backward kernel left the first reduction stage of d gamma / d beta (big sites: dy and the LayerNorm input are not
                # read a second time); the second stage joins the batch's grouped column reductions
                ws, chunks = part
                b_slot, g_slot = beta._s2s_grad.view(-1), gamma._s2s_grad.view(-1)
                _side_run(lambda: K.colreduce_partials(ws, chunks, s.shape[-1], b_slot, g_slot), keep=(ws,))
            elif _slotted(gamma, beta):
                _side_run(lambda: _reduce_to(beta, gamma, 1, dy, s, mean, rstd), keep=(dy, s, mean, rstd))
            else:
                dbeta, dgamma = _reduce_to(beta, gamma, 1, dy, s, mean, rstd)
        if fused:
            return (dh if dh is not None else ds), ds, dgamma, dbeta, None, None, None
        return ds, None, dgamma, dbeta, None, None, None


class _LayerNormPass(Function):
    """(LayerNorm(x), alias of x): a pre-LN layer feeds x to its first norm AND to the residual behind the first sub-layer.  With the
    residual taken from the alias both gradients meet inside the LayerNorm backward kernel (its `ds_extra` operand) instead of in an
    element-wise add of autograd's accumulation (8 x 7 us on the chain of an AAS-VC step: profiles/r05_aasvc_train_bf16_timeline.txt)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x = _c(x)
        y, _, mean, rstd = K.layernorm_fwd(x, gamma, beta, eps)
        ctx.params = (gamma, beta)
        ctx.save_for_backward(x, mean, rstd)
        ctx.set_materialize_grads(False)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dx_pass=None):
        x, mean, rstd = ctx.saved_tensors
        gamma, beta = ctx.params
        if dy is None:                                  # only the alias was used
            return dx_pass, None, None, None
        dy = _c(dy)
        slotted = gamma.requires_grad and _slotted(gamma, beta)
        ds, _, part = K.layernorm_bwd(dy, x, mean, rstd, gamma, ds_extra=_c(dx_pass) if dx_pass is not None else None,
                                      want_partials=True, partials_ok=slotted)
        dgamma = dbeta = None
        if gamma.requires_grad:
            if part is not None:
                ws, chunks = part
                b_slot, g_slot = beta._s2s_grad.view(-1), gamma._s2s_grad.view(-1)
                _side_run(lambda: K.colreduce_partials(ws, chunks, x.shape[-1], b_slot, g_slot), keep=(ws,))
            elif _slotted(gamma, beta):
                _side_run(lambda: _reduce_to(beta, gamma, 1, dy, x, mean, rstd), keep=(dy, x, mean, rstd))
            else:
                dbeta, dgamma = _reduce_to(beta, gamma, 1, dy, x, mean, rstd)
        return ds, dgamma, dbeta, None


def layer_norm(x, gamma, beta, eps=1e-12, passthrough=False):
    """passthrough=True -> (LayerNorm(x), alias of x): take a residual that starts at x from the alias (see _LayerNormPass)."""
    if passthrough and torch.is_grad_enabled() and x.requires_grad:
        return _LayerNormPass.apply(x, gamma, beta, eps)
    y = _AddLayerNorm.apply(x, None, gamma, beta, eps, 0.0, 1.0)
    return (y, x) if passthrough else y


class _FanOut(Function):
    """n aliases of x for n consumers: their gradients arrive at ONE autograd node and are summed by one launch of s2svc_add_n (two
    when n > 4) instead of by n - 1 element-wise adds of autograd's accumulation."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        gs = [_c(g) for g in grads if g is not None]
        if not gs:
            return None, None
        while len(gs) > 1:
            gs = [K.add_n(gs[:4])] + gs[4:]
        return gs[0], None


def fan_out(x, n):
    """x for n consumers (see _FanOut); plain copies of the reference when no gradient is recorded."""
    if n < 2 or not (torch.is_grad_enabled() and x.requires_grad):
        return (x,) * n
    return _FanOut.apply(x, n)


def add_dropout_layer_norm(res, h, gamma, beta, eps=1e-12, p=0.0, hscale=1.0):
    """s = res + hscale*dropout(h, p); y = LayerNorm(s).  Returns (y, s)."""
    return _AddLayerNorm.apply(h, res, gamma, beta, eps, p, hscale)


class _AddDropout(Function):
    """s = res + hscale * dropout(h)   (a residual that is not followed by a LayerNorm)."""

    @staticmethod
    def forward(ctx, h, res, p, hscale):
        seed = K.new_seed(h.device) if p > 0.0 else (None, 0)
        hd = K.act_dropout_fwd(_c(h), p=p, seed=seed) if p > 0.0 else _c(h)
        ctx.meta = (p, seed, hscale)
        return K.axpby(1.0, _c(res), hscale, hd)

    @staticmethod
    def backward(ctx, ds):
        p, seed, hscale = ctx.meta
        ds = _c(ds)
        dh = ds
        if p > 0.0:
            dh = K.act_dropout_bwd(ds, ds, act=None, p=p, seed=seed)
        if hscale != 1.0:
            dh = K.axpby(hscale, dh)
        return dh, ds, None, None


def add_dropout(res, h, p=0.0, hscale=1.0):
    return _AddDropout.apply(h, res, p, hscale)


# ================================================================================================
# dropout / activations
# ================================================================================================
class _ActDropout(Function):
    @staticmethod
    def forward(ctx, x, act, p):
        seed = K.new_seed(x.device) if p > 0.0 else (None, 0)
        x = _c(x)
        y = K.act_dropout_fwd(x, act=act, p=p, seed=seed)
        ctx.meta = (act, p, seed)
        ctx.save_for_backward(x if act in ("swish", "gelu") else y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (saved,) = ctx.saved_tensors
        act, p, seed = ctx.meta
        return K.act_dropout_bwd(_c(dy), saved, act=act, p=p, seed=seed), None, None


def act_dropout(x, act=None, p=0.0):
    if act is None and p <= 0.0:
        return x
    return _ActDropout.apply(x, act, p)


def dropout(x, p, training=True):
    return act_dropout(x, None, p if training else 0.0)


class _Glu(Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        ctx.save_for_backward(x)
        return K.glu_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return K.glu_bwd(x, _c(dy))


def glu(x):
    return _Glu.apply(x)


# ================================================================================================
# positional encodings          reference: layers/positional_encoding.py
# ================================================================================================
class _PosEnc(Function):
    @staticmethod
    def forward(ctx, x, alpha, pe, xscale, p):
        seed = K.new_seed(x.device) if p > 0.0 else (None, 0)
        x = _c(x)
        y = K.posenc_fwd(x, xscale, alpha, pe, p=p, seed=seed)
        ctx.meta = (xscale, p, seed)
        ctx.alpha = alpha
        ctx.save_for_backward(pe)
        return y

    @staticmethod
    def backward(ctx, dy):
        (pe,) = ctx.saved_tensors
        xscale, p, seed = ctx.meta
        alpha = ctx.alpha
        want = alpha is not None and alpha.requires_grad
        dx, dalpha = K.posenc_bwd(_c(dy), xscale, pe, p=p, seed=seed, want_dalpha=want)
        if want:
            dalpha = _emit_vgrad(alpha, dalpha)
        return dx, dalpha, None, None, None


def posenc(x, pe, alpha=None, xscale=1.0, p=0.0):
    """y = dropout(x*xscale + alpha*pe[:T])  (pe None -> scaling only)."""
    return _PosEnc.apply(x, alpha, pe, xscale, p)


# ================================================================================================
# attention cores (QK^T -> masked softmax -> PV), plain and relative-position
# reference: modules/transformer/attention.py:63-111, :262-305
# ================================================================================================
def _bop(t, dk, layout=K.KC, bs0=None):
    """GEMM operand over a (B, T, H*dk) activation that may be a column slice of a packed (B, T, n*D) tensor:
    row stride and batch stride come from the view, heads are the second batch level."""
    assert t.stride(-1) == 1
    return K.operand(t, t.stride(1), layout=layout, bs0=t.stride(0) if bs0 is None else bs0, bs1=dk)


def _pad8(n):
    return (n + 7) // 8 * 8


def _qk(q, k, B, H, T1, T2, dk, D, dtype, bs0_b=None):
    """fp32 scores (B, H, T1, ld) with ld = T2 rounded up to 8 (16-byte rows for the bf16 consumers)."""
    ld = _pad8(T2)
    scores = torch.empty((B, H, T1, ld), dtype=torch.float32, device=q.device)
    K.gemm(_bop(q, dk), _bop(k, dk, bs0=bs0_b), T1, T2, dk, scores, in_dtype=dtype, nb0=B, nb1=H, ldc=ld, cbs=(H * T1 * ld, T1 * ld))
    return scores


def _into(out, A, Bop, M, N, Kd, dk, dtype, B, H, group=None):
    """Batched (b, h) GEMM writing head h of batch b into columns [h*dk, (h+1)*dk) of the (B, T, .) view `out`.
    group: a list that collects the problem instead of launching it (K.launch_group_batched)."""
    K.gemm(A, Bop, M, N, Kd, out, in_dtype=dtype, nb0=B, nb1=H, ldc=out.stride(1), cbs=(out.stride(0), dk), group=group)
    return out


def _pop(pm, T1, H, layout=K.KC):
    """Operand over a padded probability-like tensor (B, H, T1, ld)."""
    ld = pm.shape[-1]
    return K.operand(pm, ld, layout=layout, bs0=H * T1 * ld, bs1=T1 * ld, zero_padded=True)


def _pv(pm, v, B, H, T1, T2, dk, D, dtype):
    ctxv = torch.empty((B, T1, D), dtype=dtype, device=v.device)
    return _into(ctxv, _pop(pm, T1, H), _bop(v, dk, K.RC), T1, dk, T2, dk, dtype, B, H)


def _pad_like(dattn, ref):
    """External gradient wrt the (B,H,T1,T2) attention view -> the padded (B,H,T1,ld) layout of `ref`."""
    if dattn is None:
        return None
    if dattn.shape[-1] == ref.shape[-1]:
        return _c(dattn)
    out = torch.zeros_like(ref)
    out[..., : dattn.shape[-1]].copy_(dattn)
    return out


def _attn_common_bwd(dctx, dattn, attn, pm, q, k, v, H, scale, p, seed, Lp=0, rel_mode=0, outs=None, ldb=None, groups=None):
    """Backward of softmax(QK^T)V.  attn/pm: padded (B,H,T1,ld) tensors; q/k/v may be column slices of packed
    tensors; `outs` = (dq, dk, dv) views to write into (e.g. slices of a packed gradient), allocated when None.
    The three batched products behind the softmax backward are launched as two grids (dQ = dS K | dV = P^T dctx, dK = dS^T Q:
    one grid per operand-kind pair, K.launch_group_batched); groups = (kc, rc, keep): two lists the caller launches itself
    after adding its own products (relative-position attention: d qv, d pos), and a list that keeps the operands of the queued
    products alive until then (a descriptor holds raw pointers: a temporary freed before the launch is the next allocation)."""
    B, T1, D = q.shape
    T2 = k.shape[1]
    dk = D // H
    dtype = q.dtype
    dctx = _c(dctx) if dctx is not None else torch.zeros((B, T1, D), dtype=dtype, device=q.device)
    if outs is None:
        outs = (torch.empty((B, T1, D), dtype=dtype, device=q.device), torch.empty((B, T2, D), dtype=dtype, device=q.device),
                torch.empty((B, T2, D), dtype=dtype, device=q.device))
    dq, dkk, dv = outs
    if pm is None:      # forward ran the fused kernel (no dropped copy was stored): one launch, masks regenerated
        KAT.fused_bwd(q, k, v, dctx, attn, _pad_like(dattn, attn), H, scale, p, seed, dq, dkk, dv)
        return dq, dkk, dv, None
    gkc, grc, keep = groups if groups is not None else ([], [], [])
    dq_fused = False
    rel_ok = not Lp or (rel_mode == 1 and T1 == T2 and Lp == 2 * T1 - 1 and ldb is not None and ldb % 8 == 0)
    if rel_ok and attn.is_contiguous() and KAT.map_supported(dctx, v, H):
        # up to 512 keys, bf16: dP = dctx . v^T, the softmax backward and (rel-pos) the un-shift in ONE launch, dP never in memory,
        # no zero fill of dbd (csrc/attn_map.hip)
        # ... and, for d_k in {64, 96, 128}, dq = dS . k as the same launch's second product
        dq_fused = dq.stride(2) == 1 and KAT.map_product_ok(k, H)
        ds, dbd = KAT.map_bwd(dctx, v, attn, _pad_like(dattn, attn), H, scale, p, seed, ldb=ldb if Lp else 0,
                              k=k if dq_fused else None, dq=dq if dq_fused else None)
    else:
        # dP[b,h,i,j] = sum_d dctx[b,i,hd] v[b,j,hd]
        dp = _qk(dctx, v, B, H, T1, T2, dk, D, dtype)
        ds, dbd = K.attn_softmax_bwd(attn, dp, scale, p=p, seed=seed, Lp=Lp, rel_mode=rel_mode, dattn=_pad_like(dattn, attn), T2=T2,
                                     ldb=ldb)
    # dV[b,j,hd] = sum_i pm[b,h,i,j] dctx[b,i,hd]
    _into(dv, _pop(pm, T1, H, K.RC), _bop(dctx, dk, K.RC), T2, dk, T1, dk, dtype, B, H, group=grc)
    # dQ[b,i,hd] = sum_j dS[b,h,i,j] k[b,j,hd]
    if not dq_fused:
        _into(dq, _pop(ds, T1, H), _bop(k, dk, K.RC), T1, dk, T2, dk, dtype, B, H, group=gkc)
    # dK[b,j,hd] = sum_i dS[b,h,i,j] q[b,i,hd]
    _into(dkk, _pop(ds, T1, H, K.RC), _bop(q, dk, K.RC), T2, dk, T1, dk, dtype, B, H, group=grc)
    keep += [dctx, ds, dbd, pm, q, k, v]
    if groups is None:
        K.launch_group_batched(gkc)
        K.launch_group_batched(grc)
    return dq, dkk, dv, dbd


_FUSED = object()        # stands in for `pdrop` when the fused kernel ran: nothing but the attention map was stored


def _split_pdrop(ctx, pdrop):
    ctx.fused = pdrop is _FUSED
    return None if ctx.fused else pdrop


def _pm(ctx, attn, pdrop):
    """The (dropped) probabilities the unfused backward multiplies with; None = run the fused backward kernel."""
    return None if ctx.fused else (pdrop if pdrop is not None else attn)


def _attn_fwd_views(q, k, v, klen, causal, H, p):
    B, T1, D = q.shape
    T2 = k.shape[1]
    dk = D // H
    dtype = q.dtype
    scale = 1.0 / math.sqrt(dk)
    seed = K.new_seed(q.device) if p > 0.0 else (None, 0)
    if KAT.supported(q, k, v, H):       # short sequences, bf16: scores + mask + softmax + dropout + P.V in ONE launch
        out, attn = KAT.fused_fwd(q, k, v, klen, causal, H, scale, p, seed)
        return out, attn, _FUSED, scale, seed
    if KAT.map_supported(q, k, H):      # up to 512 keys, bf16: scores + mask + softmax + dropout in ONE launch (csrc/attn_map.hip)
        # ... and, for d_k in {64, 96, 128}, the context (dropped map) . v as the same launch's second product
        attn, pdrop, out = KAT.map_fwd(q, k, klen, causal, H, scale, p, seed, v=v if KAT.map_product_ok(v, H) else None)
        if out is not None:
            return out, attn, pdrop, scale, seed
    else:
        scores = _qk(q, k, B, H, T1, T2, dk, D, dtype)
        attn, pdrop = K.attn_softmax_fwd(scores, dtype, scale, klen=klen, causal=causal, p=p, seed=seed, T2=T2)
    out = _pv(pdrop if pdrop is not None else attn, v, B, H, T1, T2, dk, D, dtype)
    return out, attn, pdrop, scale, seed


def _user_attn(attn, T2):
    """The (B,H,T1,T2) view handed to callers (`self.attn`); the padded tensor stays the saved one."""
    return attn if attn.shape[-1] == T2 else attn[..., :T2]


class _AttnPackedQKV(Function):
    """Self-attention core on ONE packed projection qkv (B, T, 3D) (fused Q/K/V GEMM); returns the gradient packed."""

    @staticmethod
    def forward(ctx, qkv, klen, causal, H, p):
        qkv = _c(qkv)
        D = qkv.shape[-1] // 3
        q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
        out, attn, pdrop, scale, seed = _attn_fwd_views(q, k, v, klen, causal, H, p)
        ctx.meta = (H, scale, p, seed, D)
        ctx.save_for_backward(qkv, attn, _split_pdrop(ctx, pdrop))
        ctx.set_materialize_grads(False)
        return out, _user_attn(attn, k.shape[1])

    @staticmethod
    def backward(ctx, dctx, dattn):
        qkv, attn, pdrop = ctx.saved_tensors
        H, scale, p, seed, D = ctx.meta
        q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
        dqkv = torch.empty_like(qkv)
        _attn_common_bwd(dctx, dattn, attn, _pm(ctx, attn, pdrop), q, k, v, H, scale, p, seed,
                         outs=(dqkv[..., :D], dqkv[..., D:2 * D], dqkv[..., 2 * D:]))
        return dqkv, None, None, None, None


class _AttnPackedKV(Function):
    """Source-attention core: q (B, T1, D) and ONE packed projection kv (B, T2, 2D) of the memory."""

    @staticmethod
    def forward(ctx, q, kv, klen, causal, H, p):
        q = _c(q)
        D = q.shape[-1]
        # a column slice of the decoder's batched K/V projection (split_cols) is used in place when the kernels that take row and
        # batch strides run (the one-launch kernel of short sequences; round 6: the attention-map kernel of medium ones, whose
        # neighbours are GEMM descriptors with strides of their own); everything else wants a dense (B, T2, 2D) tensor
        in_place = kv.stride(-1) == 1 and (KAT.supported(q, kv[..., :D], kv[..., D:], H)
                                           or (KAT.map_supported(q, kv[..., :D], H) and KAT.view_ok(kv[..., D:])))
        if not in_place:
            kv = _c(kv)
        k, v = kv[..., :D], kv[..., D:]
        out, attn, pdrop, scale, seed = _attn_fwd_views(q, k, v, klen, causal, H, p)
        ctx.meta = (H, scale, p, seed, D)
        # a column block of split_cols whose gradient can be written in place (see _GradSink): on the same paths (their backward
        # kernels / descriptors take row and batch strides for dK and dV)
        ctx.gsink = getattr(kv, "_s2s_gsink", None) if in_place else None
        ctx.save_for_backward(q, kv, attn, _split_pdrop(ctx, pdrop))
        ctx.set_materialize_grads(False)
        return out, _user_attn(attn, k.shape[1])

    @staticmethod
    def backward(ctx, dctx, dattn):
        q, kv, attn, pdrop = ctx.saved_tensors
        H, scale, p, seed, D = ctx.meta
        k, v = kv[..., :D], kv[..., D:]
        dq = torch.empty_like(q)
        if ctx.gsink is not None and dctx is not None:
            dkv = ctx.gsink[0].part(ctx.gsink[1])       # this block of the packed gradient, written where split_cols wants it
        else:
            dkv = torch.empty(kv.shape, dtype=kv.dtype, device=kv.device)
        _attn_common_bwd(dctx, dattn, attn, _pm(ctx, attn, pdrop), q, k, v, H, scale, p, seed,
                         outs=(dq, dkv[..., :D], dkv[..., D:]))
        return dq, dkv, None, None, None, None


class _GradSink:
    """The gradient of a tensor that split_cols cut into column blocks, allocated once and written IN PLACE by the consumers of
    the blocks (the source-attention blocks of all decoder layers write dK | dV of their (B, T, 2D) block straight into the
    (B, T, L * 2D) gradient of the batched projection): the backward pass of split_cols then has nothing to concatenate (one
    18 us torch.cat of 18.6 MB on the VTN chain per step)."""
    __slots__ = ("shape", "dtype", "device", "W", "buf", "claimed")

    def __init__(self, shape, dtype, device, W):
        self.shape, self.dtype, self.device, self.W, self.buf, self.claimed = shape, dtype, device, W, None, set()

    def part(self, i):
        """Block i of the buffer for its ONE consumer.  A block that feeds a second consumer gets a fresh tensor instead: two
        consumers writing the same view would leave autograd summing two aliases of one buffer (2 g2 instead of g1 + g2); with
        the fresh tensor autograd's sum is a new tensor, holds() fails and _SplitCols concatenates the correct sums."""
        if i in self.claimed:
            return torch.empty(self.shape[:-1] + (self.W,), dtype=self.dtype, device=self.device)
        self.claimed.add(i)
        if self.buf is None:
            self.buf = torch.empty(self.shape, dtype=self.dtype, device=self.device)
        return self.buf[..., i * self.W:(i + 1) * self.W]

    def holds(self, grads):
        """True if `grads` are exactly the blocks of the buffer, in order (every consumer wrote its block in place)."""
        if self.buf is None:
            return False
        base, es = self.buf.data_ptr(), self.buf.element_size()
        return all(g is not None and g.data_ptr() == base + i * self.W * es and g.shape == self.buf.shape[:-1] + (self.W,)
                   and g.stride() == self.buf.stride() and g.dtype == self.dtype for i, g in enumerate(grads))


class _SplitCols(Function):
    """(..., n*W) -> n column blocks (..., W) as VIEWS (no copy); the backward pass returns the sink's buffer if every consumer wrote
    its block's gradient in place (_GradSink), else it concatenates the n gradients with one kernel.
    (Plain slicing would make autograd build n zero-padded full-width gradients and n-1 adds.)"""

    @staticmethod
    def forward(ctx, x, n, sink):
        W = x.shape[-1] // n
        ctx.n, ctx.W, ctx.sink = n, W, sink
        ctx.meta = (x.shape, x.dtype, x.device)
        ctx.set_materialize_grads(False)
        return tuple(x[..., i * W:(i + 1) * W] for i in range(n))

    @staticmethod
    def backward(ctx, *grads):
        shape, dtype, device = ctx.meta
        if all(g is None for g in grads):
            return None, None, None
        sink = ctx.sink
        if sink is not None and sink.holds(grads):
            buf, sink.buf = sink.buf, None
            sink.claimed.clear()
            return buf, None, None
        if sink is not None:
            sink.buf = None
            sink.claimed.clear()
        blk = shape[:-1] + (ctx.W,)
        parts = [g if g is not None else torch.zeros(blk, dtype=dtype, device=device) for g in grads]
        return torch.cat(parts, dim=-1), None, None


def split_cols(x, n):
    sink = _GradSink(tuple(x.shape), x.dtype, x.device, x.shape[-1] // n) if (x.requires_grad and x.is_contiguous()) else None
    parts = _SplitCols.apply(x, n, sink)
    if sink is not None:
        for i, t in enumerate(parts):
            t._s2s_gsink = (sink, i)
    return parts


def attention_packed_qkv(qkv, klen, causal, H, p=0.0):
    return _AttnPackedQKV.apply(qkv, klen, causal, H, p)


def attention_packed_kv(q, kv, klen, causal, H, p=0.0):
    return _AttnPackedKV.apply(q, kv, klen, causal, H, p)


class _AttnCore(Function):
    @staticmethod
    def forward(ctx, q, k, v, klen, causal, H, p):
        q, k, v = _c(q), _c(k), _c(v)
        out, attn, pdrop, scale, seed = _attn_fwd_views(q, k, v, klen, causal, H, p)
        ctx.meta = (H, scale, p, seed)
        ctx.save_for_backward(q, k, v, attn, _split_pdrop(ctx, pdrop))
        ctx.set_materialize_grads(False)
        return out, _user_attn(attn, k.shape[1])

    @staticmethod
    def backward(ctx, dctx, dattn):
        q, k, v, attn, pdrop = ctx.saved_tensors
        H, scale, p, seed = ctx.meta
        pm = _pm(ctx, attn, pdrop)
        dq, dkk, dv, _ = _attn_common_bwd(dctx, dattn, attn, pm, q, k, v, H, scale, p, seed)
        return dq, dkk, dv, None, None, None, None


def attention_core(q, k, v, klen, causal, H, p=0.0):
    """(context (B,T1,D), attn (B,H,T1,T2)); klen: int32 device tensor (B,) of valid key counts or None."""
    return _AttnCore.apply(q, k, v, klen, causal, H, p)


class _RelAttnCore(Function):
    """scores = (qu.k^T + rel_shift(qv.pos^T)) / sqrt(dk); rel_mode 1 = new (pos (1,2T-1,D)), 2 = legacy (pos (1,T,D))."""

    @staticmethod
    def forward(ctx, qu, qv, k, v, pos, klen, H, p, rel_mode):
        qu, qv, k, v, pos = _c(qu), _c(qv), _c(k), _c(v), _c(pos)
        B, T, D = qu.shape
        dk = D // H
        L = pos.shape[1]
        dtype = qu.dtype
        scale = 1.0 / math.sqrt(dk)
        seed = K.new_seed(qu.device) if p > 0.0 else (None, 0)
        ac = _qk(qu, k, B, H, T, T, dk, D, dtype)
        Lq = _pad8(L)                 # bd / dbd rows padded to 16 bytes (2T-1 is odd) so the backward GEMMs vectorise
        bd = torch.empty((B, H, T, Lq), dtype=torch.float32, device=qu.device)
        K.gemm(K.operand(qv, D, bs0=T * D, bs1=dk), K.operand(pos, D, bs0=0, bs1=dk), T, L, dk, bd, in_dtype=dtype, nb0=B, nb1=H,
               ldc=Lq, cbs=(H * T * Lq, T * Lq))
        attn, pdrop = K.attn_softmax_fwd(ac, dtype, scale, klen=klen, causal=False, bd=bd, rel_mode=rel_mode, p=p, seed=seed, T2=T,
                                         Lp=L)
        pm = pdrop if pdrop is not None else attn
        out = _pv(pm, v, B, H, T, T, dk, D, dtype)
        ctx.meta = (H, scale, p, seed, rel_mode, L)
        ctx.save_for_backward(qu, qv, k, v, pos, attn, pdrop)
        ctx.set_materialize_grads(False)
        return out, _user_attn(attn, T)

    @staticmethod
    def backward(ctx, dctx, dattn):
        qu, qv, k, v, pos, attn, pdrop = ctx.saved_tensors
        H, scale, p, seed, rel_mode, L = ctx.meta
        B, T, D = qu.shape
        dk = D // H
        dtype = qu.dtype
        pm = pdrop if pdrop is not None else attn
        Lq = _pad8(L)
        dqu, dkk, dv, dbd = _attn_common_bwd(dctx, dattn, attn, pm, qu, k, v, H, scale, p, seed, Lp=L, rel_mode=rel_mode, ldb=Lq)
        # dQv[b,i,hd] = sum_c dbd[b,h,i,c] pos[c,hd]
        dqv = torch.empty((B, T, D), dtype=dtype, device=qu.device)
        K.gemm(K.operand(dbd, Lq, bs0=H * T * Lq, bs1=T * Lq, zero_padded=True), K.operand(pos, D, layout=K.RC, bs0=0, bs1=dk), T, dk,
               L, dqv, in_dtype=dtype, nb0=B, nb1=H, ldc=D, cbs=(T * D, dk))
        # dPos[c,hd] = sum_{b,i} dbd[b,h,i,c] qv[b,i,hd]  (per-batch partials, then a deterministic sum over b)
        part = torch.empty((B, L, D), dtype=dtype, device=qu.device)
        K.gemm(K.operand(dbd, Lq, layout=K.RC, bs0=H * T * Lq, bs1=T * Lq, zero_padded=True),
               K.operand(qv, D, layout=K.RC, bs0=T * D, bs1=dk), L, dk, T, part, in_dtype=dtype, nb0=B, nb1=H, ldc=D, cbs=(L * D, dk))
        dpos, _ = K.colreduce(0, part.view(B, L * D))
        dpos = K.cast(dpos.view(1, L, D), dtype)
        return dqu, dqv, dkk, dv, dpos, None, None, None, None


def rel_attention_core(qu, qv, k, v, pos, klen, H, p=0.0, rel_mode=1):
    return _RelAttnCore.apply(qu, qv, k, v, pos, klen, H, p, rel_mode)


def _head_bias_grads(u, v, dqu, dqv):
    """d pos_bias_u = sum over (batch, time) of dQu, likewise v.  With flat-gradient slots the two reductions are queued and
    join the grouped column reductions of the batch (two launches for up to 24 of them) instead of six launches per layer."""
    if not u.requires_grad:
        return None, None
    D = dqu.shape[-1]
    a, b = dqu.view(-1, D), dqv.view(-1, D)
    if _slotted(u, v):
        _side_run(lambda: (_reduce_to(u, None, 0, a), _reduce_to(v, None, 0, b)), keep=(a, b))
        return None, None
    su, _ = K.colreduce(0, a)
    sv, _ = K.colreduce(0, b)
    return su.view(u.shape), sv.view(v.shape)


class _RelAttnPacked(Function):
    """Relative-position self-attention on ONE packed projection qkv (B, T, 3D) (the Q | K | V GEMM of the layer):
    qu = q + pos_bias_u, qv = q + pos_bias_v, then _RelAttnCore's arithmetic with k, v read in place as column blocks.
    The gradient comes back packed, so the three projections also share ONE data-gradient and ONE weight-gradient GEMM
    (K = 3D instead of three products and two additions)."""

    @staticmethod
    def forward(ctx, qkv, pos, u, v, klen, H, p, rel_mode):
        qkv, pos = _c(qkv), _c(pos)
        B, T, D3 = qkv.shape
        D = D3 // 3
        dk = D // H
        L = pos.shape[1]
        dtype = qkv.dtype
        q, k, vv = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
        scale = 1.0 / math.sqrt(dk)
        seed = K.new_seed(qkv.device) if p > 0.0 else (None, 0)
        ctx.fused_rel = KAT.rel_supported(q, k, vv, pos, H, rel_mode)
        if ctx.fused_rel:       # T <= 256, bf16: head bias + both score terms + shift + softmax + dropout in ONE launch (csrc/relattn.hip)
            attn, pdrop, qu, qv = KAT.rel_fwd(q, k, pos, u.detach().reshape(-1), v.detach().reshape(-1), klen, H, scale, p, seed)
            out = _pv(pdrop if pdrop is not None else attn, vv, B, H, T, T, dk, D, dtype)
            ctx.meta = (H, scale, p, seed, rel_mode, L, D)
            ctx.params = (u, v)
            ctx.save_for_backward(qkv, qu, qv, pos, attn, pdrop)
            ctx.set_materialize_grads(False)
            return out, _user_attn(attn, T)
        qu, qv = K.add_head_bias_view(q, u.detach().reshape(-1), v.detach().reshape(-1))
        ac = _qk(qu, k, B, H, T, T, dk, D, dtype)
        Lq = _pad8(L)
        bd = torch.empty((B, H, T, Lq), dtype=torch.float32, device=qkv.device)
        K.gemm(K.operand(qv, D, bs0=T * D, bs1=dk), K.operand(pos, D, bs0=0, bs1=dk), T, L, dk, bd, in_dtype=dtype, nb0=B, nb1=H,
               ldc=Lq, cbs=(H * T * Lq, T * Lq))
        attn, pdrop = K.attn_softmax_fwd(ac, dtype, scale, klen=klen, causal=False, bd=bd, rel_mode=rel_mode, p=p, seed=seed, T2=T,
                                         Lp=L)
        pm = pdrop if pdrop is not None else attn
        out = _pv(pm, vv, B, H, T, T, dk, D, dtype)
        ctx.meta = (H, scale, p, seed, rel_mode, L, D)
        ctx.params = (u, v)
        ctx.save_for_backward(qkv, qu, qv, pos, attn, pdrop)
        ctx.set_materialize_grads(False)
        return out, _user_attn(attn, T)

    @staticmethod
    def backward(ctx, dctx, dattn):
        qkv, qu, qv, pos, attn, pdrop = ctx.saved_tensors
        H, scale, p, seed, rel_mode, L, D = ctx.meta
        u, v = ctx.params
        B, T, _ = qu.shape
        dk = D // H
        dtype = qu.dtype
        k, vv = qkv[..., D:2 * D], qkv[..., 2 * D:]
        pm = pdrop if pdrop is not None else attn
        Lq = _pad8(L)
        dqkv = torch.empty_like(qkv)
        dqu = torch.empty((B, T, D), dtype=dtype, device=qu.device)
        gkc, grc, keep = [], [], []      # the five batched products behind the softmax backward: two grids (K.launch_group_batched)
        _, _, _, dbd = _attn_common_bwd(dctx, dattn, attn, pm, qu, k, vv, H, scale, p, seed, Lp=L, rel_mode=rel_mode, ldb=Lq,
                                        outs=(dqu, dqkv[..., D:2 * D], dqkv[..., 2 * D:]), groups=(gkc, grc, keep))
        dqv = torch.empty((B, T, D), dtype=dtype, device=qu.device)
        K.gemm(K.operand(dbd, Lq, bs0=H * T * Lq, bs1=T * Lq, zero_padded=True), K.operand(pos, D, layout=K.RC, bs0=0, bs1=dk), T, dk,
               L, dqv, in_dtype=dtype, nb0=B, nb1=H, ldc=D, cbs=(T * D, dk), group=gkc)
        part = torch.empty((B, L, D), dtype=dtype, device=qu.device)
        K.gemm(K.operand(dbd, Lq, layout=K.RC, bs0=H * T * Lq, bs1=T * Lq, zero_padded=True),
               K.operand(qv, D, layout=K.RC, bs0=T * D, bs1=dk), L, dk, T, part, in_dtype=dtype, nb0=B, nb1=H, ldc=D, cbs=(L * D, dk),
               group=grc)
        K.launch_group_batched(gkc)
        K.launch_group_batched(grc)
        del keep
        dpos, _ = K.colreduce(0, part.view(B, L * D))
        dpos = K.cast(dpos.view(1, L, D), dtype)
        K.add_rows(dqu, dqv, dqkv[..., :D])
        du, dv = _head_bias_grads(u, v, dqu, dqv)
        return dqkv, dpos, du, dv, None, None, None, None


def rel_attention_packed(qkv, pos, u, v, klen, H, p=0.0, rel_mode=1):
    return _RelAttnPacked.apply(qkv, pos, u, v, klen, H, p, rel_mode)


class _HeadBias(Function):
    """qu = q + pos_bias_u, qv = q + pos_bias_v   (attention.py:283-286)."""

    @staticmethod
    def forward(ctx, q, u, v):
        q = _c(q)
        ctx.params = (u, v)
        return K.add_head_bias(q, u.detach().reshape(-1), v.detach().reshape(-1))

    @staticmethod
    def backward(ctx, dqu, dqv):
        u, v = ctx.params
        dqu, dqv = _c(dqu), _c(dqv)
        dq = K.axpby(1.0, dqu, 1.0, dqv)
        du, dv = _head_bias_grads(u, v, dqu, dqv)
        return dq, du, dv


def add_head_bias(q, u, v):
    return _HeadBias.apply(q, u, v)


# ================================================================================================
# Conv1d (stride 1, 'same' padding, channel-last activations) as an implicit GEMM
# reference: Postnet / AlignmentModule / DurationPredictor / FFN Conv1d call sites
# ================================================================================================


class _Conv1d(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, act):
        x = _c(x)
        B, T, Cin = x.shape
        Cout, _, ks = weight.shape
        pad = (ks - 1) // 2
        dtype = x.dtype
        wp = K.gather3_cached(weight, (Cout, ks, Cin), (Cin * ks, 1, ks), 0, dtype)  # (O, k, I)
        y = torch.empty((B, T, Cout), dtype=dtype, device=x.device)
        K.gemm(K.operand(x, Cin, mode=K.CONV1D, C=Cin, T=T, pad=pad), K.operand(wp, ks * Cin), B * T, Cout, ks * Cin, y,
               in_dtype=dtype, bias=bias, act=act)
        ctx.act = act
        ctx.params = (weight, bias)
        ctx.save_for_backward(x, y if act else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        weight, bias = ctx.params
        B, T, Cin = x.shape
        Cout, _, ks = weight.shape
        pad = (ks - 1) // 2
        dtype = x.dtype
        dy = _c(dy)
        if ctx.act:
            dy = K.act_dropout_bwd(dy, y, act=ctx.act)
        dx = None
        if ctx.needs_input_grad[0]:
            wd = K.gather3_cached(weight, (Cin, ks, Cout), (ks, -1, Cin * ks), ks - 1, dtype)  # (I, k flipped, O)
            dx = torch.empty((B, T, Cin), dtype=dtype, device=x.device)
            K.gemm(K.operand(dy, Cout, mode=K.CONV1D, C=Cout, T=T, pad=pad), K.operand(wd, ks * Cout), B * T, Cin, ks * Cout, dx,
                   in_dtype=dtype)
        dw = db = None
        if weight.requires_grad:
            def work():
                dwp = torch.empty((Cout, ks * Cin), dtype=torch.float32, device=x.device)
                rs, racc, dbv = _bias_sink(bias, Cout)
                tile, sk = K.plan_gemm(Cout, ks * Cin, B * T)
                # wgrad=True: big bf16 outputs (the aligner's 1536 x 4608 over 4096 frames) run on the ragged 8-wave weight-gradient kernel
                # with the implicit im2col B operand (csrc/gemm_8ph.hip "w8_conv" kind 1); everything else as before
                K.gemm(K.operand(dy, Cout, layout=K.RC), K.operand(x, Cin, layout=K.RC, mode=K.CONV1D, C=Cin, T=T, pad=pad),
                       Cout, ks * Cin, B * T, dwp, in_dtype=dtype, splitk=sk, tile=tile, a_rowsum=rs, a_rowsum_accumulate=racc,
                       wgrad=True)
                return _emit_permuted(weight, dwp, Cout, ks, Cin), dbv
            if _slotted(weight, bias):
                _side_run(work, keep=(dy, x))
            else:
                dw, db = work()
        elif bias is not None and bias.requires_grad:
            db, _ = _reduce_to(bias, None, 0, dy.view(B * T, Cout))
        return dx, dw, db, None


class _CropRows(Function):
    """Rows t >= vlens[b] of a (B, T, C) tensor are ABSENT (a captured training step on a batch shorter than its padded shape,
    modules.Lens.crop): zero on the way in -- what a 'same'-padded convolution behind this op reads where the reference's cropped
    tensor ends (models/vtn.py:208-214) -- and zero on the way back, so that no gradient leaves a frame the reference does not have."""

    @staticmethod
    def forward(ctx, x, vlens):
        from . import kernels_sdp as KS
        ctx.vlens = vlens
        return KS.mask_rows(_c(x), vlens)

    @staticmethod
    def backward(ctx, dy):
        from . import kernels_sdp as KS
        return KS.mask_rows(_c(dy), ctx.vlens), None


def crop_rows(x, vlens):
    """Identity when vlens is None (every row present)."""
    return x if vlens is None else _CropRows.apply(x, vlens)


def conv1d(x, weight, bias=None, act=None, vlens=None):
    """x (B,T,Cin) channel-last; weight (Cout,Cin,k) torch layout; returns (B,T,Cout).
    vlens (B int32, device; captured steps only): input frames t >= vlens[b] are absent -- read as zero padding (k > 1), and no
    gradient flows back into them.  The OUTPUT frames beyond vlens are not cleared: whoever mixes along time next does that."""
    if vlens is not None and weight.shape[-1] > 1:
        x = crop_rows(x, vlens)
    return _Conv1d.apply(x, weight, bias, act)


# ================================================================================================
# Conv2d 3x3 stride 2 (+ReLU) on NHWC activations         reference: subsampling.py:58-63
# ================================================================================================
#                                                                    (189 vs 190 us), so the plain launches stay the default




class _Conv2dS2(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, grad_premasked=False, input_is_relu=False):
        x = _c(x)
        ctx.grad_premasked = grad_premasked
        ctx.input_is_relu = input_is_relu
        B, T1, F1, C = x.shape
        O = weight.shape[0]
        T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
        dtype = x.dtype
        wp = K.gather3_cached(weight, (O, 9, C), (C * 9, 1, 9), 0, dtype)  # (O, tap, C)
        y = torch.empty((B, T2, F2, O), dtype=dtype, device=x.device)
        K.gemm(K.operand(x, C, mode=K.CONV2D_S2, C=C, T1=T1, F1=F1, T2=T2, F2=F2), K.operand(wp, 9 * C), B * T2 * F2, O, 9 * C, y,
               in_dtype=dtype, bias=bias, act="relu")
        ctx.params = (weight, bias)
        ctx.save_for_backward(x, y, wp)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, wp = ctx.saved_tensors
        weight, bias = ctx.params
        B, T1, F1, C = x.shape
        O = weight.shape[0]
        T2, F2 = y.shape[1], y.shape[2]
        M2 = B * T2 * F2
        dtype = x.dtype
        # the consumer may already have applied relu' in its dgrad epilogue (linear_fc_permuted(..., input_is_relu=True))
        dy = _c(dy) if ctx.grad_premasked else K.act_dropout_bwd(_c(dy), y, act="relu")
        dw = db = None
        if weight.requires_grad:
            def work():
                dwp = torch.empty((O, 9 * C), dtype=torch.float32, device=x.device)
                rs, racc, dbv = _bias_sink(bias, O)
                tile, sk = K.plan_gemm(O, 9 * C, M2)
                # bf16, C % 128 == 0: the ragged 8-wave weight-gradient kernel with the implicit im2col B operand (wgrad=True;
                # csrc/gemm_8ph.hip "w8_conv"), otherwise the 4-wave split-K kernel
                K.gemm(K.operand(dy, O, layout=K.RC),
                       K.operand(x, C, layout=K.RC, mode=K.CONV2D_S2, C=C, T1=T1, F1=F1, T2=T2, F2=F2), O, 9 * C, M2, dwp,
                       in_dtype=dtype, splitk=sk, tile=tile, a_rowsum=rs, a_rowsum_accumulate=racc, wgrad=True)
                return _emit_permuted(weight, dwp, O, 9, C), dbv
            if _slotted(weight, bias):
                # queued AND forked before the data-gradient GEMM below: this is the last big layer of the backward pass,
                # its weight gradient would otherwise run alone after the main chain has ended.  (Sending the batch queued so
                # far to ANOTHER stream first -- so that its grouped weight gradients do not wait behind this 150-280 us GEMM --
                # was measured: 4.24 vs 4.20 ms per VTN step, the extra concurrent work slows the transposed convolution on the
                # main stream, which is the critical path.  Issuing it AFTER the data-gradient GEMMs instead: 4.33 ms.)
                _side_run(work, keep=(dy, x))
                _side_flush()
            else:
                dw, db = work()
        dx = None
        if ctx.needs_input_grad[0] and dtype == torch.bfloat16 and O % 8 == 0 and O >= 64 and C % 8 == 0:
            # transposed convolution as four implicit GEMMs, one per parity class (t1 % 2, f1 % 2) of input pixels: each
            # gathers its 4 / 2 / 2 / 1 taps of dY straight from HBM and stores into its pixels of dX (no dcols, no col2im)
            wts = K.tconv2d_weights(weight.detach())
            dx = torch.empty((B, T1, F1, C), dtype=dtype, device=x.device)
            mask = x if ctx.input_is_relu else None     # x = relu(u): the GEMMs hand back dL/du (mask rows follow the c_map)
            for cls, wt in enumerate(wts):
                pt, pf = cls >> 1, cls & 1
                Tc, Fc = (T1 - pt + 1) // 2, (F1 - pf + 1) // 2
                if Tc <= 0 or Fc <= 0:
                    continue
                Kc = wt.shape[1]
                K.gemm(K.operand(dy, O, mode=K.TCONV2D_S2, C=O, T1=Tc, F1=Fc, T2=T2, F2=F2, pad=cls), K.operand(wt, Kc),
                       B * Tc * Fc, C, Kc, dx, in_dtype=dtype, c_map=(T1, F1, Tc, Fc, pt, pf), emask=mask)
        elif ctx.needs_input_grad[0]:
            dcols = torch.empty((M2, 9 * C), dtype=dtype, device=x.device)
            K.gemm(K.operand(dy, O), K.operand(wp, 9 * C, layout=K.RC), M2, 9 * C, O, dcols, in_dtype=dtype)
            dx = K.col2im_s2(dcols, B, T1, F1, C, T2, F2)
            if ctx.input_is_relu:
                dx = K.act_dropout_bwd(dx, x, act="relu")
        if not weight.requires_grad and bias is not None and bias.requires_grad:
            db, _ = _reduce_to(bias, None, 0, dy.view(M2, O))
        return dx, dw, db, None, None


def conv2d_s2_relu(x_nhwc, weight, bias, grad_premasked=False, input_is_relu=False):
    """grad_premasked=True: the ONLY consumer of the result hands back a gradient that already carries relu'
    (see linear_fc_permuted(input_is_relu=True)); the backward then skips its own mask pass.
    input_is_relu=True: x_nhwc is a ReLU output whose producer wants dL/d(pre-activation) back (conv_in1_relu(grad_premasked=True)):
    relu'(x) is applied in the epilogue of the data-gradient GEMMs."""
    return _Conv2dS2.apply(x_nhwc, weight, bias, grad_premasked, input_is_relu)


class _ConvIn1(Function):
    """Conv2d(1 -> O, 3x3, stride 2) + ReLU straight on the (B, T, F) mel batch (subsampling.py:58-60)."""

    @staticmethod
    def forward(ctx, x, weight, bias, grad_premasked=False):
        x = _c(x)
        y = K.conv_in1_fwd(x, weight.detach(), bias.detach() if bias is not None else None)
        ctx.params = (weight, bias)
        ctx.grad_premasked = grad_premasked
        ctx.save_for_backward(x, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        weight, bias = ctx.params
        # relu' is applied inside the wgrad kernel (y > 0: no mask pass over 60 M values) -- unless the consumer's data-gradient
        # GEMM already did it in its epilogue (conv2d_s2_relu(input_is_relu=True)): then the kernel reads dy only (122 instead of
        # 244 MB at VTN's shapes -- this kernel is the very end of the backward pass)
        dy = _c(dy)
        ym = None if ctx.grad_premasked else y
        wslot, bslot = getattr(weight, "_s2s_grad", None), getattr(bias, "_s2s_grad", None) if bias is not None else None
        if wslot is not None and (bias is None or bslot is not None):
            # its own fork: the last closure of the backward pass must not queue behind the batch it would otherwise end
            _side_run(lambda: K.conv_in1_wgrad(x, dy, wslot, bslot, True, y=ym), keep=(x, dy, y), solo=True)
            return None, None, None, None
        dw = torch.empty(weight.shape, dtype=torch.float32, device=x.device)
        db = torch.empty(bias.shape, dtype=torch.float32, device=x.device) if bias is not None else None
        K.conv_in1_wgrad(x, dy, dw, db, False, y=ym)
        return None, dw, db, None


def conv_in1_relu(x, weight, bias, grad_premasked=False):
    """grad_premasked=True: the ONLY consumer hands back a gradient that already carries relu' of this layer's output."""
    return _ConvIn1.apply(x, weight, bias, grad_premasked)




class _LinearPermuted(Function):
    """y = x2d . Wp^T + b where Wp[d, f*C + c] = W[d, c*F + f]: the Linear after the conv2d front-end
    (subsampling.py:64-70 flattens (c, f); our activations are (f, c) channel-last)."""

    @staticmethod
    def forward(ctx, x, weight, bias, C, Fd, input_is_relu=False):
        x = _c(x)
        ctx.input_is_relu = input_is_relu
        dtype = x.dtype
        D = weight.shape[0]
        M = x.numel() // (C * Fd)
        wp = K.gather3_cached(weight, (D, Fd, C), (C * Fd, 1, Fd), 0, dtype)
        y = torch.empty((M, D), dtype=dtype, device=x.device)
        K.gemm(K.operand(x, C * Fd), K.operand(wp, C * Fd), M, D, C * Fd, y, in_dtype=dtype, bias=bias)
        ctx.params = (weight, bias)
        ctx.meta = (C, Fd)
        ctx.save_for_backward(x, wp)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wp = ctx.saved_tensors
        weight, bias = ctx.params
        C, Fd = ctx.meta
        D = weight.shape[0]
        Kd = C * Fd
        M = x.numel() // Kd
        dtype = x.dtype
        dy = _c(dy)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((M, Kd), dtype=dtype, device=x.device)
            mask = x.view(M, Kd) if ctx.input_is_relu else None              # x = relu(u): hand back dL/du
            if dtype == torch.bfloat16:
                # the permuted weight once more, TRANSPOSED ((f, c) rows of D values: a cached copy the optimiser keeps fresh):
                # both operands K-contiguous, so the 2016 x 7296 x 384 product runs on the 8-wave kernel instead of the 4-wave
                # kernel's transposing fragment reads (65 -> ~25 us at the end of VTN's backward chain)
                wd = K.gather3_cached(weight, (Fd, C, D), (1, Fd, C * Fd), 0, dtype)
                K.gemm(K.operand(dy, D), K.operand(wd, D), M, Kd, D, dx, in_dtype=dtype, emask=mask)
            else:
                K.gemm(K.operand(dy, D), K.operand(wp, Kd, layout=K.RC), M, Kd, D, dx, in_dtype=dtype, emask=mask)
            dx = dx.view(x.shape)
        dw = db = None
        if weight.requires_grad:
            def work():
                dwp = torch.empty((D, Kd), dtype=torch.float32, device=x.device)
                rs, racc, dbv = _bias_sink(bias, D)
                tile, sk = K.plan_gemm(D, Kd, M)
                K.gemm(K.operand(dy, D, layout=K.RC), K.operand(x, Kd, layout=K.RC), D, Kd, M, dwp, in_dtype=dtype,
                       splitk=sk, tile=tile, a_rowsum=rs, a_rowsum_accumulate=racc, wgrad=True)
                return _emit_permuted(weight, dwp, D, Fd, C), dbv
            if _slotted(weight, bias):
                _side_run(work, keep=(dy, x))
            else:
                dw, db = work()
        elif bias is not None and bias.requires_grad:
            db, _ = _reduce_to(bias, None, 0, dy)
        return dx, dw, db, None, None, None


def linear_fc_permuted(x, weight, bias, C, Fd, input_is_relu=False):
    return _LinearPermuted.apply(x, weight, bias, C, Fd, input_is_relu)


# ================================================================================================
# BatchNorm1d over (B*T) rows, channel-last, fused with activation + dropout
# reference: pre_postnets.py:108-165, conformer/convolution.py:73-75
# ================================================================================================
class _BatchNormAct(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, run_mean, run_var, num_batches, training, act, p, eps, momentum, vlens=None):
        x = _c(x)
        C = x.shape[-1]
        rows = x.numel() // C
        seed = K.new_seed(x.device) if p > 0.0 else (None, 0)
        if not training:
            vlens = None
        Tn = x.shape[-2] if vlens is not None else 0       # absent rows (include/s2svc_hip.h): out of the statistics, zero in y and dx
        ctx.vl = (vlens, Tn)
        ctx.vec = training and K.bn_vec_ok(x)
        if ctx.vec:     # bf16: 16-byte kernels; the backward pass recomputes the activation / dropout derivative (csrc/convmod.hip)
            mean, rstd = K.bn_stats_vec(x, rows, C, eps, momentum, run_mean, run_var, num_batches, vlens=vlens, Tn=Tn)
            need_pre = act in ("swish", "gelu")
            y, pre = K.bn_act_apply_vec(x, mean, rstd, gamma.detach(), beta.detach(), act=act, p=p, seed=seed, want_pre=need_pre,
                                        vlens=vlens, Tn=Tn)
            ctx.meta = (training, act, p, seed)
            ctx.params = (gamma, beta)
            ctx.save_for_backward(x, mean, rstd, (pre if need_pre else y) if (act or p > 0.0) else None)
            return y
        if training:
            if x.dtype == torch.bfloat16:      # one pass over x: mean and E[x^2] together (fp32 sums of bf16 values)
                mean, rstd = K.bn_stats(x, rows, C, eps, momentum, run_mean, run_var, num_batches, vlens=vlens, Tn=Tn)
            else:                              # parity mode: the two-pass variance torch computes
                sc = 1.0 / rows if vlens is None else -1.0          # < 0: 1 / (number of present rows), on the device
                mean, _ = K.colreduce(0, x.view(rows, C), scale=sc, vlens=vlens, Tn=Tn)
                var, _ = K.colreduce(3, None, x=x.view(rows, C), mean=mean, scale=sc, rows=rows, D=C, vlens=vlens, Tn=Tn)
                rstd = K.bn_finalize(mean, var, rows, eps, momentum, run_mean, run_var, num_batches, vlens=vlens, Tn=Tn)
        else:
            mean = run_mean
            rstd = K.rstd_from_var(run_var, eps)
        need_pre = act in ("swish", "gelu")
        y, pre = K.bn_apply(x, mean, rstd, gamma, beta, act=act, p=p, seed=seed, want_pre=need_pre, vlens=vlens, Tn=Tn)
        ctx.meta = (training, act, p, seed)
        ctx.params = (gamma, beta)
        ctx.save_for_backward(x, mean, rstd, pre if need_pre else y)
        return y

    @staticmethod
    def backward(ctx, dz):
        x, mean, rstd, saved = ctx.saved_tensors
        gamma, beta = ctx.params
        training, act, p, seed = ctx.meta
        C = x.shape[-1]
        rows = x.numel() // C
        dy = _c(dz)
        vlens, Tn = ctx.vl
        if ctx.vec:
            g_slot = getattr(gamma, "_s2s_grad", None) if gamma.requires_grad else None
            b_slot = getattr(beta, "_s2s_grad", None) if beta.requires_grad else None
            both = g_slot is not None and b_slot is not None
            dx, sdy, sdyx = K.bn_act_bwd_vec(dy, saved, x, mean, rstd, gamma.detach(), act=act, p=p, seed=seed,
                                             dgamma_acc=g_slot.view(-1) if both else None, dbeta_acc=b_slot.view(-1) if both else None,
                                             vlens=vlens, Tn=Tn)
            dgamma = dbeta = None
            if gamma.requires_grad and not both:
                dgamma, dbeta = _emit_vgrad(gamma, sdyx), _emit_vgrad(beta, sdy)
            return dx, dgamma, dbeta, None, None, None, None, None, None, None, None, None
        if act or p > 0.0:
            dy = K.act_dropout_bwd(dy, saved, act=act, p=p, seed=seed)
        sdy, sdyx = K.colreduce(2, dy.view(rows, C), x.view(rows, C), mean, rstd, want_dot=True, vlens=vlens, Tn=Tn)
        dx = K.bn_bwd(dy, x, mean, rstd, gamma, sdy, sdyx, use_batch_stats=training, vlens=vlens, Tn=Tn)
        dgamma = dbeta = None
        if gamma.requires_grad:
            if _slotted(gamma, beta):      # the two accumulations into the gradient slots leave the data-gradient chain
                _side_run(lambda: (_emit_vgrad(gamma, sdyx), _emit_vgrad(beta, sdy)), keep=(sdy, sdyx))
            else:
                dgamma, dbeta = _emit_vgrad(gamma, sdyx), _emit_vgrad(beta, sdy)
        return dx, dgamma, dbeta, None, None, None, None, None, None, None, None, None


def batch_norm_act(x, gamma, beta, run_mean, run_var, num_batches, training, act=None, p=0.0, eps=1e-5, momentum=0.1, vlens=None):
    """vlens (B int32, device; x must be (B, T, C)): frames t >= vlens[b] are absent -- not in the batch statistics, zero in the
    output and in the data gradient (captured steps on batches shorter than their padded shape; include/s2svc_hip.h)."""
    return _BatchNormAct.apply(x, gamma, beta, run_mean, run_var, num_batches, training, act, p, eps, momentum, vlens)


# ================================================================================================
# losses
# ================================================================================================
_CONST_SCALAR = {}


def const_scalar(device, value=0.0):
    """A constant 0-dim fp32 tensor per (device, value): made once, outside any capture (the first step of a shape runs eagerly),
    so that roots of backward passes and absent loss terms cost no fill launch inside a captured step."""
    device = torch.device(device)
    key = (device.type, device.index, float(value))
    if key not in _CONST_SCALAR:
        _CONST_SCALAR[key] = torch.full((), float(value), dtype=torch.float32, device=device)
    return _CONST_SCALAR[key]


def _zero_scalar(device):
    return const_scalar(device, 0.0)


def root_backward(loss, scale=1.0, retain_graph=False):
    """(scale * loss).backward() with the root gradient handed in as a cached constant: autograd's implicit ones_like and a
    `loss * scale` would each be an ATen launch inside a captured step."""
    if loss.dim() == 0 and loss.dtype == torch.float32:
        loss.backward(gradient=const_scalar(loss.device, scale), retain_graph=retain_graph)
    else:
        (loss * scale if scale != 1.0 else loss).backward(retain_graph=retain_graph)


class _WeightedSum(Function):
    """sum_i w_i * sum(x_i) of fp32 tensors (scalars or small vectors) as ONE launch each way: the training loss from its parts
    (trainers/ar_vc.py:86-97, trainers/aas_vc.py:100-139) without a chain of 0-dim ATen adds / muls and their autograd nodes."""

    @staticmethod
    def forward(ctx, weights, *xs):
        xs = [_c(x.float()) for x in xs]
        ctx.meta = ([tuple(x.shape) for x in xs], list(weights))
        return K.weighted_sum(list(zip(xs, weights)))

    @staticmethod
    def backward(ctx, g):
        shapes, weights = ctx.meta
        outs = K.weighted_sum_bwd(_c(g.float()), list(zip(shapes, weights)), g.device)
        return (None,) + tuple(outs)


def weighted_sum(terms):
    """terms: [(tensor, weight)] (at most 8) -> 0-dim fp32 tensor sum_i weight_i * tensor_i.sum()."""
    return _WeightedSum.apply(tuple(float(w) for _, w in terms), *[t for t, _ in terms])


class _SeqLoss(Function):
    @staticmethod
    def forward(ctx, after, before, logits, ys, labels, olens_i32, pos_weight):
        after = _c(after) if after is not None else None
        before = _c(before)
        logits = _c(logits) if logits is not None else None
        ys = _c(ys.float())
        labels = _c(labels.float()) if labels is not None else None
        out = K.seq_loss_fwd(after, before, logits, ys, labels, olens_i32, pos_weight)
        ctx.pos_weight = pos_weight
        ctx.save_for_backward(after, before, logits, ys, labels, olens_i32, out)
        ctx.set_materialize_grads(False)          # an unused loss (bce of the L1-only criterion) arrives as None, not as a zeros launch
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_l1, g_bce):
        after, before, logits, ys, labels, olens_i32, out = ctx.saved_tensors
        g1 = _c(g_l1.float()) if g_l1 is not None else None
        g2 = _c(g_bce.float()) if g_bce is not None else None
        if g1 is None:
            g1 = _zero_scalar(before.device)
        if g2 is None:
            g2 = _zero_scalar(before.device)
        da, db, dl = K.seq_loss_bwd(after, before, logits, ys, labels, olens_i32, ctx.pos_weight, out, g1, g2)
        return da, db, dl, None, None, None, None


def seq2seq_loss(after, before, logits, ys, labels, olens_i32, pos_weight=10.0):
    """(l1, bce): losses/seq2seq_loss.py:30-59; `logits`/`labels` None -> l1 only (losses/l1_loss.py)."""
    return _SeqLoss.apply(after, before, logits, ys, labels, olens_i32, pos_weight)


class _GuidedAttnLoss(Function):
    @staticmethod
    def forward(ctx, att, ilens_i32, olens_i32, sigma, alpha):
        att = _c(att)
        out = K.guided_attn_loss_fwd(att, ilens_i32, olens_i32, sigma, alpha)
        ctx.meta = (att.shape, att.dtype, sigma, alpha)
        ctx.save_for_backward(ilens_i32, olens_i32, out)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        ilens_i32, olens_i32, out = ctx.saved_tensors
        shape, dtype, sigma, alpha = ctx.meta
        return K.guided_attn_loss_bwd(shape, dtype, g.device, ilens_i32, olens_i32, sigma, alpha, out, _c(g.float())), None, None, None, None


def guided_attention_loss(att, ilens_i32, olens_i32, sigma=0.4, alpha=1.0):
    return _GuidedAttnLoss.apply(att, ilens_i32, olens_i32, sigma, alpha)


# ================================================================================================
# monotonic alignment search + binarisation loss     reference: modules/alignments.py:281-310
# ================================================================================================
class _Viterbi(Function):
    @staticmethod
    def forward(ctx, log_p_attn, text_lens_i32, feat_lens_i32):
        lp = _c(log_p_attn.float())
        ds, path, binmean = K.mas(lp, text_lens_i32, feat_lens_i32)
        B = lp.shape[0]
        bin_loss = K.weighted_sum([(binmean, -1.0 / B)])          # -(sum_b mean_t log p[b, t, path]) / B, one launch
        ctx.save_for_backward(path, feat_lens_i32)
        ctx.shape = lp.shape
        ctx.in_dtype = log_p_attn.dtype
        ctx.mark_non_differentiable(ds, path)
        ctx.set_materialize_grads(False)
        return ds, bin_loss, path

    @staticmethod
    def backward(ctx, _dds, g, _dpath):
        if g is None:
            return None, None, None
        path, feat_lens_i32 = ctx.saved_tensors
        dlogp = K.zeros(ctx.shape, torch.float32, g.device)
        K.mas_binloss_bwd(path, feat_lens_i32, _c(g.float()), dlogp)
        return dlogp.to(ctx.in_dtype), None, None


def viterbi_decode(log_p_attn, text_lens_i32, feat_lens_i32):
    """-> (ds (B,T_text) fp32, bin_loss scalar (differentiable), path (B,T_feats) int32)."""
    return _Viterbi.apply(log_p_attn, text_lens_i32, feat_lens_i32)
