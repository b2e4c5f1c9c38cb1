"""Thin, non-differentiable Python launchers over the C ABI (include/s2svc_hip.h).

Every function here takes torch tensors that already live on the GPU, hands their raw device
pointers to libs2svc_hip.so on torch's current HIP stream, and returns torch tensors.  torch is
used for memory (caching allocator) and streams only.
"""
import ctypes
import math
import os

import torch

from .. import _lib

KC, RC = 0, 1
DENSE, CONV1D, CONV2D_S2, TCONV2D_S2 = 0, 1, 2, 3
ACT = {None: 0, "none": 0, "relu": 1, "tanh": 2, "swish": 3, "sigmoid": 4, "gelu": 5}

_DT = {torch.float32: 0, torch.bfloat16: 1}


def dt(t):
    try:
        return _DT[t.dtype if isinstance(t, torch.Tensor) else t]
    except KeyError:
        raise TypeError(f"unsupported dtype {t.dtype if isinstance(t, torch.Tensor) else t}")


def ptr(t):
    return None if t is None else t.data_ptr()


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_GET_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


_RAW_STREAM_OK = None        # None: not validated yet; True / False after the first call


def _validate_raw_stream():
    """The private accessors are used only after ONE check against the public API on a non-default stream (a torch release that
    changes their signature or meaning falls back to the public path instead of handing launchers a wrong stream)."""
    global _RAW_STREAM_OK
    ok = False
    if _RAW_STREAM is not None and _GET_DEVICE is not None:
        try:
            side = torch.cuda.Stream()
            with torch.cuda.stream(side):
                ok = int(_RAW_STREAM(_GET_DEVICE())) == int(torch.cuda.current_stream().cuda_stream) == int(side.cuda_stream)
            ok = ok and int(_RAW_STREAM(_GET_DEVICE())) == int(torch.cuda.current_stream().cuda_stream)
        except Exception:       # noqa: BLE001 -- any failure of the private path means: do not use it
            ok = False
    _RAW_STREAM_OK = ok
    return ok


def stream():
    """The raw hipStream_t of torch's current stream on the current device.  torch.cuda.current_stream() costs ~10 us of host time
    per call (device-index resolution through torch.cuda.is_available() -> device count); the raw accessor is what it ends in.
    Every launcher calls this: ~160 times per eager VTN step."""
    if _RAW_STREAM_OK or (_RAW_STREAM_OK is None and _validate_raw_stream()):
        return _RAW_STREAM(_GET_DEVICE())
    return torch.cuda.current_stream().cuda_stream


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("seq2seq_vc_amd kernels need GPU tensors (there is no CPU path)")


# ----------------------------------------------------------------------------------------------
# dropout seed state: kernels read (*seed_base + seed_off); the base lives in device memory so a
# captured hipGraph draws fresh masks on every replay once the trainer bumps it.
# ----------------------------------------------------------------------------------------------
class _SeedState:
    def __init__(self):
        self.base = {}
        self.counter = 0
        self.init = 0x5EED5EED

    def tensor(self, device):
        key = (device.type, device.index)
        if key not in self.base:
            self.base[key] = torch.full((1,), self.init, dtype=torch.int64, device=device)
        return self.base[key]

    def next_offset(self):
        self.counter += 1
        return (self.counter * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF


SEED = _SeedState()


def manual_seed(seed, device=None):
    SEED.init = int(seed)
    SEED.counter = 0
    for t in SEED.base.values():
        t.fill_(int(seed))


def advance_seed(device):
    """Bump the device-resident seed base (graph-capturable)."""
    t = SEED.tensor(device)
    if t.is_cuda:
        _lib.check(_lib.lib().s2svc_seed_advance(t.data_ptr(), 0x10001, stream()), "seed_advance")
    else:
        t.add_(0x10001)


# ----------------------------------------------------------------------------------------------
# scalar / index glue of a step (csrc/glue.hip): no ATen kernel inside a captured step
# ----------------------------------------------------------------------------------------------
def zero_(t):
    """t.zero_() as a launch of this library (t contiguous)."""
    if not t.is_contiguous():
        raise ValueError("zero_: contiguous tensors only")
    _lib.check(_lib.lib().s2svc_fill_zero(t.data_ptr(), t.numel() * t.element_size(), stream()), "fill_zero")
    return t


def zeros(shape, dtype, device):
    return zero_(torch.empty(shape, dtype=dtype, device=device))


def _terms(pairs):
    """pairs: [(fp32 contiguous tensor, weight)] -> ScalarTerms (the tensors must outlive the launch: callers keep them)."""
    if not 1 <= len(pairs) <= 8:
        raise ValueError("1 to 8 terms")
    t = _lib.ScalarTerms()
    for i, (x, w) in enumerate(pairs):
        if x.dtype != torch.float32 or not x.is_contiguous():
            raise TypeError("scalar terms are contiguous fp32 tensors")
        t.x[i], t.n[i], t.w[i] = x.data_ptr(), x.numel(), float(w)
    t.k = len(pairs)
    return t


def weighted_sum(pairs, out=None):
    """sum_i w_i * sum(x_i) -> 0-dim fp32 tensor, one launch (fixed order)."""
    out = torch.empty((), dtype=torch.float32, device=pairs[0][0].device) if out is None else out
    t = _terms(pairs)
    _lib.check(_lib.lib().s2svc_weighted_sum(ctypes.byref(t), ptr(out), stream()), "weighted_sum")
    return out


def weighted_sum_bwd(g, shapes_weights, device):
    """-> [w_i * g broadcast to shape_i]: ONE buffer, one launch; the results are views of it."""
    sizes = [int(math.prod(s)) for s, _ in shapes_weights]
    buf = torch.empty(max(1, sum(sizes)), dtype=torch.float32, device=device)
    outs, off = [], 0
    for (s, _), n in zip(shapes_weights, sizes):
        outs.append(buf[off:off + n].view(s))
        off += n
    t = _terms([(o.reshape(-1), w) for o, (_, w) in zip(outs, shapes_weights)])
    _lib.check(_lib.lib().s2svc_weighted_sum_bwd(ctypes.byref(t), ptr(g), stream()), "weighted_sum_bwd")
    return outs


def scalars_axpy(pairs, acc, beta=1.0):
    """acc[i] = beta * acc[i] + w_i * sum(x_i)   (acc: fp32, at least len(pairs) elements)."""
    t = _terms(pairs)
    _lib.check(_lib.lib().s2svc_scalars_axpy(ctypes.byref(t), float(beta), ptr(acc), stream()), "scalars_axpy")
    return acc


def pad_cols(x2, ldo):
    """(rows, N) -> (rows, ldo) with zero columns behind N, one launch."""
    rows, N = x2.shape
    out = torch.empty((rows, ldo), dtype=x2.dtype, device=x2.device)
    _lib.check(_lib.lib().s2svc_pad_cols(dt(x2), rows, N, ldo, ptr(x2), ptr(out), stream()), "pad_cols")
    return out


def decoder_input(ys, r, out_dtype):
    """Teacher-forcing input (models/vtn.py:236-243): cat(zeros, ys[:, r-1::r][:, :-1]) in out_dtype, one launch.  ys (B, T, D) fp32."""
    B, T, D = ys.shape
    if ys.dtype != torch.float32 or ys.stride(2) != 1 or ys.stride(1) != D:
        raise TypeError("decoder_input: fp32 (B, T, D) with contiguous frames")
    Tin = T // r
    out = torch.empty((B, Tin, D), dtype=out_dtype, device=ys.device)
    _lib.check(_lib.lib().s2svc_decoder_input(_DT[out_dtype], B, Tin, r, D, ys.stride(0), ptr(ys), ptr(out), stream()), "decoder_input")
    return out


def append_eos(xs, lens_i32, eos, pad):
    """(B, T) int64 tokens -> (B, T + 1) with eos behind each sequence (models/transformer_tts.py:139-142), one launch."""
    B, T = xs.shape
    if xs.dtype != torch.int64 or xs.stride(1) != 1:
        raise TypeError("append_eos: int64 tokens with contiguous rows")
    out = torch.empty((B, T + 1), dtype=torch.int64, device=xs.device)
    _lib.check(_lib.lib().s2svc_append_eos(B, T, xs.stride(0), ptr(xs), ptr(lens_i32), int(eos), int(pad), ptr(out), stream()), "append_eos")
    return out


def dense_rows(x):
    """x.contiguous() for a tensor whose last dim is dense and whose leading dims share one row stride (a column block of a wider
    matrix, e.g. one layer's K|V out of the batched projection), as a launch of this library; other layouts: torch."""
    if x.is_contiguous():
        return x
    D = x.shape[-1]
    lds = x.stride(-2) if x.dim() >= 2 else D
    ok = x.is_cuda and x.dtype in _DT and x.stride(-1) == 1 and x.dim() >= 2
    for d in range(x.dim() - 2):                 # leading dims must be laid out as one run of rows
        ok = ok and x.stride(d) == x.stride(d + 1) * x.shape[d + 1]
    if not ok:
        return x.contiguous()
    out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    _lib.check(_lib.lib().s2svc_copy_rows(dt(x), x.numel() // D, D, lds, ptr(x), ptr(out), stream()), "copy_rows")
    return out


def stop_labels(labels, lens_i32, T):
    """labels[:, :T] with a 1 at frame lens[b] - 1 (models/vtn.py:253-260), one launch."""
    B = labels.shape[0]
    if labels.dtype != torch.float32 or labels.stride(1) != 1:
        raise TypeError("stop_labels: fp32 labels with contiguous rows")
    out = torch.empty((B, T), dtype=torch.float32, device=labels.device)
    _lib.check(_lib.lib().s2svc_stop_labels(B, T, labels.stride(0), ptr(labels), ptr(lens_i32), ptr(out), stream()), "stop_labels")
    return out


def launch_floor(sink, workgroups=256, threads=256):
    """One launch of the do-nothing kernel (s2svc_launch_floor): `sink` = 1024 int32 words of device memory."""
    _lib.check(_lib.lib().s2svc_launch_floor(int(workgroups), int(threads), ptr(sink), stream()), "s2svc_launch_floor")


def reset_op_counter():
    SEED.counter = 0


def new_seed(device):
    return SEED.tensor(device).data_ptr(), SEED.next_offset()


# ----------------------------------------------------------------------------------------------
# GEMM
# ----------------------------------------------------------------------------------------------
def operand(t, ld, layout=KC, mode=DENSE, C=0, T=0, pad=0, T1=0, F1=0, T2=0, F2=0, bs0=0, bs1=0, offset=0, zero_padded=False):
    o = _lib.Operand()
    o.ptr = t.data_ptr() + offset * t.element_size()
    o.ld, o.layout, o.mode, o.C, o.T, o.pad = ld, layout, mode, C, T, pad
    o.T1, o.F1, o.T2, o.F2, o.bs0, o.bs1 = T1, F1, T2, F2, bs0, bs1
    o.zero_padded = 1 if zero_padded else 0
    return o




def plan_gemm(M, N, K, nbatch=1, allow_split=True, dtype=None):
    """Tiny cost model -> (tile, splitk) for the MFMA GEMM: estimated time = waves of resident workgroups x
    (k-tiles per workgroup x time per k-tile + fixed), plus the split-K reduction.  Numbers are microseconds
    fitted to MI355X measurements of this kernel (profiles/).
    fp32 problems with few output tiles (the duration predictor's 384 x 384 weight gradients over 1024 rows): 32 x 32 tiles
    (gemm_fast_kernel<float, 32, 32, 128>: the fp32 MFMA makes a 64 x 64 workgroup matrix-pipe bound) and only as much split-K
    as it takes to reach ~100 workgroups -- 144 unsplit workgroups instead of 36 x 4 + a reduction launch."""
    if dtype == torch.float32 and nbatch == 1 and K >= 256 and ((M + 63) // 64) * ((N + 63) // 64) < 64:
        tiles32, sk = ((M + 31) // 32) * ((N + 31) // 32), 1
        while allow_split and tiles32 * sk < 96 and K // (2 * sk) >= 256 and sk < 8:
            sk *= 2
        return 32, sk
    if M >= 128 and N >= 128 and ((M + 127) // 128) * ((N + 127) // 128) * nbatch >= 256:
        return 128, 1          # a full wave of 128x128 tiles: measured best for every operand layout (tools/gemm_bench.py --sweep)
    best = None
    for tile, bk, t_tile, resident in ((64, 128, 1.6, 3 * 256), (128, 64, 2.2, 2 * 256)):
        if tile == 128 and (M < 128 or N < 128):
            continue
        tiles = ((M + tile - 1) // tile) * ((N + tile - 1) // tile) * nbatch
        ktiles = (K + bk - 1) // bk
        for s in ((1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64) if allow_split else (1,)):
            if s > 1 and ktiles // s < 2:
                break
            per = (ktiles + s - 1) // s
            waves = (tiles * s + resident - 1) // resident
            cost = waves * (per * t_tile + 3.0)
            if s > 1:
                cost += 4.0 + M * N * nbatch * (s + 1) * 4 / 3.0e6     # reduce kernel: launch + ws traffic at ~3 TB/s
            if best is None or cost < best[0]:
                best = (cost, tile, s)
    return best[1], best[2]


def pick_splitk(M, N, K, nbatch=1):
    return plan_gemm(M, N, K, nbatch)[1]


# ----------------------------------------------------------------------------------------------
# Hand-off points inside a captured graph (s2svc_event_* of the C ABI): see distributed.OverlappedBackward.mark
# ----------------------------------------------------------------------------------------------
class GraphMark:
    """An event that, recorded on a CAPTURING stream, becomes an event-record node of the graph: every replay records it when the work
    captured before it has run, and `wait(stream)` issued after the replay's launch makes `stream` wait for that point of that
    replay (torch.cuda.Event(external=True) is refused on ROCm builds of torch; HIP itself supports it)."""

    def __init__(self):
        h = ctypes.c_void_p()
        _lib.check(_lib.lib().s2svc_event_create(ctypes.byref(h)), "event_create")
        self.handle = h.value

    def record(self, stream_handle=None):
        """-> True if the record became a graph node (the stream is capturing)."""
        rc = _lib.lib().s2svc_event_record(self.handle, stream() if stream_handle is None else stream_handle)
        if rc < 0:
            _lib.check(rc, "event_record")
        return rc == 1

    def wait(self, torch_stream):
        _lib.check(_lib.lib().s2svc_stream_wait_event(torch_stream.cuda_stream, self.handle), "stream_wait_event")

    def __del__(self):
        try:
            if self.handle:
                _lib.lib().s2svc_event_destroy(self.handle)
        except Exception:       # noqa: BLE001 -- interpreter shutdown
            pass


# ----------------------------------------------------------------------------------------------
# Writer audit (S2SVC_AUDIT_SLOTS=1 / audit_slots(True); tests).  Gradient slots are accumulated into by several kernels of a
# backward pass (weight-gradient GEMMs, fused bias row sums, column reductions) that run on the issuing stream, on side streams
# or as background launches.  Two accumulating writers of ONE slot on DIFFERENT streams with no wait between them are a race that
# eager launches hide (the GPU happens to keep issue order) and a captured graph exposes (only edges order its branches) --
# the signature of round 3's "second side stream" experiment, whose gradient norm moved under capture only.  With the audit on,
# every accumulating launch records (slot pointer -> stream); a second writer from another stream raises unless a join
# (ops.functional.side_join) came in between.
# ----------------------------------------------------------------------------------------------
class _Audit:
    on = os.environ.get("S2SVC_AUDIT_SLOTS", "0") == "1"
    writers = {}          # slot pointer -> (stream handle, what)
    checked = 0


def audit_slots(on=True):
    _Audit.on = bool(on)
    _Audit.writers = {}
    _Audit.checked = 0


def audit_reset(ptrs=None):
    """A join: every slot (or the listed ones) may be written from any stream again."""
    if ptrs is None:
        _Audit.writers = {}
    else:
        for p in ptrs:
            _Audit.writers.pop(p, None)


def _audit_write(what, *ptrs, st=None):
    if not _Audit.on:
        return
    st = stream() if st is None else st
    for p in ptrs:
        if not p:
            continue
        _Audit.checked += 1
        prev = _Audit.writers.get(p)
        if prev is not None and prev[0] != st:
            raise RuntimeError(f"gradient-slot race: {what} accumulates into {p:#x} on stream {st:#x} while {prev[1]} wrote it on "
                               f"stream {prev[0]:#x} with no join in between")
        _Audit.writers[p] = (st, what)


def gemm(A, B, M, N, K, out, *, in_dtype, bias=None, act=None, res=None, ldr=None, alpha=1.0, nb0=1, nb1=1,
         ldc=None, cbs=(0, 0), rbs=(0, 0), out_offset=0, splitk=1, accumulate=False, a_rowsum=None,
         a_rowsum_accumulate=False, tile=0, emask=None, drop_p=0.0, seed=(None, 0), c_map=None, group=None, pre_out=None,
         emask_mode=0, wgrad=False):
    """C = act(alpha * A.B^T + bias) [* dropmask] [* (emask > 0)] + res   (see s2svc_gemm in include/s2svc_hip.h).
    c_map = (T1, F1, Tc, Fc, pt, pf): GEMM row (b, i, j) of the Tc x Fc class grid goes to row (b, 2i+pt, 2j+pf) of C.
    group = list: the descriptor is appended instead of launched (launch_group() runs the list as one grid).
    pre_out: a tensor like `out` that receives alpha * A.B^T + bias BEFORE the activation (Swish backward needs it);
    emask_mode = 1: the stage multiplies by swish'(emask) (emask = that pre-activation) instead of masking by emask > 0.
    wgrad=True: a weight gradient into a TEMPORARY (no accumulation) takes the ragged 8-wave weight-gradient kernel as well
    when its descriptor is eligible (dense row-contiguous bf16 operands, fp32 C) -- launched at once, never queued."""
    d = _lib.GemmDesc()
    if pre_out is not None:
        if c_map is not None or pre_out.dtype != out.dtype or pre_out.shape != out.shape or not pre_out.is_contiguous():
            raise ValueError("gemm: pre_out must be a contiguous tensor like `out` (and no c_map)")
        d.c_pre = pre_out.data_ptr()
    if c_map is not None:
        d.c_map = 1
        d.cm_T1, d.cm_F1, d.cm_Tc, d.cm_Fc, d.cm_pt, d.cm_pf = c_map
    d.A, d.B = A, B
    d.C = out.data_ptr() + out_offset * out.element_size()
    d.ldc = N if ldc is None else ldc
    d.cbs0, d.cbs1 = cbs
    d.c_dtype = dt(out)
    d.bias = ptr(bias)
    d.res = ptr(res)
    d.ldr = (N if ldr is None else ldr)
    d.rbs0, d.rbs1 = rbs
    d.M, d.N, d.K, d.nb0, d.nb1 = M, N, K, nb0, nb1
    d.act = ACT[act]
    d.alpha = alpha
    d.dtype = _DT[in_dtype]
    d.accumulate = 1 if accumulate else 0
    d.splitk = splitk
    d.tile_hint = tile
    if emask is not None or drop_p > 0.0:
        if nb0 * nb1 != 1 or d.ldc != N:
            raise ValueError("gemm: the dropout / mask epilogue needs an unbatched, contiguous (M, N) output")
        if emask is not None:
            if emask.dtype != out.dtype:
                raise TypeError("gemm: emask must have the output dtype")
            d.emask, d.ldm, d.emask_mode = emask.data_ptr(), N, int(emask_mode)
        d.drop_p, d.seed_base, d.seed_off = drop_p, seed[0], seed[1]
    if group is not None:
        d.splitk, d.ws = 1, None
        group.append(d)
        return out
    # queue it for the grouped launch (unsplit) if the kernel takes it -- only problems that ACCUMULATE into their output (a
    # flat-gradient slot, final once the batch has been flushed).  A GEMM into a fresh temporary is consumed by the caller's
    # next launch: deferring it made the permuted weight gradient of the Linear behind the Conv2d front-end (gather3 +
    # accumulate, ops.functional._LinearPermuted) read its buffer before the GEMM had run -- found by the full-size gradient
    # test of round 3 (tests/gpu_model_check.py: vtn_full_size_grads); steady-state training had been using the buffer's
    # previous contents, i.e. the gradient of the step before.
    if _RECORDER is not None and accumulate:
        d.splitk, d.ws = 1, None
        if a_rowsum is not None:
            d.a_rowsum = a_rowsum.data_ptr()
            d.a_rowsum_accumulate = 1 if a_rowsum_accumulate else 0
        # small outputs only: a problem with >= _GROUP_MAX_TILES 128x128 tiles fills the chip on its own -- except exact multiples
        # of the 8-wave kernel's 256 x 256 tile (AAS-VC's 1536 / 3072 / 4608-feature layers): those are ~1.3x faster per flop on
        # that tile, which needs the tiles of several problems to fill the chip (108 for the largest one alone); grouping all of
        # them: AAS-VC step 14.57 -> 14.33 ms
        tiles8 = M % 256 == 0 and N % 256 == 0 and K % 64 == 0
        if (tiles8 or ((M + 127) // 128) * ((N + 127) // 128) < _GROUP_MAX_TILES) and _lib.lib().s2svc_gemm_grouped_ok(ctypes.addressof(d)):
            _RECORDER.append(d)
            return out
        d.splitk, d.a_rowsum = splitk, None
    if (accumulate or wgrad) and in_dtype == torch.bfloat16 and group is None:
        # a weight gradient launched on its own (no batch is being recorded: immediate mode): the same kernel, chunking and sums as in a
        # grouped launch (csrc/gemm_8ph.hip "W8") -- WHICH kernel a problem gets depends on its descriptor only
        d.splitk, d.ws = 1, None
        if a_rowsum is not None:
            d.a_rowsum = a_rowsum.data_ptr()
            d.a_rowsum_accumulate = 1 if a_rowsum_accumulate else 0
        if _lib.lib().s2svc_gemm_wgrad_ok(ctypes.addressof(d)):
            if _Audit.on:
                _audit_write("weight gradient (W8)", d.C, d.a_rowsum if a_rowsum_accumulate else None)
            launch_wgrad_group([d], capped=accumulate)
            return out
        d.splitk, d.a_rowsum = splitk, None
    ws = None
    if splitk > 1:
        ws = torch.empty(splitk * nb0 * nb1 * M * N, dtype=torch.float32, device=out.device)
        d.ws = ws.data_ptr()
    else:
        d.ws = None
    rs_ws = None
    if a_rowsum is not None:
        d.a_rowsum = a_rowsum.data_ptr()
        d.a_rowsum_accumulate = 1 if a_rowsum_accumulate else 0
        if splitk > 1:
            rs_ws = torch.empty(splitk * M, dtype=torch.float32, device=out.device)
            d.a_rowsum_ws = rs_ws.data_ptr()
    if bias is not None and bias.dtype != torch.float32:
        raise TypeError("gemm bias must be fp32")
    if res is not None and res.dtype != out.dtype:
        raise TypeError("gemm residual must have the output dtype")
    if _Audit.on and (accumulate or a_rowsum_accumulate):
        _audit_write("gemm", d.C if accumulate else None, d.a_rowsum if a_rowsum_accumulate else None)
    _lib.check(_lib.lib().s2svc_gemm(ctypes.byref(d), stream()), "s2svc_gemm")
    return out


# ----------------------------------------------------------------------------------------------
# Grouped weight-gradient GEMMs: while a recorder is active, eligible GEMMs (bf16, both operands dense and
# row-contiguous, i.e. dW = dY^T X) are queued instead of launched; flush_grouped() runs the queue as ONE grid per
# <= 11 problems (s2svc_gemm_grouped).  The caller keeps the operand tensors alive until the flush.
# ----------------------------------------------------------------------------------------------
_RECORDER = None            # list of GemmDesc while recording
_GROUP_TILE = 64             # output tile edge of the grouped kernel for small outputs
_GROUP_MAX_TILES = 200
_GROUP_BIG_TILES = 64


class record_grouped:
    """with record_grouped(queue): K.gemm(...) appends eligible problems to `queue` (list) instead of launching."""

    def __init__(self, queue):
        self.queue = queue

    def __enter__(self):
        global _RECORDER
        self.prev, _RECORDER = _RECORDER, self.queue
        return self.queue

    def __exit__(self, *exc):
        global _RECORDER
        _RECORDER = self.prev
        return False


_CR_RECORDER = None         # list of (ColreduceItem, keepalive) while recording


class record_colreduce:
    """with record_colreduce(queue): accumulating K.colreduce(...) calls are queued instead of launched."""

    def __init__(self, queue):
        self.queue = queue

    def __enter__(self):
        global _CR_RECORDER
        self.prev, _CR_RECORDER = _CR_RECORDER, self.queue
        return self.queue

    def __exit__(self, *exc):
        global _CR_RECORDER
        _CR_RECORDER = self.prev
        return False


def flush_colreduce(queue):
    """Launch the queued column reductions on the current stream, two launches per <= 24 of them; reductions into the same
    output go to successive launches.  Empties the queue."""
    pending = list(queue)
    del queue[:]
    while pending:
        seen, group, rest = set(), [], []
        for it, keep in pending:
            keys = {it.out_sum} | ({it.out_dot} if it.out_dot else set())
            if keys & seen:
                rest.append((it, keep))
            else:
                seen |= keys
                group.append((it, keep))
        pending = rest
        arr = (_lib.ColreduceItem * len(group))(*[g[0] for g in group])
        if _Audit.on:
            _audit_write("grouped column reduction", *[g[0].out_sum for g in group], *[g[0].out_dot for g in group])
        _lib.check(_lib.lib().s2svc_colreduce_grouped(ctypes.addressof(arr), len(group), stream()), "s2svc_colreduce_grouped")


def launch_group(descs, tile=128):
    """One grid for the listed problems (K.gemm(..., group=descs)): the parity-class GEMMs of a transposed convolution."""
    arr = (_lib.GemmDesc * len(descs))(*descs)
    _lib.check(_lib.lib().s2svc_gemm_grouped(ctypes.addressof(arr), len(descs), tile, stream()), "s2svc_gemm_grouped")


_W8_CAP = [0]


def set_wgrad_cap(wgs):
    """Workgroups of a grouped weight-gradient launch on the current stream (0 = one per unit).  ops.functional.enable_side_streams
    sets 64 for FORKED gradient batches (VTN / TTS): the launch runs beside the latency-bound data-gradient chain, and a 144 KB-LDS
    workgroup on every CU makes each small kernel of the chain wait for one -- measured 3.789 -> 3.773 ms per VTN step (five
    interleaved repeats; 48 / 96 / 128 workgroups: 3.778 / 3.774 / 3.780).  Units are walked in order: bit-identical results."""
    _W8_CAP[0] = max(0, int(wgs))


def get_wgrad_cap():
    return _W8_CAP[0]


def launch_wgrad_group(descs, capped=True):
    """The listed weight-gradient problems (every one s2svc_gemm_wgrad_ok) on the ragged 8-wave kernel, one grid per <= 40 of
    them (+ one reduction launch when a reduction is long enough to be cut into chunks: its partial tiles go through `ws`).
    capped=False: a chip-filling problem launched on its own (the front-end's Conv2d / Linear weight gradients) ignores the cap."""
    L = _lib.lib()
    arr = (_lib.GemmDesc * len(descs))(*descs)
    nws = L.s2svc_gemm_wgrad_ws_floats(ctypes.addressof(arr), len(descs))
    ws = torch.empty(nws, dtype=torch.float32, device=torch.cuda.current_device()) if nws else None
    if capped and _W8_CAP[0] > 0:        # forked gradient batches: a capped grid leaves CUs to the chain the launch runs beside (set_wgrad_cap)
        _lib.check(L.s2svc_gemm_wgrad_grouped_bg(ctypes.addressof(arr), len(descs), ptr(ws), stream(), _W8_CAP[0]),
                   "s2svc_gemm_wgrad_grouped_bg")
        return
    _lib.check(L.s2svc_gemm_wgrad_grouped(ctypes.addressof(arr), len(descs), ptr(ws), stream()), "s2svc_gemm_wgrad_grouped")


def launch_group_batched(descs):
    """The listed BATCHED problems (K.gemm(..., group=descs)) of one operand-kind pair as one grid (s2svc_gemm_grouped_batched:
    the batched products of an attention backward pass); one by one if the library does not take them as a group."""
    if not descs:
        return
    if len(descs) > 1:
        arr = (_lib.GemmDesc * len(descs))(*descs)
        rc = _lib.lib().s2svc_gemm_grouped_batched(ctypes.addressof(arr), len(descs), stream())
        if rc == 0:
            return
        if rc != 1:
            _lib.check(rc, "s2svc_gemm_grouped_batched")
    for d in descs:
        _lib.check(_lib.lib().s2svc_gemm(ctypes.byref(d), stream()), "s2svc_gemm")


def flush_grouped(queue):
    """Launch every queued problem on the current stream; problems that would write the same C / a_rowsum concurrently
    go to successive launches.  Empties the queue."""
    pending = list(queue)
    del queue[:]
    while pending:
        seen, group, rest = set(), [], []
        for d in pending:
            keys = {d.C} | ({d.a_rowsum} if d.a_rowsum else set())
            if keys & seen:
                rest.append(d)
            else:
                seen |= keys
                group.append(d)
        pending = rest
        # ragged dense weight gradients (any M, N % 8 == 0: VTN's 384 / 1152 / 1536-feature layers): the 8-wave kernel on (problem,
        # K chunk, 256 x 128 tile) units (csrc/gemm_8ph.hip "W8"); WHICH kernel a problem gets depends on its descriptor only
        L = _lib.lib()
        w8 = [d for d in group if L.s2svc_gemm_wgrad_ok(ctypes.addressof(d))]
        if w8:
            group = [d for d in group if not any(d is w for w in w8)]
            if _Audit.on:
                _audit_write("grouped weight gradient (W8)", *[d.C for d in w8], *[d.a_rowsum for d in w8])
            launch_wgrad_group(w8)
        # big outputs (>= _GROUP_BIG_TILES 128x128 tiles each) share launches of 128x128 tiles, the rest of 64x64 tiles
        big = [d for d in group if _GROUP_TILE == 64 and ((d.M + 127) // 128) * ((d.N + 127) // 128) >= _GROUP_BIG_TILES]
        small = [d for d in group if not any(d is b for b in big)]
        for part, tile in ((big, 128), (small, _GROUP_TILE)):
            if part:
                arr = (_lib.GemmDesc * len(part))(*part)
                if _Audit.on:
                    _audit_write("grouped weight gradient", *[d.C for d in part], *[d.a_rowsum for d in part])
                _lib.check(_lib.lib().s2svc_gemm_grouped(ctypes.addressof(arr), len(part), tile, stream()), "s2svc_gemm_grouped")


# ----------------------------------------------------------------------------------------------
# norms / reductions
# ----------------------------------------------------------------------------------------------
def layernorm_fwd(x, gamma, beta, eps, res=None, p=0.0, seed=(None, 0), need_stats=True, hscale=1.0):
    _need_cuda(x)
    D = x.shape[-1]
    rows = x.numel() // D
    y = torch.empty_like(x)
    s = torch.empty_like(x) if res is not None else None
    mean = torch.empty(rows, dtype=torch.float32, device=x.device) if need_stats else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if need_stats else None
    _lib.check(_lib.lib().s2svc_layernorm_fwd(dt(x), rows, D, ptr(x), ptr(res), p, hscale, seed[0], seed[1], ptr(gamma), ptr(beta),
                                              eps, ptr(y), ptr(s), ptr(mean), ptr(rstd), stream()), "layernorm_fwd")
    return y, s, mean, rstd


def layernorm_bwd(dy, s, mean, rstd, gamma, ds_extra=None, p=0.0, seed=(None, 0), want_dh=False, hscale=1.0, want_partials=False,
                  partials_ok=True):
    """-> (ds, dh) or, with want_partials, (ds, dh, partials): partials = (ws, chunks) when the launch also wrote the first
    reduction stage of the parameter gradients (s2svc_layernorm_bwd_pg: big bf16 sites), else None (reduce dy / s separately)."""
    D = s.shape[-1]
    rows = s.numel() // D
    ds = torch.empty_like(s)
    dh = torch.empty_like(s) if want_dh else None
    L = _lib.lib()
    if want_partials and partials_ok:
        chunks = L.s2svc_layernorm_bwd_pg_chunks(dt(s), rows, D, ptr(dy), ptr(s), ptr(gamma), ptr(ds_extra), ptr(ds), ptr(dh))
        if chunks > 0:
            ws = torch.empty(chunks * 2 * D, dtype=torch.float32, device=s.device)
            _lib.check(L.s2svc_layernorm_bwd_pg(dt(s), rows, D, ptr(dy), ptr(s), ptr(mean), ptr(rstd), ptr(gamma), ptr(ds_extra), p,
                                                hscale, seed[0], seed[1], ptr(ds), ptr(dh), ptr(ws), stream()), "layernorm_bwd_pg")
            return ds, dh, (ws, chunks)
    _lib.check(L.s2svc_layernorm_bwd(dt(s), rows, D, ptr(dy), ptr(s), ptr(mean), ptr(rstd), ptr(gamma),
                                     ptr(ds_extra), p, hscale, seed[0], seed[1], ptr(ds), ptr(dh), stream()),
               "layernorm_bwd")
    return (ds, dh, None) if want_partials else (ds, dh)


def colreduce_partials(ws, chunks, D, out_sum, out_dot):
    """Second reduction stage for partial row pairs ws[chunks][2][D] that a producer kernel already wrote (layernorm_bwd with
    want_partials): out_sum / out_dot (fp32 gradient slots) += sum over the chunks.  Queued with the batch's grouped column
    reductions when a recorder is active, else a grouped launch of its own."""
    it = _lib.ColreduceItem()
    it.dy, it.x, it.mean, it.rstd = None, None, None, None
    it.out_sum, it.out_dot, it.ws = ptr(out_sum), ptr(out_dot), ptr(ws)
    it.dtype, it.rows, it.D, it.mode, it.accumulate, it.ws_chunks, it.scale = _DT[torch.float32], chunks, D, 7, 1, chunks, 1.0
    if _CR_RECORDER is not None:
        _CR_RECORDER.append((it, (out_sum, out_dot, ws)))
        return
    if _Audit.on:
        _audit_write("column reduction", ptr(out_sum), ptr(out_dot))
    arr = (_lib.ColreduceItem * 1)(it)
    _lib.check(_lib.lib().s2svc_colreduce_grouped(ctypes.addressof(arr), 1, stream()), "colreduce_grouped")


_WS_CHUNKS = 64


def colreduce(mode, dy, x=None, mean=None, rstd=None, scale=1.0, want_dot=False, rows=None, D=None, out_sum=None,
              out_dot=None, accumulate=False, vlens=None, Tn=0):
    """Deterministic column reduction; `out_sum`/`out_dot` (fp32, D) may be given (e.g. flat-gradient slots,
    with accumulate=True) instead of being allocated.  vlens (B int32, device) with Tn: rows b * Tn + t, t >= vlens[b], are absent
    (include/s2svc_hip.h); scale < 0 then means 1 / (number of present rows)."""
    t = dy if dy is not None else x
    D = t.shape[-1] if D is None else D
    rows = t.numel() // D if rows is None else rows
    if _CR_RECORDER is not None and out_sum is not None and accumulate and vlens is None:
        # parameter-gradient reduction into a flat-gradient slot: queued for the grouped launch (flush_colreduce)
        ws = torch.empty(_WS_CHUNKS * 2 * D, dtype=torch.float32, device=t.device)
        it = _lib.ColreduceItem()
        it.dy, it.x, it.mean, it.rstd = ptr(dy), ptr(x), ptr(mean), ptr(rstd)
        it.out_sum, it.out_dot, it.ws = ptr(out_sum), ptr(out_dot), ptr(ws)
        it.dtype, it.rows, it.D, it.mode, it.accumulate, it.ws_chunks, it.scale = dt(t), rows, D, mode, 1, _WS_CHUNKS, scale
        _CR_RECORDER.append((it, (dy, x, mean, rstd, out_sum, out_dot, ws)))
        return out_sum, out_dot
    if out_sum is None:
        out_sum = torch.empty(D, dtype=torch.float32, device=t.device)
    if out_dot is None and want_dot:
        out_dot = torch.empty(D, dtype=torch.float32, device=t.device)
    ws = torch.empty(_WS_CHUNKS * 2 * D, dtype=torch.float32, device=t.device)
    if _Audit.on and accumulate:
        _audit_write("column reduction", ptr(out_sum), ptr(out_dot))
    _lib.check(_lib.lib().s2svc_colreduce(dt(t), rows, D, mode, ptr(dy), ptr(x), ptr(mean), ptr(rstd), scale, ptr(out_sum),
                                          ptr(out_dot), 1 if accumulate else 0, ptr(ws), _WS_CHUNKS, Tn, ptr(vlens), stream()), "colreduce")
    return out_sum, out_dot


def bn_finalize(mean, var, n, eps, momentum, run_mean=None, run_var=None, num_batches=None, var_is_ex2=False, vlens=None, Tn=0):
    """var_is_ex2: `var` holds E[x^2] (one-pass statistics, colreduce mode 6); the variance is E[x^2] - mean^2."""
    C = mean.numel()
    rstd = torch.empty_like(mean)
    _lib.check(_lib.lib().s2svc_bn_finalize(C, n, eps, momentum, ptr(mean), ptr(var), ptr(rstd), ptr(run_mean),
                                            ptr(run_var), ptr(num_batches), 1 if var_is_ex2 else 0, Tn, ptr(vlens), stream()), "bn_finalize")
    return rstd


def bn_stats(x, rows, C, eps, momentum, run_mean=None, run_var=None, num_batches=None, vlens=None, Tn=0):
    """(mean, rstd) of training-mode BatchNorm over (rows, C), one pass over x, two launches (s2svc_bn_stats)."""
    mean = torch.empty(C, dtype=torch.float32, device=x.device)
    rstd = torch.empty(C, dtype=torch.float32, device=x.device)
    ws = torch.empty(_WS_CHUNKS * 2 * C, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().s2svc_bn_stats(dt(x), rows, C, ptr(x), eps, momentum, ptr(mean), ptr(rstd), ptr(run_mean), ptr(run_var),
                                         ptr(num_batches), ptr(ws), _WS_CHUNKS, Tn, ptr(vlens), stream()), "bn_stats")
    return mean, rstd


def rstd_from_var(var, eps):
    rstd = torch.empty_like(var)
    _lib.check(_lib.lib().s2svc_rstd_from_var(var.numel(), eps, ptr(var), ptr(rstd), stream()), "rstd_from_var")
    return rstd


def bn_apply(x, mean, rstd, gamma, beta, act=None, p=0.0, seed=(None, 0), want_pre=False, vlens=None, Tn=0):
    C = x.shape[-1]
    rows = x.numel() // C
    y = torch.empty_like(x)
    pre = torch.empty_like(x) if want_pre else None
    _lib.check(_lib.lib().s2svc_bn_apply(dt(x), rows, C, ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ACT[act], p,
                                         seed[0], seed[1], ptr(y), ptr(pre), Tn, ptr(vlens), stream()), "bn_apply")
    return y, pre


def bn_vec_ok(x):
    """bf16 channel-last rows with C % 8 == 0: the 16-byte BatchNorm kernels of csrc/convmod.hip apply (S2SVC_NO_BN_VEC=1: A/B switch)."""
    return (x.dtype == torch.bfloat16 and x.shape[-1] % 8 == 0 and x.data_ptr() % 16 == 0 and x.numel() < 2 ** 31
            and os.environ.get("S2SVC_NO_BN_VEC", "0") != "1")


def bn_stats_vec(x, rows, C, eps, momentum, run_mean=None, run_var=None, num_batches=None, vlens=None, Tn=0):
    mean = torch.empty(C, dtype=torch.float32, device=x.device)
    rstd = torch.empty(C, dtype=torch.float32, device=x.device)
    ws = torch.empty(((rows + 63) // 64) * 2 * C, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().s2svc_bn_stats_vec(rows, C, ptr(x), eps, momentum, ptr(mean), ptr(rstd), ptr(run_mean), ptr(run_var),
                                             ptr(num_batches), ptr(ws), Tn, ptr(vlens), stream()), "bn_stats_vec")
    return mean, rstd


def bn_act_apply_vec(x, mean, rstd, gamma, beta, act=None, p=0.0, seed=(None, 0), want_pre=False, vlens=None, Tn=0):
    C = x.shape[-1]
    y = torch.empty_like(x)
    pre = torch.empty_like(x) if want_pre else None
    _lib.check(_lib.lib().s2svc_bn_act_apply_vec(x.numel() // C, C, ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ACT[act], p,
                                                 seed[0], seed[1], ptr(y), ptr(pre), Tn, ptr(vlens), stream()), "bn_act_apply_vec")
    return y, pre


def bn_act_bwd_vec(dz, saved, x, mean, rstd, gamma, act=None, p=0.0, seed=(None, 0), dgamma_acc=None, dbeta_acc=None, vlens=None, Tn=0):
    """-> dx, sdy (= d beta), sdyx (= d gamma); dgamma_acc / dbeta_acc: fp32 (C) gradient slots the sums are ADDED to."""
    C = x.shape[-1]
    rows = x.numel() // C
    dx = torch.empty_like(x)
    sdy = torch.empty(C, dtype=torch.float32, device=x.device)
    sdyx = torch.empty(C, dtype=torch.float32, device=x.device)
    ws = torch.empty(((rows + 63) // 64) * 2 * C, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().s2svc_bn_act_bwd_vec(rows, C, ptr(dz), ptr(saved), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ACT[act], p,
                                               seed[0], seed[1], ptr(dx), ptr(sdy), ptr(sdyx), ptr(dgamma_acc), ptr(dbeta_acc),
                                               ptr(ws), Tn, ptr(vlens), stream()), "bn_act_bwd_vec")
    return dx, sdy, sdyx


def bn_bwd(dy, x, mean, rstd, gamma, sum_dy, sum_dy_xhat, use_batch_stats=True, vlens=None, Tn=0):
    C = x.shape[-1]
    rows = x.numel() // C
    dx = torch.empty_like(x)
    _lib.check(_lib.lib().s2svc_bn_bwd(dt(x), rows, C, ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(sum_dy),
                                       ptr(sum_dy_xhat), 1 if use_batch_stats else 0, ptr(dx), Tn, ptr(vlens), stream()), "bn_bwd")
    return dx


# ----------------------------------------------------------------------------------------------
# attention probabilities
# ----------------------------------------------------------------------------------------------
def attn_softmax_fwd(scores, out_dtype, scale, klen=None, causal=False, bd=None, rel_mode=0, p=0.0, seed=(None, 0), T2=None,
                     Lp=None):
    """scores: fp32 (B, H, T1, ld) with ld >= T2 (rows padded to a vector multiple); outputs share that layout.
    bd: fp32 (B, H, T1, ldb) relative-position term whose rows hold Lp <= ldb values."""
    B, H, T1, ld = scores.shape
    T2 = ld if T2 is None else T2
    attn = torch.empty(scores.shape, dtype=out_dtype, device=scores.device)
    pdrop = torch.empty_like(attn) if p > 0.0 else None
    ldb = bd.shape[-1] if bd is not None else 0
    Lp = ldb if Lp is None else Lp
    _lib.check(_lib.lib().s2svc_attn_softmax_fwd(_DT[out_dtype], B, H, T1, T2, ld, ptr(scores), ptr(bd), Lp, ldb, rel_mode, scale,
                                                 ptr(klen), 1 if causal else 0, p, seed[0], seed[1], ptr(attn), ptr(pdrop),
                                                 stream()), "attn_softmax_fwd")
    return attn, pdrop


def attn_softmax_bwd(attn, dp, scale, p=0.0, seed=(None, 0), Lp=0, rel_mode=0, dattn=None, T2=None, ldb=None):
    """Returns (dscores, dbd); dbd is (B, H, T1, ldb) with columns >= Lp zero (ldb defaults to Lp)."""
    B, H, T1, ld = attn.shape
    T2 = ld if T2 is None else T2
    ldb = Lp if ldb is None else ldb
    dscores = torch.empty_like(attn)
    dbd = torch.empty((B, H, T1, ldb), dtype=attn.dtype, device=attn.device) if Lp else None
    _lib.check(_lib.lib().s2svc_attn_softmax_bwd(dt(attn), B, H, T1, T2, ld, ptr(attn), ptr(dp), ptr(dattn), scale, p, seed[0],
                                                 seed[1], ptr(dscores), ptr(dbd), Lp, ldb, rel_mode, stream()), "attn_softmax_bwd")
    return dscores, dbd


# ----------------------------------------------------------------------------------------------
# elementwise
# ----------------------------------------------------------------------------------------------
def act_dropout_fwd(x, act=None, p=0.0, seed=(None, 0)):
    y = torch.empty_like(x)
    _lib.check(_lib.lib().s2svc_act_dropout_fwd(dt(x), x.numel(), ptr(x), ACT[act], p, seed[0], seed[1], ptr(y), stream()),
               "act_dropout_fwd")
    return y


def act_dropout_bwd(dz, saved, act=None, p=0.0, seed=(None, 0)):
    dx = torch.empty_like(dz)
    _lib.check(_lib.lib().s2svc_act_dropout_bwd(dt(dz), dz.numel(), ptr(dz), ptr(saved), ACT[act], p, seed[0], seed[1],
                                                ptr(dx), stream()), "act_dropout_bwd")
    return dx


def posenc_fwd(x, xscale, alpha, pe, p=0.0, seed=(None, 0)):
    B, T, D = x.shape
    y = torch.empty_like(x)
    _lib.check(_lib.lib().s2svc_posenc_fwd(dt(x), B, T, D, ptr(x), xscale, ptr(alpha), ptr(pe), p, seed[0], seed[1], ptr(y),
                                           stream()), "posenc_fwd")
    return y


def posenc_bwd(dy, xscale, pe, p=0.0, seed=(None, 0), want_dalpha=False):
    B, T, D = dy.shape
    dx = torch.empty_like(dy)
    dalpha = torch.empty((), dtype=torch.float32, device=dy.device) if want_dalpha else None
    part = torch.empty(2048, dtype=torch.float32, device=dy.device) if want_dalpha else None
    _lib.check(_lib.lib().s2svc_posenc_bwd(dt(dy), B, T, D, ptr(dy), xscale, ptr(pe), p, seed[0], seed[1], ptr(dx),
                                           ptr(dalpha), ptr(part), stream()), "posenc_bwd")
    return dx, dalpha


def axpby(a, x, b=0.0, y=None, out=None):
    out = torch.empty_like(x) if out is None else out
    _lib.check(_lib.lib().s2svc_axpby(dt(x), x.numel(), a, ptr(x), b, ptr(y), ptr(out), stream()), "axpby")
    return out


def add_n(xs, out=None):
    """Sum of 2..4 tensors of one shape / dtype in one launch (fp32 arithmetic, fixed order)."""
    x0 = xs[0]
    out = torch.empty_like(x0) if out is None else out
    p = [ptr(t) for t in xs] + [None] * (4 - len(xs))
    _lib.check(_lib.lib().s2svc_add_n(dt(x0), x0.numel(), len(xs), p[0], p[1], p[2], p[3], ptr(out), stream()), "add_n")
    return out


def add_head_bias(q, u, v):
    D = q.shape[-1]
    qu, qv = torch.empty_like(q), torch.empty_like(q)
    _lib.check(_lib.lib().s2svc_add_head_bias(dt(q), q.numel() // D, D, ptr(q), ptr(u), ptr(v), ptr(qu), ptr(qv), stream()),
               "add_head_bias")
    return qu, qv


def add_head_bias_view(q, u, v):
    """q: a (B, T, D) column block of a packed projection (unit column stride, row stride q.stride(-2)) -> dense qu, qv."""
    D = q.shape[-1]
    assert q.stride(-1) == 1 and q.dim() == 3 and q.stride(0) == q.shape[1] * q.stride(1)
    qu = torch.empty(q.shape, dtype=q.dtype, device=q.device)
    qv = torch.empty(q.shape, dtype=q.dtype, device=q.device)
    _lib.check(_lib.lib().s2svc_add_head_bias_ld(dt(q), q.numel() // D, D, ptr(q), q.stride(1), ptr(u), ptr(v), ptr(qu), ptr(qv),
                                                 stream()), "add_head_bias_ld")
    return qu, qv


def add_rows(a, b, out):
    """out = a + b for (B, T, D) tensors that may be column blocks of packed tensors (unit column stride)."""
    D = a.shape[-1]
    for t in (a, b, out):
        assert t.stride(-1) == 1 and t.stride(0) == t.shape[1] * t.stride(1)
    _lib.check(_lib.lib().s2svc_add_rows(dt(a), a.numel() // D, D, ptr(a), a.stride(1), ptr(b), b.stride(1), ptr(out), out.stride(1),
                                         stream()), "add_rows")
    return out


def glu_fwd(x):
    C = x.shape[-1] // 2
    y = torch.empty(x.shape[:-1] + (C,), dtype=x.dtype, device=x.device)
    _lib.check(_lib.lib().s2svc_glu_fwd(dt(x), y.numel() // C, C, ptr(x), ptr(y), stream()), "glu_fwd")
    return y


def glu_bwd(x, dy):
    C = x.shape[-1] // 2
    dx = torch.empty_like(x)
    _lib.check(_lib.lib().s2svc_glu_bwd(dt(x), dy.numel() // C, C, ptr(x), ptr(dy), ptr(dx), stream()), "glu_bwd")
    return dx


def cast(x, dtype, out=None):
    if x.dtype == dtype and out is None:
        return x
    y = torch.empty(x.shape, dtype=dtype, device=x.device) if out is None else out
    if out is not None and (out.dtype != dtype or out.numel() != x.numel() or not out.is_contiguous() or not x.is_contiguous()):
        raise ValueError("cast(out=): contiguous tensors of the target dtype and the same size are required")
    _lib.check(_lib.lib().s2svc_cast(dt(x), _DT[dtype], x.numel(), ptr(x), ptr(y), stream()), "cast")
    return y


def embedding_fwd(idx, w, out_dtype):
    """idx int64 (...,), w fp32 (V, D) -> (..., D)."""
    V, D = w.shape
    idx = idx.contiguous()
    y = torch.empty((*idx.shape, D), dtype=out_dtype, device=w.device)
    _lib.check(_lib.lib().s2svc_embedding_fwd(_DT[out_dtype], idx.numel(), D, V, ptr(idx), ptr(w), ptr(y), stream()), "embedding_fwd")
    return y


def embedding_bwd(idx, dy, V, padding_idx, out=None, accumulate=False):
    D = dy.shape[-1]
    dw = torch.empty((V, D), dtype=torch.float32, device=dy.device) if out is None else out
    _lib.check(_lib.lib().s2svc_embedding_bwd(dt(dy), idx.numel(), D, V, ptr(idx), ptr(dy), -1 if padding_idx is None else padding_idx,
                                              ptr(dw), 1 if accumulate else 0, stream()), "embedding_bwd")
    return dw


class PermRegistry(list):
    """The derived weight copies one optimiser (optim.FlatAdam) keeps fresh -- the permuted convolution weights registered
    here by gather3_cached, and the transposed bf16 shadow -- plus the state of the step PROLOGUE: after an optimiser step
    the copies are `due`; FlatAdam.begin_step() launches the refresh (and the zero-fill of the flat gradient buffer) on a
    prologue stream beside the forward pass, and the consumers wait for its events (`sync`, `join`).  Nothing that reads a
    copy can see a stale one: a consumer that arrives while the refresh is still due runs it on its own stream first.
    `covered` = how many of the permuted copies the CAPTURED refresh launch updates on replay (None: it runs from Python
    and updates all of them)."""
    covered = None
    due = False          # an optimiser step ran and the refresh has not been launched
    refresh = None       # callable set by the optimiser: launch the refresh now, on the current stream
    ev_perm = None       # recorded on the prologue stream behind the permuted copies / behind everything
    ev_all = None
    waited = ()

    def sync(self, what="all"):
        """Called by every consumer of a derived copy (what = "perm": the permuted convolution weights only)."""
        if self.due and self.refresh is not None:
            # no begin_step() since the optimiser step (evaluation / inference after training): refresh here.  Other streams
            # may read the copies next without passing an event, so the host waits -- a rare path; inside a capture the
            # refresh becomes part of the graph on the capturing stream, where its consumers are
            self.refresh()
            if torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
                torch.cuda.current_stream().synchronize()
        ev = self.ev_perm if what == "perm" else self.ev_all
        if ev is not None:
            cur = torch.cuda.current_stream()
            key = (cur.cuda_stream, what)
            if key not in self.waited and (cur.cuda_stream, "all") not in self.waited:
                cur.wait_event(ev)
                self.waited.add(key)

    def join(self):
        """The current stream waits for the whole prologue; later consumers (streams forked from this one) need not."""
        if self.ev_all is not None:
            torch.cuda.current_stream().wait_event(self.ev_all)
        self.ev_perm = self.ev_all = None
        self.waited = set()


def gather3_cached(weight, n, strides, off, out_dtype):
    """gather3 of a PARAMETER.  When optim.FlatAdam manages it, the permuted copy lives in a persistent buffer that the
    optimiser refreshes after every update -- ONE grouped launch for all such copies of the model (gather3_refresh) instead of
    a launch per convolution and pass on the step's dependency chain; otherwise a plain gather3 of the current values."""
    reg = getattr(weight, "_s2s_perm_registry", None)
    if reg is None:
        return gather3(weight.detach(), n, strides, off, out_dtype)
    reg.sync("perm")
    key = (tuple(n), tuple(strides), int(off), out_dtype)
    ent = weight._s2s_perms.get(key)
    if ent is None:
        buf = gather3(weight.detach(), n, strides, off, out_dtype)
        weight._s2s_perms[key] = [buf, weight._version, len(reg)]
        reg.append((weight, key, buf))
        return buf
    cov = getattr(reg, "covered", None)
    # changed in place by torch (load_state_dict, copy_) since the copy was taken, or registered after the optimiser step
    # was captured (replays update the weights through raw pointers and refresh only the copies they knew): gather again
    if ent[1] != weight._version or (cov is not None and ent[2] >= cov):
        _lib.check(_lib.lib().s2svc_gather3(dt(weight), _DT[out_dtype], n[0], n[1], n[2], strides[0], strides[1], strides[2], off,
                                            ptr(weight), ptr(ent[0]), stream()), "gather3")
        ent[1] = weight._version
    return ent[0]


def gather3_refresh(registry):
    """Recompute every registered permuted copy from the current fp32 weights (one launch per 24 of them)."""
    if not registry:
        return
    jobs = (_lib.Gather3Job * len(registry))()
    for jb, (w, (n, st, off, odt), buf) in zip(jobs, registry):
        jb.in_, jb.out = w.data_ptr(), buf.data_ptr()
        jb.s0, jb.s1, jb.s2, jb.off = st[0], st[1], st[2], off
        jb.n0, jb.n1, jb.n2, jb.out_dtype = n[0], n[1], n[2], _DT[odt]
    _lib.check(_lib.lib().s2svc_gather3_grouped(ctypes.addressof(jobs), len(registry), stream()), "gather3_grouped")


def gather3(x, n, strides, off, out_dtype):
    """out[i0,i1,i2] = x.flat[off + i0*s0 + i1*s1 + i2*s2] (contiguous result of shape n)."""
    y = torch.empty(n, dtype=out_dtype, device=x.device)
    _lib.check(_lib.lib().s2svc_gather3(dt(x), _DT[out_dtype], n[0], n[1], n[2], strides[0], strides[1], strides[2], off,
                                        ptr(x), ptr(y), stream()), "gather3")
    return y


def permute_inner(src, n, A, Bn, out=None, accumulate=False):
    """out[o][b][a] (+)= src[o][a][b] (fp32, o < n): a convolution weight gradient from its GEMM layout (C_out, taps, C_in) into the
    parameter's (C_out, C_in, taps) -- straight into the flat-gradient slot when `out` is one (accumulate=True)."""
    _need_cuda(src)
    if src.dtype != torch.float32 or not src.is_contiguous():
        raise TypeError("permute_inner: contiguous fp32 source")
    if src.numel() != n * A * Bn:
        raise ValueError(f"permute_inner: source has {src.numel()} elements, n * A * Bn = {n * A * Bn}")
    if out is None:
        out, accumulate = torch.empty(n * A * Bn, dtype=torch.float32, device=src.device), False
    elif out.dtype != torch.float32 or not out.is_contiguous() or out.numel() != n * A * Bn or out.device != src.device:
        raise TypeError(f"permute_inner: `out` must be a contiguous fp32 tensor of {n * A * Bn} elements on {src.device} "
                        f"(got {out.dtype}, contiguous={out.is_contiguous()}, {out.numel()} elements)")
    _lib.check(_lib.lib().s2svc_permute_inner(n, A, Bn, ptr(src), ptr(out), 1 if accumulate else 0, stream()), "permute_inner")
    return out


def permute_inner_ok(A, Bn):
    return A * (Bn + 1) * 4 <= 64 * 1024


# ----------------------------------------------------------------------------------------------
# monotonic alignment search
# ----------------------------------------------------------------------------------------------
def mas(log_p_attn, text_lens_i32, feat_lens_i32):
    B, Tf, Tx = log_p_attn.shape
    dev = log_p_attn.device
    path = torch.empty((B, Tf), dtype=torch.int32, device=dev)
    ds = torch.empty((B, Tx), dtype=torch.float32, device=dev)
    binmean = torch.empty((B,), dtype=torch.float32, device=dev)
    nbytes = _lib.lib().s2svc_mas_ws_bytes(B, Tf, Tx)
    ws = torch.empty(nbytes // 8 + 1, dtype=torch.int64, device=dev)
    _lib.check(_lib.lib().s2svc_mas(B, Tf, Tx, ptr(log_p_attn), ptr(text_lens_i32), ptr(feat_lens_i32), ptr(path), ptr(ds),
                                    ptr(binmean), ptr(ws), stream()), "mas")
    return ds, path, binmean


def mas_binloss_bwd(path, feat_lens_i32, gout, dlogp):
    B, Tf = path.shape
    Tx = dlogp.shape[-1]
    _lib.check(_lib.lib().s2svc_mas_binloss_bwd(B, Tf, Tx, ptr(path), ptr(feat_lens_i32), ptr(gout), ptr(dlogp), stream()),
               "mas_binloss_bwd")
    return dlogp


# ----------------------------------------------------------------------------------------------
# losses
# ----------------------------------------------------------------------------------------------
def seq_loss_fwd(after, before, logits, ys, labels, olens_i32, pos_weight):
    B, Tm, D = before.shape
    dev = before.device
    partial = torch.empty(3 * 1024, dtype=torch.float32, device=dev)
    out = torch.empty(3, dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().s2svc_seq_loss_fwd(dt(before), B, Tm, D, ptr(after), ptr(before), ptr(logits), ptr(ys), ptr(labels),
                                             ptr(olens_i32), pos_weight, ptr(partial), ptr(out), stream()), "seq_loss_fwd")
    return out


def seq_loss_bwd(after, before, logits, ys, labels, olens_i32, pos_weight, stats, g_l1, g_bce):
    B, Tm, D = before.shape
    d_after = torch.empty_like(after) if after is not None else None
    d_before = torch.empty_like(before)
    d_logits = torch.empty_like(logits) if logits is not None else None
    _lib.check(_lib.lib().s2svc_seq_loss_bwd(dt(before), B, Tm, D, ptr(after), ptr(before), ptr(logits), ptr(ys), ptr(labels),
                                             ptr(olens_i32), pos_weight, ptr(stats), ptr(g_l1), ptr(g_bce), ptr(d_after),
                                             ptr(d_before), ptr(d_logits), stream()), "seq_loss_bwd")
    return d_after, d_before, d_logits


def guided_attn_loss_fwd(att, ilens_i32, olens_i32, sigma, alpha):
    B, H, To, Ti = att.shape
    partial = torch.empty(1024, dtype=torch.float32, device=att.device)
    out = torch.empty(2, dtype=torch.float32, device=att.device)
    _lib.check(_lib.lib().s2svc_guided_attn_loss_fwd(dt(att), B, H, To, Ti, ptr(att), ptr(ilens_i32), ptr(olens_i32), sigma, alpha,
                                                     ptr(partial), ptr(out), stream()), "guided_attn_loss_fwd")
    return out


def guided_attn_loss_bwd(shape, dtype, device, ilens_i32, olens_i32, sigma, alpha, stats, gout):
    B, H, To, Ti = shape
    datt = torch.empty(shape, dtype=dtype, device=device)
    _lib.check(_lib.lib().s2svc_guided_attn_loss_bwd(_DT[dtype], B, H, To, Ti, ptr(ilens_i32), ptr(olens_i32), sigma, alpha,
                                                     ptr(stats), ptr(gout), ptr(datt), stream()), "guided_attn_loss_bwd")
    return datt


# ----------------------------------------------------------------------------------------------
# optimiser
# ----------------------------------------------------------------------------------------------
def adam_step(params, grads, exp_avg, exp_avg_sq, shadow, state, partial, lr, betas=(0.9, 0.999), eps=1e-8, max_norm=1.0,
              warmup_steps=4000.0):
    _lib.check(_lib.lib().s2svc_adam_step(params.numel(), ptr(params), ptr(grads), ptr(exp_avg), ptr(exp_avg_sq), ptr(shadow),
                                          betas[0], betas[1], eps, max_norm, lr, warmup_steps, ptr(partial), ptr(state),
                                          stream()), "adam_step")


def transpose_tiles(tiles, src, dst):
    """tiles: int64 device tensor (ntiles, 4) built by optim.FlatAdam; src / dst: flat bf16 buffers."""
    _lib.check(_lib.lib().s2svc_transpose_tiles(tiles.shape[0], ptr(tiles), ptr(src), ptr(dst), stream()), "transpose_tiles")


# ----------------------------------------------------------------------------------------------
# convolution helpers
# ----------------------------------------------------------------------------------------------
def tconv2d_weights(weight):
    """fp32 (O, C, 3, 3) -> bf16 class matrices of the stride-2 transposed convolution, list of 4 views (C, ntaps*O)
    in class order (pt, pf) = (0,0), (0,1), (1,0), (1,1)  (s2svc_tconv2d_weights)."""
    O, C = weight.shape[0], weight.shape[1]
    out = torch.empty(9 * C * O, dtype=torch.bfloat16, device=weight.device)
    _lib.check(_lib.lib().s2svc_tconv2d_weights(O, C, ptr(weight), ptr(out), stream()), "tconv2d_weights")
    views, off = [], 0
    for ntaps in (4, 2, 2, 1):
        views.append(out[off:off + C * ntaps * O].view(C, ntaps * O))
        off += C * ntaps * O
    return views


def col2im_s2(dcols, B, T1, F1, C, T2, F2):
    dx = torch.empty((B, T1, F1, C), dtype=dcols.dtype, device=dcols.device)
    _lib.check(_lib.lib().s2svc_col2im_s2(dt(dcols), B, T1, F1, C, T2, F2, ptr(dcols), ptr(dx), stream()), "col2im_s2")
    return dx


def interp_nearest(x, Tout, ext_in=None, ext_out=None):
    """ext_in / ext_out (int32 device tensors, element 0 is read): the cropped lengths behind padded Tin / Tout (include/s2svc_hip.h)."""
    B, Tin, C = x.shape
    y = torch.empty((B, Tout, C), dtype=x.dtype, device=x.device)
    _lib.check(_lib.lib().s2svc_interp_nearest(dt(x), B, Tin, Tout, C, ptr(x), ptr(y), ptr(ext_in), ptr(ext_out), stream()), "interp_nearest")
    return y


def interp_nearest_bwd(dy, Tin, ext_in=None, ext_out=None):
    B, Tout, C = dy.shape
    dx = torch.empty((B, Tin, C), dtype=dy.dtype, device=dy.device)
    _lib.check(_lib.lib().s2svc_interp_nearest_bwd(dt(dy), B, Tin, Tout, C, ptr(dy), ptr(dx), ptr(ext_in), ptr(ext_out), stream()),
               "interp_nearest_bwd")
    return dx


def conv_in1_fwd(x, w, bias):
    """x (B,T,F) compute dtype; w fp32 (O,1,3,3); -> relu(conv3x3 s2) as NHWC (B,T1,F1,O)."""
    B, T, Fd = x.shape
    O = w.shape[0]
    y = torch.empty((B, (T - 3) // 2 + 1, (Fd - 3) // 2 + 1, O), dtype=x.dtype, device=x.device)
    _lib.check(_lib.lib().s2svc_conv_in1_fwd(dt(x), B, T, Fd, O, ptr(x), ptr(w), ptr(bias), ptr(y), stream()), "conv_in1_fwd")
    return y


_CI_CHUNKS = 512


def conv_in1_wgrad(x, dy, dw, db, accumulate, y=None):
    B, T, Fd = x.shape
    O = dy.shape[-1]
    chunks = _CI_CHUNKS
    partial = torch.empty(chunks * O * 10, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().s2svc_conv_in1_wgrad(dt(x), B, T, Fd, O, ptr(x), ptr(dy), ptr(y), ptr(dw), ptr(db), 1 if accumulate else 0,
                                               ptr(partial), chunks, stream()), "conv_in1_wgrad")
