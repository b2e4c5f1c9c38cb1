"""Flat-buffer Adam with gradient clipping and WarmupLR, one fused HIP launch sequence per step.

Replaces the reference's `clip_grad_norm_` -> `torch.optim.Adam.step()` -> `WarmupLR.step()` sequence
(trainers/ar_vc.py:99-107, trainers/aas_vc.py:151-158, schedulers/warmup_lr.py:54-61).

All trainable parameters of the model are re-pointed into ONE contiguous fp32 buffer (`flat_p`); their
gradients live in a parallel buffer (`flat_g`, exposed as `p.grad` views and as `p._s2s_grad` so the
wgrad kernels accumulate straight into it); Adam moments and the bf16 shadow used by bf16 GEMMs are
flat as well, and the matrix-shaped weights additionally keep a TRANSPOSED bf16 copy (refreshed by one batched tile-
transpose launch per step) so that data-gradient GEMMs read K-contiguous operands.  The step counter, learning rate, gradient norm and clip coefficient live in a 4-float
device tensor, so the optimiser step is hipGraph-capturable and needs no host synchronisation.  The
flat gradient buffer is also what data-parallel training all-reduces (distributed.py): a handful of
large RCCL collectives instead of one per tensor.
"""
import os

import torch

from .ops import kernels as K


class FlatAdam:
    def __init__(self, model, lr=8e-5, betas=(0.9, 0.999), eps=1e-8, grad_norm=1.0, warmup_steps=4000, bf16_shadow=False,
                 align=64, fuse_qkv=True, transposed_shadow=True):
        # torch.optim.Adam(model.parameters()) -- what the reference builds (bin/vc_train.py:405-416) -- numbers its state by
        # position in model.parameters(), frozen parameters included; kept for checkpoint interchange (state_dict below)
        self.all_params = list(model.parameters())
        self.params = [p for p in self.all_params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        self._model = model
        # lay the Q/K/V weights (and biases) of every attention module out back to back, so that
        # [Wq;Wk;Wv] is ONE (3D, D) matrix in the flat buffer (fused projection GEMMs, see modules.py)
        stacks = self._source_attention_stacks(model) if fuse_qkv else []
        in_stack = {id(m) for st in stacks for m in st["mods"]}
        groups = self._attention_groups(model, skip=in_stack) if fuse_qkv else []
        grouped = {id(p) for g in groups for p in g["w"] + g["b"]} | {id(p) for st in stacks for p in st["w"] + st["b"]}
        first = {id(g["w"][0]): g for g in groups}
        first.update({id(st["w"][0]): st for st in stacks})
        ordered, tight = [], set()     # `tight` members start right where the previous one ends (no alignment gap)
        for p in self.params:
            if id(p) in first:
                g = first[id(p)]
                ordered += g["w"] + g["b"]
                if all(q.numel() % 8 == 0 for q in g["w"] + g["b"]):
                    tight.update(id(q) for q in g["w"][1:] + g["b"][1:])
            elif id(p) not in grouped:
                ordered.append(p)
        self.params = ordered
        dev = self.params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatAdam needs the model on the GPU (there is no CPU path)")
        self.lr, self.betas, self.eps = float(lr), betas, float(eps)
        self.grad_norm, self.warmup_steps = float(grad_norm), float(warmup_steps or 0)
        offs, n = [], 0
        for p in self.params:
            if id(p) not in tight:
                n = (n + align - 1) // align * align
            offs.append(n)
            n += p.numel()
        n = (n + align - 1) // align * align
        self.offsets, self.numel = offs, n
        self.flat_p = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.shadow = torch.zeros(n, dtype=torch.bfloat16, device=dev) if bf16_shadow else None
        self.state = torch.zeros(4, dtype=torch.float32, device=dev)  # step, lr, grad_norm, clip_coef
        self.partial = torch.empty(1024, dtype=torch.float64, device=dev)
        self._perm_jobs = K.PermRegistry()   # (parameter, permutation, persistent buffer): filled by ops.kernels.gather3_cached
        self._perm_jobs.refresh = self._refresh_transposed
        self._zero_due = False
        self._began = False           # begin_step() was called since the last step()
        for p, o in zip(self.params, offs):
            k = p.numel()
            self.flat_p[o:o + k].copy_(p.data.reshape(-1))
            p.data = self.flat_p[o:o + k].view(p.shape)
            p._s2s_grad = self.flat_g[o:o + k].view(p.shape)
            p.grad = p._s2s_grad
            p._s2s_perm_registry, p._s2s_perms = self._perm_jobs, {}
            if self.shadow is not None:
                p._s2s_bf16 = self.shadow[o:o + k].view(p.shape)
        off_of = {id(p): o for p, o in zip(self.params, offs)}
        # transposed bf16 shadow of the matrix-shaped weights (Linear, 1x1 Conv1d): dX = dY.W then reads a K-contiguous
        # operand and runs on the all-DMA GEMM kernel.  Matrices keep their offset; fused QKV / KV views get extra room.
        self.shadow_t, self._t_descs, self._t_extra = None, [], n
        if self.shadow is not None and transposed_shadow:
            for p, o in zip(self.params, offs):
                if p.dim() == 2 or (p.dim() == 3 and p.shape[-1] == 1):
                    self._t_descs.append((o, o, p.shape[0], p.shape[1], p))
        for g in groups:
            ws, bs = g["w"], g["b"]
            D = ws[0].shape[0]
            ow, ob = off_of[id(ws[0])], off_of[id(bs[0])]
            contiguous = all(off_of[id(ws[i])] == ow + i * D * D for i in range(3)) and \
                all(off_of[id(bs[i])] == ob + i * D for i in range(3))
            if contiguous:
                g["module"]._fused = {
                    "w_qkv": self._view(ow, (3 * D, D)), "b_qkv": self._view(ob, (3 * D,)),
                    "w_q": self._view(ow, (D, D)), "b_q": self._view(ob, (D,)),
                    "w_kv": self._view(ow + D * D, (2 * D, D)), "b_kv": self._view(ob + D, (2 * D,))}
                if self.shadow is not None and transposed_shadow:
                    f = g["module"]._fused
                    for key, off, rows in (("w_qkv", ow, 3 * D), ("w_kv", ow + D * D, 2 * D)):
                        self._t_descs.append((off, self._t_extra, rows, D, f[key]))
                        self._t_extra += (rows * D + align - 1) // align * align
        # decoder stacks: [Wk_1; Wv_1; ...; Wk_L; Wv_L] of the source-attention blocks is ONE (L*2D, D) matrix -- the memory is
        # projected for all layers by a single GEMM (modules.Decoder.forward); each block keeps its own (2D, D) view
        for st in stacks:
            ws, bs, mods = st["w"], st["b"], st["mods"]
            D = ws[0].shape[0]
            ow, ob = off_of[id(ws[0])], off_of[id(bs[0])]
            contiguous = all(off_of[id(w)] == ow + i * D * D for i, w in enumerate(ws)) and \
                all(off_of[id(b)] == ob + i * D for i, b in enumerate(bs))
            if not contiguous:
                continue
            for li, m in enumerate(mods):
                q_w, q_b = m.linear_q.weight, m.linear_q.bias
                m._fused = {"w_q": q_w, "b_q": q_b,
                            "w_kv": self._view(ow + li * 2 * D * D, (2 * D, D)), "b_kv": self._view(ob + li * 2 * D, (2 * D,))}
            allv = {"w": self._view(ow, (len(ws) * D, D)), "b": self._view(ob, (len(bs) * D,))}
            st["decoder"]._src_kv_all = allv
            if self.shadow is not None and transposed_shadow:
                self._t_descs.append((ow, self._t_extra, len(ws) * D, D, allv["w"]))
                self._t_extra += (len(ws) * D * D + align - 1) // align * align
        if self._t_descs:
            self.shadow_t = torch.zeros(self._t_extra, dtype=torch.bfloat16, device=dev)
            tiles = []
            for so, do, rows, cols, t in self._t_descs:
                t._s2s_bf16_t = self.shadow_t[do:do + rows * cols].view(cols, rows)
                nt = ((rows + 63) // 64) * ((cols + 63) // 64)
                tiles += [(so, do, (rows << 32) | cols, i) for i in range(nt)]
            self.t_tiles = torch.tensor(tiles, dtype=torch.int64, device=dev)
            for g in groups:                                                   # w_q view: the transposed copy of linear_q.weight
                f = getattr(g["module"], "_fused", None)
                if f is not None:
                    f["w_q"]._s2s_bf16_t = g["w"][0]._s2s_bf16_t
        if self.shadow is not None:
            self.refresh_shadow()

    @staticmethod
    def _source_attention_stacks(model):
        """Per modules.Decoder: the K / V weights and biases of the source-attention blocks of all its layers, in layer order
        (k_1, v_1, k_2, v_2, ...)."""
        from .modules import Decoder, MultiHeadedAttention
        stacks = []
        for dec in model.modules():
            if not isinstance(dec, Decoder):
                continue
            mods = [layer.src_attn for layer in dec.decoders]
            if not mods or not all(type(m) is MultiHeadedAttention for m in mods):
                continue
            ws = [w for m in mods for w in (m.linear_k.weight, m.linear_v.weight)]
            bs = [b for m in mods for b in (m.linear_k.bias, m.linear_v.bias)]
            qs = [p for m in mods for p in (m.linear_q.weight, m.linear_q.bias)]
            if all(p is not None and p.requires_grad for p in ws + bs + qs) and len({w.shape for w in ws}) == 1:
                stacks.append({"decoder": dec, "mods": mods, "w": ws, "b": bs})
        return stacks

    @staticmethod
    def _attention_groups(model, skip=()):
        from .modules import MultiHeadedAttention
        groups = []
        for m in model.modules():
            if id(m) in skip:
                continue
            if isinstance(m, MultiHeadedAttention):
                ws = [m.linear_q.weight, m.linear_k.weight, m.linear_v.weight]
                bs = [m.linear_q.bias, m.linear_k.bias, m.linear_v.bias]
                if all(p is not None and p.requires_grad for p in ws + bs):
                    groups.append({"module": m, "w": ws, "b": bs})
        return groups

    def _view(self, off, shape):
        """A trainable-looking view of the flat buffers (fp32 master, flat-gradient slot, bf16 shadow)."""
        n = 1
        for d in shape:
            n *= d
        t = self.flat_p[off:off + n].view(shape)
        t.requires_grad_(True)
        t._s2s_grad = self.flat_g[off:off + n].view(shape)
        t._s2s_perm_registry = self._perm_jobs
        if self.shadow is not None:
            t._s2s_bf16 = self.shadow[off:off + n].view(shape)
        return t

    def _touch(self):
        """The parameters changed through raw pointers (no tensor version bump): advance the model's weight generation so
        that cached derived state (decode sessions with packed / bf16 weight copies, decode.py) is rebuilt."""
        self._model.__dict__["_s2s_weight_gen"] = self._model.__dict__.get("_s2s_weight_gen", 0) + 1

    def refresh_shadow(self):
        """bf16 copy of the fp32 master weights (after loading a checkpoint / at start)."""
        self._touch()
        if self.shadow is not None:
            self.shadow.copy_(K.cast(self.flat_p, torch.bfloat16))
        self._refresh_transposed()

    def _refresh_transposed(self):
        """Derived copies of the weights: the transposed bf16 shadow and the permuted convolution weights that the forward /
        backward passes registered (ops.kernels.gather3_cached) -- one launch each.  The permuted copies first: the forward pass
        needs them (PermRegistry.ev_perm), the transposed shadow only feeds data-gradient GEMMs."""
        reg = self._perm_jobs
        reg.due = False
        K.gather3_refresh(reg)
        if self.flat_p.is_cuda and torch.cuda.is_current_stream_capturing():
            # a captured refresh updates exactly the copies registered NOW on every replay; copies that register later (first
            # evaluation in another dtype, a convolution first used later) are not in it
            reg.covered = len(reg) if reg.covered is None else min(reg.covered, len(reg))
        if self.shadow_t is not None:
            K.transpose_tiles(self.t_tiles, self.shadow, self.shadow_t)

    # -- the step prologue -------------------------------------------------------------------------------------------------
    # After an optimiser step three memory-bound passes stand between it and the next backward pass: the permuted convolution
    # weights (needed by the forward pass), the transposed bf16 shadow (data-gradient GEMMs) and the zero-fill of the flat
    # gradient buffer (weight-gradient kernels accumulate): 136 + 144 + 80 us of a 13 ms AAS-VC step, 32 + 27 + 18 us of a
    # 4.0 ms VTN step.  They run IN LINE: the refresh at the end of step(), the zero-fill in begin_step() / zero_grad().
    # (Rounds 3-5 carried an opt-in that ran them on a prologue stream beside the forward pass: 13.0 vs 12.74 ms per AAS-VC step,
    # VTN equal; as capped grids of 16 ... 2048 workgroups 11.52-11.59 vs 11.40 ms -- what runs beside the forward chain is not
    # free even when it is HBM-bound.  Removed in round 6; profiles/AB_LOG.md.)

    def begin_step(self, zero=True):
        """First call of a training step (before the forward pass).  zero: True = clear the gradients, None = only if a
        zero_grad(defer=True) is pending (gradient accumulation), False = leave them."""
        reg = self._perm_jobs
        if zero is None:
            zero = self._zero_due
        self._zero_due = False
        self._began = True
        if reg.due:
            self._refresh_transposed()
        if zero:
            self._zero_gradients()

    def join_prologue(self):
        """The current stream (the one that starts the backward pass) waits for the prologue."""
        if self._zero_due:                       # zero_grad(defer=True) without a begin_step() since
            self._zero_due = False
            self._zero_gradients()
        self._perm_jobs.join()

    def _zero_gradients(self):
        if self.flat_g.is_cuda:
            K.zero_(self.flat_g)                 # (a launch of this library: no ATen kernel inside a captured step)
        else:
            self.flat_g.zero_()

    def param_range(self, module):
        """[lo, hi) of the flat buffers covered by the parameters of `module`, or None if parameters of other modules
        lie inside that range."""
        r = self.param_ranges([module])
        return r[0] if r is not None and len(r) == 1 else None

    def param_ranges(self, modules):
        """Maximal contiguous [lo, hi) ranges of the flat buffers that hold exactly the trainable parameters of `modules`
        (alignment gaps between neighbours are absorbed): the buckets of the data-parallel gradient exchange.  Returns []
        if the modules have no trainable parameters."""
        mine = set()
        for m in modules:
            mine.update(id(p) for p in (m.parameters() if isinstance(m, torch.nn.Module) else [m]))
        order = sorted(zip(self.offsets, self.params), key=lambda t: t[0])
        ranges, cur = [], None
        for i, (o, p) in enumerate(order):
            end = order[i + 1][0] if i + 1 < len(order) else self.numel      # up to the next parameter (absorbs the padding)
            if id(p) in mine:
                if cur is not None and cur[1] == o:
                    cur[1] = end
                else:
                    cur = [o, end]
                    ranges.append(cur)
            else:
                cur = None
        return [tuple(r) for r in ranges]

    def zero_grad(self, set_to_none=False, defer=False):
        """The zero-fill of the flat gradient buffer, in line.  `defer` is accepted from the trainers that clear the gradients AFTER
        the optimiser step (it used to move the fill into the next begin_step(); the fill is one launch either way)."""
        self._zero_due = False
        self._zero_gradients()

    def step(self):
        self.join_prologue()                     # (no-op when the backward pass joined it, as it must)
        K.adam_step(self.flat_p, self.flat_g, self.exp_avg, self.exp_avg_sq, self.shadow, self.state, self.partial, self.lr,
                    self.betas, self.eps, self.grad_norm, self.warmup_steps)
        self._refresh_transposed()           # in line: a captured step leaves the derived copies fresh for its next replay
        self._began = False
        self._touch()

    # -- introspection (host sync; for logging / tests only) -------------------------------------
    def last_stats(self):
        s = self.state.tolist()
        return {"step": int(s[0]), "lr": s[1], "grad_norm": s[2], "clip_coef": s[3]}

    # -- checkpoints: the reference saves `optimizer.state_dict()` of torch.optim.Adam (trainers/base.py:85-105) --------------
    def state_dict(self):
        """torch.optim.Adam's format -- {"state": {i: {step, exp_avg, exp_avg_sq}}, "param_groups": [...]} with i = position
        in model.parameters() -- so that the reference (or a stock torch Adam over the same model) resumes from a checkpoint
        written here and vice versa.  Deviation (documented): ONE step counter for all parameters, so every entry carries the
        same `step`; torch keeps one per parameter, which only differs for parameters that joined training late."""
        step = self.state[0].detach().cpu().clone()
        off_of = {id(p): o for p, o in zip(self.params, self.offsets)}
        state = {}
        if float(step) > 0:
            for i, p in enumerate(self.all_params):
                o = off_of.get(id(p))
                if o is None:
                    continue
                k = p.numel()
                state[i] = {"step": step.clone(), "exp_avg": self.exp_avg[o:o + k].view(p.shape).detach().cpu().clone(),
                            "exp_avg_sq": self.exp_avg_sq[o:o + k].view(p.shape).detach().cpu().clone()}
        k = int(float(step))
        # torch's group["lr"] after k optimiser + k scheduler steps is the value the NEXT step will use (WarmupLR at k + 1)
        w = self.warmup_steps
        next_lr = self.lr * w ** 0.5 * min((k + 1) ** -0.5, (k + 1) * w ** -1.5) if w > 0 else self.lr
        group = {"lr": next_lr, "betas": tuple(self.betas), "eps": self.eps,
                 "weight_decay": 0, "amsgrad": False, "maximize": False, "foreach": None, "capturable": False,
                 "differentiable": False, "fused": None, "initial_lr": self.lr, "params": list(range(len(self.all_params)))}
        return {"state": state, "param_groups": [group],
                "s2svc": {"grad_norm": self.grad_norm, "warmup_steps": self.warmup_steps, "layout": "torch.optim.Adam"}}

    def load_state_dict(self, sd):
        """Accepts torch.optim.Adam state dicts (reference checkpoints) and the flat format of round 1."""
        if "state" in sd and "param_groups" in sd:
            ids = sd["param_groups"][0]["params"]
            if len(ids) != len(self.all_params):
                raise ValueError(f"optimizer state is for {len(ids)} parameters, the model has {len(self.all_params)}")
            pos = {pid: i for i, pid in enumerate(ids)}
            off_of = {id(p): o for p, o in zip(self.params, self.offsets)}
            self.exp_avg.zero_()
            self.exp_avg_sq.zero_()
            steps = set()
            for pid, st in sd["state"].items():
                i = pos[pid]
                p = self.all_params[i]
                o = off_of.get(id(p))
                if o is None:
                    continue            # state of a parameter that is frozen here
                if tuple(st["exp_avg"].shape) != tuple(p.shape):
                    raise ValueError(f"optimizer state {pid}: shape {tuple(st['exp_avg'].shape)} vs parameter {tuple(p.shape)}")
                k = p.numel()
                self.exp_avg[o:o + k].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[o:o + k].copy_(st["exp_avg_sq"].reshape(-1))
                steps.add(int(float(st["step"])))
            if len(steps) > 1:
                import logging
                logging.warning(f"per-parameter Adam steps {sorted(steps)} differ; FlatAdam keeps one counter and resumes at {max(steps)}")
            g = sd["param_groups"][0]
            self.lr = float(g.get("initial_lr", self.lr))
            self.betas, self.eps = tuple(g.get("betas", self.betas)), float(g.get("eps", self.eps))
            extra = sd.get("s2svc", {})
            self.grad_norm = float(extra.get("grad_norm", self.grad_norm))
            self.warmup_steps = float(extra.get("warmup_steps", self.warmup_steps))
            self.state.zero_()
            self.state[0] = float(max(steps)) if steps else 0.0
        else:
            if list(sd.get("offsets", [])) != list(self.offsets) or sd["exp_avg"].numel() != self.numel:
                raise ValueError("flat optimizer state was written for a different parameter layout (frozen modules, fused "
                                 "projections or alignment differ); re-save it in the torch.optim.Adam format")
            self.state.copy_(sd["step"])
            self.exp_avg.copy_(sd["exp_avg"])
            self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.refresh_shadow()
