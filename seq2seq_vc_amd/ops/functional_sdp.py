"""Differentiable building blocks of the stochastic duration predictor (forward and backward are HIP kernels,
csrc/sdp.hip).  Reference: modules/duration_predictor.py:211-304, modules/vits/flow.py, modules/vits/transform.py.

All tensors are channel-last rows r = (b, t); `lens` is the int32 device vector of valid lengths (the reference's
x_mask is t < lens[b]).  The spline and the three "glue" stages are fp32.

The log-determinants of the spline couplings never enter autograd: every coupling of one NLL evaluation adds its
log|det| rows into a running buffer held by a `Shared` object, the tail stage reads the buffers, and -- because
d nll / d lad = -1 for every row, whatever the flow -- the tail's backward leaves the per-utterance gradient
-g[b] in the same object for the spline backward kernels to pick up (autograd runs the tail's backward first: it
is downstream of every coupling).
"""
import torch
from torch.autograd import Function

from . import kernels as K
from . import kernels_sdp as KS
from . import kernels_aas as KA
from .functional import _bias_sink, _c, _emit_vgrad, _emit_wgrad, _reduce_to, _side_run, _slotted


_QUEUE_LN = __import__("os").environ.get("S2SVC_SDP_UNGROUPED", "0") != "1"      # A/B switch (also sdp.py: residual pass-through)


class Shared:
    """Per-call scratch shared by the couplings and the tail of one NLL evaluation."""

    def __init__(self):
        self.lad = {"q": None, "p": None}
        self.neg_g = None


class _MaskRows(Function):
    @staticmethod
    def forward(ctx, x, lens):
        ctx.lens = lens
        return KS.mask_rows(_c(x), lens)

    @staticmethod
    def backward(ctx, dy):
        return KS.mask_rows(_c(dy), ctx.lens), None


def mask_rows(x, lens):
    return _MaskRows.apply(x, lens)


class _Expand(Function):
    """Conv1d(1 -> C, k=1) of a scalar sequence + conditioning, masked (flow.py:290-292, duration_predictor.py:236)."""

    @staticmethod
    def forward(ctx, a, weight, bias, g, lens, out_dtype):
        a = _c(a)
        w = weight.detach().reshape(-1)
        ctx.params = (weight, bias)
        ctx.lens, ctx.has_g = lens, g is not None
        ctx.save_for_backward(a)
        return KS.expand_fwd(a, w, bias.detach() if bias is not None else None, _c(g) if g is not None else None, lens, out_dtype)

    @staticmethod
    def backward(ctx, dy):
        (a,) = ctx.saved_tensors
        weight, bias = ctx.params
        dg, da = KS.expand_bwd(_c(dy), weight.detach().reshape(-1), ctx.lens)
        C = dg.shape[-1]
        w_slot = getattr(weight, "_s2s_grad", None) if weight.requires_grad else None
        b_slot = getattr(bias, "_s2s_grad", None) if (bias is not None and bias.requires_grad) else None
        if _QUEUE_LN and w_slot is not None and b_slot is not None:
            # straight into the gradient slots, queued: joins the grouped column reductions of the batch (was 2 + 2 launches)
            dg2, a1 = dg.view(-1, C), a.view(-1)
            _side_run(lambda: K.colreduce(5, dg2, mean=a1, want_dot=True, out_sum=b_slot.view(-1), out_dot=w_slot.view(-1), accumulate=True),
                      keep=(dg2, a1))
            return (da if ctx.needs_input_grad[0] else None), None, None, (dg if ctx.has_g else None), None, None
        db, dw = K.colreduce(5, dg.view(-1, C), mean=a.view(-1), want_dot=True)
        dw = _emit_vgrad(weight, dw) if weight.requires_grad else None
        db = _emit_vgrad(bias, db) if bias is not None and bias.requires_grad else None
        return (da if ctx.needs_input_grad[0] else None), dw, db, (dg if ctx.has_g else None), None, None


def expand(a, weight, bias, g, lens, out_dtype=torch.float32):
    return _Expand.apply(a, weight, bias, g, lens, out_dtype)


class _LnAct(Function):
    """mask * (res + dropout(act(LayerNorm(x)))) -- one half of a DDS layer (flow.py:148-190)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, act, res, lens, T, p):
        x = _c(x)
        seed = K.new_seed(x.device) if p > 0.0 else (None, 0)
        y, mean, rstd = KS.ln_act_fwd(x, gamma.detach(), beta.detach(), eps, act, _c(res) if res is not None else None, lens, T, p, seed)
        ctx.params = (gamma, beta)
        ctx.meta = (eps, act, lens, T, p, seed, res is not None)
        ctx.save_for_backward(x, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd = ctx.saved_tensors
        gamma, beta = ctx.params
        eps, act, lens, T, p, seed, has_res = ctx.meta
        du, dx, dres = KS.ln_act_bwd(_c(dy), x, mean, rstd, gamma.detach(), beta.detach(), act, lens, T, p, seed, want_dres=has_res)
        D = x.shape[-1]
        du2, x2 = du.view(-1, D), x.view(-1, D)
        if _slotted(gamma, beta) and _QUEUE_LN:       # queued: joins the grouped column reductions of the batch (two launches for up to 24 of them)
            _side_run(lambda: _reduce_to(beta, gamma, 1, du2, x2, mean, rstd), keep=(du2, x2, mean, rstd))
            return dx, None, None, None, None, dres, None, None, None
        dbeta, dgamma = _reduce_to(beta, gamma, 1, du2, x2, mean, rstd)
        return dx, dgamma, dbeta, None, None, dres, None, None, None


def ln_act(x, gamma, beta, eps, act, res=None, lens=None, T=0, p=0.0):
    return _LnAct.apply(x, gamma, beta, eps, act, res, lens, T, p)


_DDS_FUSED = __import__("os").environ.get("S2SVC_DDS_FUSED", "1") != "0"      # A/B switch: one launch per DDS layer and direction


def dds_layer_ok(x, dw, pw):
    """The fused layer kernels take it: fp32 rows on the GPU, kernel size 3, 192 / 256 / 384 / 512 channels, all parameters of the layer
    either in flat-gradient slots or not (the gradient bookkeeping below is written for the two clean cases)."""
    return (_DDS_FUSED and x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and dw.weight.shape[-1] == 3 and
            KS.dds_layer_supported(x.shape[-1], dw.weight.shape[-1]))


class _DDSLayer(Function):
    """x -> mask * (x + dropout(gelu(LN2(conv1x1(gelu(LN1(dwconv_dilated(x)))))))): one DilatedDepthSeparableConv layer
    (flow.py:148-190) on csrc/dds.hip -- 1 launch forward, 1 + the depthwise data gradient backward; the weight-gradient work
    (1x1 weight + bias, the two LayerNorms' gamma / beta, the depthwise weight + bias) is queued off the chain exactly as the
    separate ops queue theirs (_Linear / _LnAct / _DwConv), reading the tensors the fused kernels wrote."""

    @staticmethod
    def forward(ctx, x, dw_w, dw_b, g1, b1, pw_w, pw_b, g2, b2, eps, dil, lens, p):
        x = _c(x)
        C = x.shape[-1]
        seed = K.new_seed(x.device) if p > 0.0 else (None, 0)
        W = pw_w.detach().reshape(C, C)
        out, y1, mean1, rstd1, y2, y3, mean2, rstd2 = KS.dds_layer_fwd(
            x, lens, dil, dw_w.detach().reshape(C, 3), None if dw_b is None else dw_b.detach(), g1.detach(), b1.detach(), W,
            None if pw_b is None else pw_b.detach(), g2.detach(), b2.detach(), eps, p, seed)
        ctx.params = (dw_w, dw_b, g1, b1, pw_w, pw_b, g2, b2)
        ctx.meta = (dil, lens, p, seed)
        ctx.save_for_backward(x, y1, mean1, rstd1, y2, y3, mean2, rstd2)
        return out

    @staticmethod
    def backward(ctx, g):
        x, y1, mean1, rstd1, y2, y3, mean2, rstd2 = ctx.saved_tensors
        dw_w, dw_b, g1, b1, pw_w, pw_b, g2, b2 = ctx.params
        dil, lens, p, seed = ctx.meta
        B, T, C = x.shape
        rows = B * T
        # W^T for the data gradient: a cached copy the optimiser refreshes with the other derived weight copies (one grouped launch)
        Wt = K.gather3_cached(pw_w, (1, C, C), (0, 1, C), 0, torch.float32).view(C, C)
        dres, du2, dy3, du1, dy1 = KS.dds_layer_bwd(_c(g), lens, y3, mean2, rstd2, g2.detach(), b2.detach(), p, seed, Wt, y1, mean1, rstd1,
                                                    g1.detach(), b1.detach())
        # data gradient of the depthwise convolution + the residual path, one launch (dres rides in as `add`)
        dx = None
        if ctx.needs_input_grad[0]:
            if KA.dwconv_add_ok(dy1, 3):
                dx = KA.dwconv(dy1, dw_w.detach(), None, 3, dil, flip=True, add=dres)
            else:
                dx = KA.dwconv(dy1, dw_w.detach(), None, 3, dil, flip=True) + dres
        grads = [None] * 8
        # -- LayerNorm 2 / 1: gamma, beta = column reductions of du * xhat, du
        for (gam, bet, du, xin, mean, rstd, gi) in ((g2, b2, du2, y3, mean2, rstd2, 6), (g1, b1, du1, y1, mean1, rstd1, 2)):
            du_, x_ = du.view(rows, C), xin.view(rows, C)
            if _slotted(gam, bet) and _QUEUE_LN:
                _side_run(lambda bet=bet, gam=gam, du_=du_, x_=x_, mean=mean, rstd=rstd: _reduce_to(bet, gam, 1, du_, x_, mean, rstd),
                          keep=(du_, x_, mean, rstd))
            else:
                dbeta, dgamma = _reduce_to(bet, gam, 1, du_, x_, mean, rstd)
                grads[gi], grads[gi + 1] = dgamma, dbeta
        # -- 1x1 convolution: dW (+)= dy3^T . y2, bias = row sums of dy3^T (fused into the weight-gradient GEMM)
        dy3_, y2_ = dy3.view(rows, C), y2.view(rows, C)
        if pw_w.requires_grad:
            tile, sk = K.plan_gemm(C, C, rows)
            rs, racc, db = _bias_sink(pw_b, C)

            def wr(out, acc):
                K.gemm(K.operand(dy3_, C, layout=K.RC), K.operand(y2_, C, layout=K.RC), C, C, rows, out, in_dtype=torch.float32, splitk=sk,
                       tile=tile, accumulate=acc, a_rowsum=rs, a_rowsum_accumulate=racc)
            if _slotted(pw_w, pw_b):
                _side_run(lambda: _emit_wgrad(pw_w, (C, C), wr), keep=(dy3_, y2_))
            else:
                dw = _emit_wgrad(pw_w, (C, C), wr)
                grads[4] = dw.view(pw_w.shape) if dw is not None else None
                grads[5] = db
        elif pw_b is not None and pw_b.requires_grad:
            grads[5], _ = _reduce_to(pw_b, None, 0, dy3_)
        # -- depthwise convolution: weight / bias gradients from (x, dy1), as _DwConv does
        if dw_w.requires_grad:
            slot = getattr(dw_w, "_s2s_grad", None)
            if slot is not None and slot.is_contiguous():
                KA.dwconv_wgrad(x, dy1, 3, dil, out=slot)
            else:
                grads[0] = _emit_vgrad(dw_w, KA.dwconv_wgrad(x, dy1, 3, dil))
        if dw_b is not None and dw_b.requires_grad:
            dy1_ = dy1.view(rows, C)
            if _slotted(dw_b):
                _side_run(lambda: _reduce_to(dw_b, None, 0, dy1_), keep=(dy1_,))
            else:
                grads[1], _ = _reduce_to(dw_b, None, 0, dy1_)
        return (dx, *grads, None, None, None, None)


def dds_layer(x, dw, ln1, pw, ln2, lens, p):
    """One DDS layer on the fused kernels (modules: depthwise Conv1d, LayerNorm, 1x1 Conv1d, LayerNorm of the reference's Sequential)."""
    return _DDSLayer.apply(x, dw.weight, dw.bias, ln1.weight, ln1.bias, pw.weight, pw.bias, ln2.weight, ln2.bias, ln1.eps, dw.dilation[0],
                           lens, p)


class _Spline(Function):
    """xb' = RQ-spline(xb; h) on valid rows (ConvFlow, flow.py:294-308); log|det| rows go to shared.lad[which]."""

    @staticmethod
    def forward(ctx, xb, h, hscale, bound, lens, shared, which):
        xb, h = _c(xb), _c(h)
        buf = shared.lad[which]
        out, lad = KS.rq_spline_fwd(xb, h, hscale, bound, lens, inverse=False, lad=buf, accumulate=buf is not None)
        shared.lad[which] = lad
        ctx.meta = (hscale, bound, lens, shared)
        ctx.save_for_backward(xb, h)
        return out

    @staticmethod
    def backward(ctx, g_out):
        xb, h = ctx.saved_tensors
        hscale, bound, lens, shared = ctx.meta
        if shared.neg_g is None:
            raise RuntimeError("spline backward before the NLL tail's backward (the tail publishes d nll / d logdet)")
        dx, dh = KS.rq_spline_bwd(xb, h, hscale, bound, lens, _c(g_out), shared.neg_g)
        return dx, dh, None, None, None, None, None


def spline(xb, h, hscale, bound, lens, shared, which):
    return _Spline.apply(xb, h, hscale, bound, lens, shared, which)


def _param_grads(m, logs, part):
    """part (B, 4) = per-utterance [dm0, dm1, dlogs0, dlogs1] -> gradients of the (2,1) parameters."""
    s, _ = K.colreduce(0, part)
    dm = _emit_vgrad(m, s[:2]) if m.requires_grad else None
    dl = _emit_vgrad(logs, s[2:]) if logs.requires_grad else None
    return dm, dl


class _Head(Function):
    """noise -> first ElementwiseAffine of the posterior flows (duration_predictor.py:239-245)."""

    @staticmethod
    def forward(ctx, noise, m, logs, lens):
        ctx.params = (m, logs)
        ctx.lens = lens
        ctx.save_for_backward(noise)
        z0, z1 = KS.sdp_head_fwd(noise, lens, m.detach().reshape(-1), logs.detach().reshape(-1))
        return z0, z1

    @staticmethod
    def backward(ctx, dz0, dz1):
        (noise,) = ctx.saved_tensors
        m, logs = ctx.params
        part = KS.sdp_head_bwd(noise, ctx.lens, logs.detach().reshape(-1), _c(dz0), _c(dz1))
        dm, dl = _param_grads(m, logs, part)
        return None, dm, dl, None


def head(noise, m, logs, lens):
    return _Head.apply(noise, m, logs, lens)


class _Mid(Function):
    """Variational dequantisation + LogFlow + first ElementwiseAffine of the prior flows (duration_predictor.py:249-262)."""

    @staticmethod
    def forward(ctx, zu, z1, w, m, logs, lens):
        zu, z1, w = _c(zu), _c(z1), _c(w)
        ctx.params = (m, logs)
        ctx.lens = lens
        ctx.save_for_backward(zu, z1, w)
        return KS.sdp_mid_fwd(zu, z1, w, lens, m.detach().reshape(-1), logs.detach().reshape(-1))

    @staticmethod
    def backward(ctx, dy0, dy1, dlz):
        zu, z1, w = ctx.saved_tensors
        m, logs = ctx.params
        dzu, dz1, part = KS.sdp_mid_bwd(zu, z1, w, ctx.lens, logs.detach().reshape(-1), _c(dy0), _c(dy1), _c(dlz))
        dm, dl = _param_grads(m, logs, part)
        return dzu, dz1, None, dm, dl, None


def mid(zu, z1, w, m, logs, lens):
    return _Mid.apply(zu, z1, w, m, logs, lens)


class _Tail(Function):
    """Per-utterance nll + logq (duration_predictor.py:246-280)."""

    @staticmethod
    def forward(ctx, noise, zu, lz, af, bf, logs_q, logs_p, lens, shared):
        zu, lz, af, bf = _c(zu), _c(lz), _c(af), _c(bf)
        ctx.params = (logs_q, logs_p)
        ctx.meta = (lens, shared)
        ctx.save_for_backward(zu, af, bf)
        return KS.sdp_tail_fwd(noise, lens, zu, lz, shared.lad["q"], shared.lad["p"], af, bf, logs_q.detach().reshape(-1),
                               logs_p.detach().reshape(-1))

    @staticmethod
    def backward(ctx, g):
        zu, af, bf = ctx.saved_tensors
        logs_q, logs_p = ctx.params
        lens, shared = ctx.meta
        d_af, d_bf, d_lz, d_zu, neg_g, part = KS.sdp_tail_bwd(_c(g), lens, zu, af, bf)
        shared.neg_g = neg_g
        s, _ = K.colreduce(0, part)                       # (2,): the same scalar for both entries of a logs parameter
        dq = _emit_vgrad(logs_q, s) if logs_q.requires_grad else None
        dp = _emit_vgrad(logs_p, s.clone()) if logs_p.requires_grad else None   # never hand one tensor to two leaves
        return None, d_zu, d_lz, d_af, d_bf, dq, dp, None, None


def tail(noise, zu, lz, af, bf, logs_q, logs_p, lens, shared):
    return _Tail.apply(noise, zu, lz, af, bf, logs_q, logs_p, lens, shared)
