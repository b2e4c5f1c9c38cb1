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
from .functional import _c, _emit_vgrad, _reduce_to, _side_run, _slotted


_QUEUE_LN = True        # LayerNorm parameter gradients of the duration predictor join the grouped column reductions of their batch


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




class _DwLnAct(Function):
    """act(LayerNorm(depthwise_conv1d(x))) -- the first half of a DDS layer (flow.py:137-160) as ONE forward launch
    (s2svc_dw_ln_act_fwd); the backward pass is the two modular ones in sequence (LayerNorm', then the convolution's gradients).
    Second output: x itself (the residual's pass-through, see functional_aas._DwConv)."""

    @staticmethod
    def forward(ctx, x, dw_w, dw_b, dil, gamma, beta, eps, act):
        x = _c(x)
        ks = dw_w.shape[-1]
        u, y, mean, rstd = KS.dw_ln_act_fwd(x, dw_w.detach(), dw_b.detach() if dw_b is not None else None, ks, dil, gamma.detach(),
                                            beta.detach(), eps, act)
        ctx.params = (dw_w, dw_b, gamma, beta)
        ctx.meta = (ks, dil, act)
        ctx.save_for_backward(x, u, mean, rstd)
        ctx.set_materialize_grads(False)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, g_pass=None):
        if dy is None:
            return g_pass, None, None, None, None, None, None, None
        from .functional_aas import dwconv_backward
        x, u, mean, rstd = ctx.saved_tensors
        dw_w, dw_b, gamma, beta = ctx.params
        ks, dil, act = ctx.meta
        du, dxu, _ = KS.ln_act_bwd(_c(dy), u, mean, rstd, gamma.detach(), beta.detach(), act)
        D = u.shape[-1]
        du2, u2 = du.view(-1, D), u.view(-1, D)
        dgamma = dbeta = None
        if _slotted(gamma, beta) and _QUEUE_LN:
            _side_run(lambda: _reduce_to(beta, gamma, 1, du2, u2, mean, rstd), keep=(du2, u2, mean, rstd))
        else:
            dbeta, dgamma = _reduce_to(beta, gamma, 1, du2, u2, mean, rstd)
        dx, dw, db = dwconv_backward(x, dxu, dw_w, dw_b, ks, dil, g_pass, ctx.needs_input_grad[0])
        return dx, dw, db, None, dgamma, dbeta, None, None


def dw_ln_act_ok(x, dw_w):
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and x.shape[-1] <= 512 and dw_w.shape[-1] % 2 == 1
            and dw_w.dtype == torch.float32)


def dw_ln_act(x, dw_w, dw_b, dil, gamma, beta, eps, act):
    """-> (act(LN(dwconv(x))), x): the second output is x for the residual connection (one consumer of x)."""
    return _DwLnAct.apply(x, dw_w, dw_b, dil, gamma, beta, eps, act)


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
    def forward(ctx, noise, zu, lz, af, bf, logs_q, logs_p, lens, shared, normalize=False):
        zu, lz, af, bf = _c(zu), _c(lz), _c(af), _c(bf)
        ctx.params = (logs_q, logs_p)
        ctx.meta = (lens, shared, normalize)
        ctx.save_for_backward(zu, af, bf)
        return KS.sdp_tail_fwd(noise, lens, zu, lz, shared.lad["q"], shared.lad["p"], af, bf, logs_q.detach().reshape(-1),
                               logs_p.detach().reshape(-1), normalize)

    @staticmethod
    def backward(ctx, g):
        zu, af, bf = ctx.saved_tensors
        logs_q, logs_p = ctx.params
        lens, shared, normalize = ctx.meta
        d_af, d_bf, d_lz, d_zu, neg_g, part = KS.sdp_tail_bwd(_c(g), lens, zu, af, bf, normalize)
        shared.neg_g = neg_g
        s, _ = K.colreduce(0, part)                       # (2,): the same scalar for both entries of a logs parameter
        dq = _emit_vgrad(logs_q, s) if logs_q.requires_grad else None
        dp = _emit_vgrad(logs_p, K.axpby(1.0, s)) if logs_p.requires_grad else None   # (a copy: never hand one tensor to two leaves)
        return None, d_zu, d_lz, d_af, d_bf, dq, dp, None, None, None


def tail(noise, zu, lz, af, bf, logs_q, logs_p, lens, shared, normalize=False):
    """normalize: the per-utterance NLL divided by the number of non-padded text positions of the batch (models/aas_vc.py:403)."""
    return _Tail.apply(noise, zu, lz, af, bf, logs_q, logs_p, lens, shared, normalize)
