"""Differentiable AAS-VC / Conformer specific ops (forward and backward are HIP kernels)."""
import torch
from torch.autograd import Function

from . import kernels as K
from . import kernels_aas as KA
from .functional import _c, _emit_vgrad, _reduce_to, _side_run, _slotted


class _DwConv(Function):
    """Depthwise Conv1d on channel-last activations (conformer/convolution.py:42-51; vits/flow.py:137-146)."""

    @staticmethod
    def forward(ctx, x, weight, bias, dil, passthrough=False):
        """passthrough: also return x itself as a second output.  A caller whose x feeds BOTH this convolution and a residual
        connection uses the second output for the residual: x then has one consumer, and the two gradients are summed inside the
        data-gradient kernel (`add`) instead of by a separate element-wise add of the autograd engine."""
        x = _c(x)
        ks = weight.shape[-1]
        y = KA.dwconv(x, weight.detach(), bias.detach() if bias is not None else None, ks, dil)
        ctx.params = (weight, bias)
        ctx.meta = (ks, dil)
        ctx.save_for_backward(x)
        if passthrough:
            ctx.set_materialize_grads(False)
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, g_pass=None):
        if dy is None:                 # only the pass-through output was used
            return g_pass, None, None, None, None
        (x,) = ctx.saved_tensors
        weight, bias = ctx.params
        ks, dil = ctx.meta
        dx, dw, db = dwconv_backward(x, _c(dy), weight, bias, ks, dil, g_pass, ctx.needs_input_grad[0])
        return dx, dw, db, None, None


def dwconv_backward(x, dy, weight, bias, ks, dil, g_pass, need_dx):
    """Data / weight / bias gradients of the depthwise convolution (shared with the fused DDS half-layer, functional_sdp._DwLnAct);
    g_pass: a gradient for x that arrived over a pass-through output, summed inside the data-gradient kernel."""
    dx = None
    if need_dx:
        if g_pass is not None and KA.dwconv_add_ok(dy, ks):
            dx = KA.dwconv(dy, weight.detach(), None, ks, dil, flip=True, add=_c(g_pass))
        else:
            dx = KA.dwconv(dy, weight.detach(), None, ks, dil, flip=True)
            if g_pass is not None:
                dx = dx + g_pass
    dw = db = None
    if weight.requires_grad:
        slot = getattr(weight, "_s2s_grad", None)
        if slot is not None and slot.is_contiguous():       # straight into the flat gradient buffer
            KA.dwconv_wgrad(x, dy, ks, dil, out=slot)
        else:
            dw = _emit_vgrad(weight, KA.dwconv_wgrad(x, dy, ks, dil))
    if bias is not None and bias.requires_grad:
        dy2 = dy.view(-1, dy.shape[-1])
        if _slotted(bias):                                    # queued: joins the grouped column reductions of the batch
            _side_run(lambda: _reduce_to(bias, None, 0, dy2), keep=(dy2,))
        else:
            db, _ = _reduce_to(bias, None, 0, dy2)
    return dx, dw, db


def dwconv1d(x, weight, bias=None, dilation=1):
    return _DwConv.apply(x, weight, bias, dilation)


def dwconv1d_pass(x, weight, bias=None, dilation=1):
    """-> (dwconv(x), x): use the second output wherever x is needed again (residual connection)."""
    return _DwConv.apply(x, weight, bias, dilation, True)


class _ConvModCore(Function):
    """GLU -> depthwise conv -> BatchNorm1d (batch statistics) -> Swish of the Conformer convolution module as ONE autograd node on the
    fused bf16 kernels of csrc/convmod.hip (convolution.py:68-75): 3 launches forward, 3 backward (+ 1 off the chain), where the
    separate ops take 5 and 8; only y2 and z are kept for the backward pass (g = glu(y2) and the pre-activation are recomputed)."""

    @staticmethod
    def forward(ctx, y2, dw_weight, dw_bias, gamma, beta, run_mean, run_var, num_batches, eps, momentum, vlens=None):
        y2 = _c(y2)
        ks = dw_weight.shape[-1]
        z, mean, rstd = KA.convmod_fwd(y2, dw_weight.detach(), None if dw_bias is None else dw_bias.detach(), ks, eps, momentum,
                                       run_mean, run_var, num_batches, vlens=vlens)
        out = KA.bn_swish_apply(z, mean, rstd, gamma.detach(), beta.detach(), vlens=vlens)
        ctx.params = (dw_weight, dw_bias, gamma, beta)
        ctx.ks = ks
        ctx.vlens = vlens
        ctx.save_for_backward(y2, z, mean, rstd)
        return out

    @staticmethod
    def backward(ctx, da):
        y2, z, mean, rstd = ctx.saved_tensors
        dw_weight, dw_bias, gamma, beta = ctx.params
        ks = ctx.ks
        C = z.shape[-1]
        g_slot = getattr(gamma, "_s2s_grad", None) if gamma.requires_grad else None
        b_slot = getattr(beta, "_s2s_grad", None) if beta.requires_grad else None
        both = g_slot is not None and b_slot is not None
        dy2, sdy, sdyx, (ws_w, chunks) = KA.convmod_bwd(_c(da), z, y2, dw_weight.detach(), mean, rstd, gamma.detach(), beta.detach(), ks,
                                                        g_slot.view(-1) if both else None, b_slot.view(-1) if both else None,
                                                        vlens=ctx.vlens)
        dgamma = dbeta = None
        if not both:
            dgamma = _emit_vgrad(gamma, sdyx) if gamma.requires_grad else None
            dbeta = _emit_vgrad(beta, sdy) if beta.requires_grad else None
        dw = db = None
        need_w = dw_weight.requires_grad
        need_b = dw_bias is not None and dw_bias.requires_grad
        if need_w or need_b:
            w_slot = getattr(dw_weight, "_s2s_grad", None) if need_w else None
            bi_slot = getattr(dw_bias, "_s2s_grad", None) if need_b else None
            if (not need_w or (w_slot is not None and w_slot.is_contiguous())) and (not need_b or bi_slot is not None):
                # straight into the flat gradient buffer, off the data-gradient chain
                _side_run(lambda: KA.convmod_wgrad_final(ws_w, chunks, C, ks, w_slot if need_w else None,
                                                         bi_slot.view(-1) if need_b else None, accumulate=True), keep=(ws_w,))
            else:
                dwv, dbv = KA.convmod_wgrad_final(ws_w, chunks, C, ks)
                dw = _emit_vgrad(dw_weight, dwv) if need_w else None
                db = _emit_vgrad(dw_bias, dbv) if need_b else None
        return dy2, dw, db, dgamma, dbeta, None, None, None, None, None, None


def convmod_core(y2, dw_weight, dw_bias, gamma, beta, run_mean, run_var, num_batches, eps=1e-5, momentum=0.1, vlens=None):
    """swish(batch_norm(dwconv1d(glu(y2)))) in training mode on the fused kernels; use convmod_core_ok() first.
    vlens (B int32, device): frames t >= vlens[b] are absent (captured steps on batches shorter than their padded shape)."""
    return _ConvModCore.apply(y2, dw_weight, dw_bias, gamma, beta, run_mean, run_var, num_batches, eps, momentum, vlens)


def convmod_core_ok(y2, dw_weight, training, activation):
    import os
    return (training and activation == "swish" and y2.dtype == torch.bfloat16 and os.environ.get("S2SVC_NO_CONVMOD", "0") != "1"
            and dw_weight.shape[1] == 1 and KA.convmod_supported(dw_weight.shape[0], dw_weight.shape[-1]))


class _PairwiseLogSoftmax(Function):
    """log_p_attn[b,i,:] = log_softmax_j(-||feats[b,i]-text[b,j]||_2), padded text columns -inf
    (modules/alignments.py:51-59)."""

    @staticmethod
    def forward(ctx, feats, text, text_lens_i32):
        feats, text = _c(feats), _c(text)
        logp, dist = KA.pairwise_l2_logsoftmax(feats, text, text_lens_i32)
        ctx.save_for_backward(feats, text, text_lens_i32, logp, dist)
        return logp

    @staticmethod
    def backward(ctx, dlogp):
        feats, text, text_lens_i32, logp, dist = ctx.saved_tensors
        B, Tf, A = feats.shape
        Tx = text.shape[1]
        dtype = feats.dtype
        G, rowsum = KA.pairwise_l2_bwd_g(logp, dist, _c(dlogp.float()), text_lens_i32, dtype)
        # dfeats[b,i,:] = feats[b,i,:]*sum_j G[b,i,j] - sum_j G[b,i,j] text[b,j,:]
        r1 = KA.rowscale(feats, rowsum.view(-1))
        dfe = torch.empty_like(feats)
        K.gemm(K.operand(G, Tx, bs0=Tf * Tx), K.operand(text, A, layout=K.RC, bs0=Tx * A), Tf, A, Tx, dfe, in_dtype=dtype,
               nb0=B, nb1=1, cbs=(Tf * A, 0), alpha=-1.0, res=r1, rbs=(Tf * A, 0))
        # dtext[b,j,:] = text[b,j,:]*sum_i G[b,i,j] - sum_i G[b,i,j] feats[b,i,:]
        # column sums over the frame axis, one deterministic reduction per utterance: all of them as ONE grouped launch pair
        cs = K.zeros((B, Tx), torch.float32, feats.device)
        queue = []
        with K.record_colreduce(queue):
            for b in range(B):
                K.colreduce(0, G[b], out_sum=cs[b], accumulate=True)
        K.flush_colreduce(queue)
        r2 = KA.rowscale(text, cs.view(-1))
        dtx = torch.empty_like(text)
        K.gemm(K.operand(G, Tx, layout=K.RC, bs0=Tf * Tx), K.operand(feats, A, layout=K.RC, bs0=Tf * A), Tx, A, Tf, dtx,
               in_dtype=dtype, nb0=B, nb1=1, cbs=(Tx * A, 0), alpha=-1.0, res=r2, rbs=(Tx * A, 0))
        return dfe, dtx, None


def pairwise_logsoftmax(feats, text, text_lens_i32):
    return _PairwiseLogSoftmax.apply(feats, text, text_lens_i32)


class _GaussUpsample(Function):
    """hs_up = softmax_j(-delta (t - c_j)^2) @ hs   (modules/length_regulator.py:111-154); ds carries no grad."""

    @staticmethod
    def forward(ctx, hs, ds, text_lens_i32, feat_lens_i32, Tf, delta):
        hs = _c(hs)
        B, Tx, A = hs.shape
        dtype = hs.dtype
        P = KA.gauss_upsample_probs(_c(ds.float()), text_lens_i32, feat_lens_i32, Tf, dtype, delta)
        out = torch.empty((B, Tf, A), dtype=dtype, device=hs.device)
        K.gemm(K.operand(P, Tx, bs0=Tf * Tx), K.operand(hs, A, layout=K.RC, bs0=Tx * A), Tf, A, Tx, out, in_dtype=dtype, nb0=B,
               nb1=1, cbs=(Tf * A, 0))
        ctx.save_for_backward(P)
        ctx.dims = (B, Tf, Tx, A)
        return out

    @staticmethod
    def backward(ctx, dout):
        (P,) = ctx.saved_tensors
        B, Tf, Tx, A = ctx.dims
        dout = _c(dout)
        dhs = torch.empty((B, Tx, A), dtype=dout.dtype, device=dout.device)
        K.gemm(K.operand(P, Tx, layout=K.RC, bs0=Tf * Tx), K.operand(dout, A, layout=K.RC, bs0=Tf * A), Tx, A, Tf, dhs,
               in_dtype=dout.dtype, nb0=B, nb1=1, cbs=(Tx * A, 0))
        return dhs, None, None, None, None, None


def gaussian_upsample(hs, ds, text_lens_i32, feat_lens_i32, Tf, delta=0.1):
    return _GaussUpsample.apply(hs, ds, text_lens_i32, feat_lens_i32, Tf, delta)


class _ForwardSum(Function):
    @staticmethod
    def forward(ctx, log_p_attn, prior, text_lens_i32, feat_lens_i32, blank_prob):
        lp = _c(log_p_attn.float())
        loss_b, grad = KA.forward_sum(lp, prior, text_lens_i32, feat_lens_i32, blank_prob)
        ctx.save_for_backward(grad)
        ctx.in_dtype = log_p_attn.dtype
        return K.weighted_sum([(loss_b, 1.0 / lp.shape[0])])

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        # d loss / d log_p_attn = g * grad  (the scalar g lives on the device: broadcast it as a row scale)
        (gs,) = K.weighted_sum_bwd(_c(g.float()), [((grad.shape[0] * grad.shape[1],), 1.0)], g.device)
        out = KA.rowscale(grad.view(-1, grad.shape[-1]), gs).view_as(grad)
        return out.to(ctx.in_dtype), None, None, None, None


def forward_sum_loss(log_p_attn, prior, text_lens_i32, feat_lens_i32, blank_prob):
    return _ForwardSum.apply(log_p_attn, prior, text_lens_i32, feat_lens_i32, blank_prob)


class _ForwardSumPrefetched(Function):
    """The same loss from per-utterance losses / the gradient that losses.ForwardSumLoss.prefetch computed on the auxiliary
    stream beside the decoder (same kernels, same inputs: same bits)."""

    @staticmethod
    def forward(ctx, log_p_attn, loss_b, grad):
        ctx.save_for_backward(grad)
        ctx.in_dtype = log_p_attn.dtype
        return K.weighted_sum([(_c(loss_b.float()), 1.0 / log_p_attn.shape[0])])

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        (gs,) = K.weighted_sum_bwd(_c(g.float()), [((grad.shape[0] * grad.shape[1],), 1.0)], g.device)      # g on every row
        out = KA.rowscale(grad.view(-1, grad.shape[-1]), gs).view_as(grad)
        return out.to(ctx.in_dtype), None, None


def forward_sum_loss_prefetched(log_p_attn, loss_b, grad):
    return _ForwardSumPrefetched.apply(log_p_attn, loss_b, grad)


class _InterpNearest(Function):
    @staticmethod
    def forward(ctx, x, Tout, ext_in=None, ext_out=None):
        x = _c(x)
        ctx.Tin = x.shape[1]
        ctx.ext = (ext_in, ext_out)
        return K.interp_nearest(x, Tout, ext_in, ext_out)

    @staticmethod
    def backward(ctx, dy):
        return K.interp_nearest_bwd(_c(dy), ctx.Tin, *ctx.ext), None, None, None


def interp_nearest(x, Tout, ext_in=None, ext_out=None):
    """F.interpolate(x^T, size=Tout)^T per batch item on channel-last (B, T, C)  (models/aas_vc.py:340-349).
    ext_in / ext_out: the cropped lengths as graph data when the tensors are padded (captured steps), see kernels.interp_nearest."""
    return _InterpNearest.apply(x, Tout, ext_in, ext_out)


class _LengthRegulate(Function):
    @staticmethod
    def forward(ctx, x, ds_i32, Tout, pad_value):
        x = _c(x)
        start, idx, _ = KA.length_regulate_index(ds_i32, Tout)
        ctx.save_for_backward(start, ds_i32)
        ctx.Tx = x.shape[1]
        return KA.length_regulate_fwd(x, idx.contiguous(), Tout, pad_value)

    @staticmethod
    def backward(ctx, dy):
        start, ds_i32 = ctx.saved_tensors
        return KA.length_regulate_bwd(_c(dy), start, ds_i32, ctx.Tx), None, None, None


def length_regulate(x, ds_i32, Tout, pad_value=0.0):
    """Repeat frame i of utterance b ds[b, i] times along time, zero-pad to Tout (modules/length_regulator.py:67-97).
    x (B,Tx,D) compute dtype, ds_i32 (B,Tx) int32 on the device; Tout is a host int (the caller knows the durations on the
    host: they come from the collater / from `.tolist()` of the predicted durations, as in the reference's pad_list)."""
    return _LengthRegulate.apply(x, ds_i32, Tout, pad_value)
