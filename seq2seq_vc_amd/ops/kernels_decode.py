"""Launchers for the autoregressive decode-step kernels (csrc/decode.hip; C-ABI in include/s2svc_hip.h)."""
import ctypes

import torch

from .. import _lib
from .kernels import _DT, ACT, dt, operand, ptr, stream


def _p(t, off=0):
    return None if t is None else t.data_ptr() + off * t.element_size()


def decode_posenc(x, xscale, alpha, pe, pos, y):
    B, D = x.shape
    _lib.check(_lib.lib().s2svc_decode_posenc(dt(x), B, D, ptr(x), xscale, ptr(alpha), ptr(pe), ptr(pos), ptr(y), stream()),
               "decode_posenc")
    return y


def decode_attn(q, q_off, ldq, kc, k_off, vc, v_off, ldt, cbs, new, knew_off, vnew_off, ldn, pos, klen, Tk, scale, ctx, B, H, dk,
                att=None, att_strides=(0, 0, 0)):
    """q/kc/vc/new are tensors, *_off element offsets into them (packed projections); see s2svc_decode_attn."""
    _lib.check(_lib.lib().s2svc_decode_attn(dt(ctx), B, H, dk, _p(q, q_off), ldq, _p(kc, k_off), _p(vc, v_off), ldt, cbs,
                                            _p(new, knew_off) if new is not None else None,
                                            _p(new, vnew_off) if new is not None else None, ldn, ptr(pos), ptr(klen), Tk, scale,
                                            ptr(ctx), ctx.shape[-1], ptr(att), att_strides[0], att_strides[1], att_strides[2],
                                            stream()), "decode_attn")
    return ctx


def decode_emit(feat, logit, r, odim, threshold, minlen, maxlen, pos, outs, probs, prev, stop_at):
    B = feat.shape[0]
    _lib.check(_lib.lib().s2svc_decode_emit(dt(feat), B, r, odim, ptr(feat), ptr(logit), threshold, ptr(minlen), ptr(maxlen),
                                            ptr(pos), ptr(outs), outs.stride(0), ptr(probs), probs.stride(0), ptr(prev),
                                            ptr(stop_at), stream()), "decode_emit")


def decode_advance(pos, seed_base_ptr=None, seed_stride=0):
    _lib.check(_lib.lib().s2svc_decode_advance(ptr(pos), seed_base_ptr, seed_stride, stream()), "decode_advance")


def decode_emit_advance(out, r, odim, threshold, minlen, maxlen, pos, outs, probs, prev, stop_at, seed_base_ptr, seed_stride, ticket,
                        pe=None, alpha=None, pe_next=None):
    """decode_emit + decode_advance (+ the next position's positional row -> pe_next) in one launch (csrc/decode_fused.hip).
    out (B, r * odim + r): the packed feat_out | prob_out projection of this position."""
    B = out.shape[0]
    D, rows = (pe.shape[1], pe.shape[0]) if pe_next is not None else (0, 0)
    _lib.check(_lib.lib().s2svc_decode_emit_advance(dt(out), B, r, odim, ptr(out), _p(out, r * odim), out.stride(0), threshold, ptr(minlen),
                                                    ptr(maxlen), ptr(pos), ptr(outs), outs.stride(0), ptr(probs), probs.stride(0),
                                                    ptr(prev), ptr(stop_at), seed_base_ptr, seed_stride, ptr(ticket),
                                                    ptr(pe) if pe_next is not None else None, ptr(alpha), D, rows, ptr(pe_next), stream()),
               "decode_emit_advance")


def ln_linear_supported(dtype, M, K):
    """True if the fused LayerNorm + skinny projection kernel takes an (M, K) input (decode.py falls back to LayerNorm + GEMM)."""
    return bool(_lib.lib().s2svc_decode_ln_linear_supported(_DT[dtype], M, K))


def ln_linear(x, w, bias, *, norm=None, act=None, res=None, y_out=None, drop_p=0.0, seed=(None, 0)):
    """out = act(LN(x) . w^T + bias) [dropout] (+ res) in ONE launch; norm = (gamma, beta, eps) or None (plain linear);
    y_out: a tensor that receives LN(x) as well.  x (M <= 64, K), w (N, K) in the compute dtype; bias / gamma / beta fp32."""
    M, Kd = x.shape
    N = w.shape[0]
    out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    d = _lib.GemmDesc()
    d.A, d.B = operand(x, x.stride(0)), operand(w, w.stride(0))
    d.C, d.ldc, d.c_dtype = out.data_ptr(), N, dt(out)
    d.bias, d.res, d.ldr = ptr(bias), ptr(res), (res.stride(0) if res is not None else N)
    d.M, d.N, d.K, d.nb0, d.nb1 = M, N, Kd, 1, 1
    d.act, d.alpha, d.dtype, d.splitk = ACT[act], 1.0, _DT[x.dtype], 1
    if drop_p > 0.0:
        d.drop_p, d.seed_base, d.seed_off = drop_p, seed[0], seed[1]
    g, b, eps = (None, None, 0.0) if norm is None else norm
    _lib.check(_lib.lib().s2svc_decode_ln_linear(ctypes.byref(d), ptr(g), ptr(b), float(eps), ptr(y_out),
                                                 y_out.stride(0) if y_out is not None else 0, stream()), "decode_ln_linear")
    return out
