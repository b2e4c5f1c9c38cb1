"""Launchers for the stochastic-duration-predictor kernels (csrc/sdp.hip; C-ABI in include/s2svc_hip.h)."""
import torch

from .. import _lib
from .kernels import ACT, dt, ptr, stream


def _f32(*ts):
    for t in ts:
        if t is not None and t.dtype != torch.float32:
            raise TypeError("SDP spline / glue kernels are fp32")


def mask_rows(x, lens):
    B, T, C = x.shape
    y = torch.empty_like(x)
    _lib.check(_lib.lib().s2svc_mask_rows(dt(x), B, T, C, ptr(x), ptr(lens), ptr(y), stream()), "mask_rows")
    return y


def expand_fwd(a, w, bias, g, lens, out_dtype):
    """a (B,T) fp32, w / bias (C) fp32, g (B,T,C) or None -> (B,T,C)."""
    _f32(a, w, bias)
    B, T = a.shape
    C = w.numel()
    y = torch.empty((B, T, C), dtype=out_dtype, device=a.device)
    _lib.check(_lib.lib().s2svc_expand_fwd(dt(y), B, T, C, ptr(a), ptr(w), ptr(bias), ptr(g), ptr(lens), ptr(y), stream()), "expand_fwd")
    return y


def expand_bwd(dy, w, lens):
    B, T, C = dy.shape
    dg = torch.empty_like(dy)
    da = torch.empty((B, T), dtype=torch.float32, device=dy.device)
    _lib.check(_lib.lib().s2svc_expand_bwd(dt(dy), B, T, C, ptr(dy), ptr(w), ptr(lens), ptr(dg), ptr(da), stream()), "expand_bwd")
    return dg, da


def ln_act_fwd(x, gamma, beta, eps, act, res=None, lens=None, T=0, p=0.0, seed=(None, 0)):
    D = x.shape[-1]
    rows = x.numel() // D
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().s2svc_ln_act_fwd(dt(x), rows, D, T, ptr(x), ptr(gamma), ptr(beta), eps, ACT[act], ptr(res), ptr(lens), p,
                                           seed[0], seed[1], ptr(y), ptr(mean), ptr(rstd), stream()), "ln_act_fwd")
    return y, mean, rstd


def dw_ln_act_fwd(x, dw_w, dw_b, ks, dil, gamma, beta, eps, act):
    """x (B, T, D) fp32 -> (u = dwconv(x), y = act(LN(u)), mean, rstd): the first half of a DDS layer in one launch."""
    _f32(x, dw_w, dw_b, gamma, beta)
    B, T, D = x.shape
    u, y = torch.empty_like(x), torch.empty_like(x)
    mean = torch.empty(B * T, dtype=torch.float32, device=x.device)
    rstd = torch.empty(B * T, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().s2svc_dw_ln_act_fwd(B, T, D, ks, dil, ptr(x), ptr(dw_w), ptr(dw_b), ptr(gamma), ptr(beta), eps, ACT[act], ptr(u),
                                              ptr(y), ptr(mean), ptr(rstd), stream()), "dw_ln_act_fwd")
    return u, y, mean, rstd


def ln_act_bwd(dy, x, mean, rstd, gamma, beta, act, lens=None, T=0, p=0.0, seed=(None, 0), want_dres=False):
    D = x.shape[-1]
    rows = x.numel() // D
    du, dx = torch.empty_like(x), torch.empty_like(x)
    dres = torch.empty_like(x) if want_dres else None
    _lib.check(_lib.lib().s2svc_ln_act_bwd(dt(x), rows, D, T, ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ACT[act],
                                           ptr(lens), p, seed[0], seed[1], ptr(du), ptr(dx), ptr(dres), stream()), "ln_act_bwd")
    return du, dx, dres


def rq_spline_fwd(x, h, hscale, bound, lens, inverse=False, lad=None, accumulate=False):
    """x (B,T), h (B,T,3*bins-1) fp32 -> (out (B,T), lad (B,T)); `lad` may be a running buffer (accumulate=True)."""
    _f32(x, h, lad)
    B, T = x.shape
    bins = (h.shape[-1] + 1) // 3
    out = torch.empty_like(x)
    if lad is None:
        lad, accumulate = torch.empty_like(x), False
    _lib.check(_lib.lib().s2svc_rq_spline_fwd(B, T, bins, ptr(x), ptr(h), hscale, bound, ptr(lens), 1 if inverse else 0, ptr(out),
                                              ptr(lad), 1 if accumulate else 0, stream()), "rq_spline_fwd")
    return out, lad


def rq_spline_bwd(x, h, hscale, bound, lens, g_out, g_lad):
    _f32(x, h, g_out, g_lad)
    B, T = x.shape
    bins = (h.shape[-1] + 1) // 3
    dx, dh = torch.empty_like(x), torch.empty_like(h)
    _lib.check(_lib.lib().s2svc_rq_spline_bwd(B, T, bins, ptr(x), ptr(h), hscale, bound, ptr(lens), ptr(g_out), ptr(g_lad), ptr(dx),
                                              ptr(dh), stream()), "rq_spline_bwd")
    return dx, dh


def sdp_head_fwd(noise, lens, m, logs):
    _f32(noise, m, logs)
    B, _, T = noise.shape
    z0 = torch.empty((B, T), dtype=torch.float32, device=noise.device)
    z1 = torch.empty_like(z0)
    _lib.check(_lib.lib().s2svc_sdp_head_fwd(B, T, ptr(noise), ptr(lens), ptr(m), ptr(logs), ptr(z0), ptr(z1), stream()), "sdp_head_fwd")
    return z0, z1


def sdp_head_bwd(noise, lens, logs, dz0, dz1):
    B, _, T = noise.shape
    part = torch.empty((B, 4), dtype=torch.float32, device=noise.device)
    _lib.check(_lib.lib().s2svc_sdp_head_bwd(B, T, ptr(noise), ptr(lens), ptr(logs), ptr(dz0), ptr(dz1), ptr(part), stream()),
               "sdp_head_bwd")
    return part


def sdp_mid_fwd(zu, z1, w, lens, m, logs):
    _f32(zu, z1, w, m, logs)
    B, T = zu.shape
    y0, y1, lz = torch.empty_like(zu), torch.empty_like(zu), torch.empty_like(zu)
    _lib.check(_lib.lib().s2svc_sdp_mid_fwd(B, T, ptr(zu), ptr(z1), ptr(w), ptr(lens), ptr(m), ptr(logs), ptr(y0), ptr(y1), ptr(lz),
                                            stream()), "sdp_mid_fwd")
    return y0, y1, lz


def sdp_mid_bwd(zu, z1, w, lens, logs, dy0, dy1, dlz):
    B, T = zu.shape
    dzu, dz1 = torch.empty_like(zu), torch.empty_like(zu)
    part = torch.empty((B, 4), dtype=torch.float32, device=zu.device)
    _lib.check(_lib.lib().s2svc_sdp_mid_bwd(B, T, ptr(zu), ptr(z1), ptr(w), ptr(lens), ptr(logs), ptr(dy0), ptr(dy1), ptr(dlz),
                                            ptr(dzu), ptr(dz1), ptr(part), stream()), "sdp_mid_bwd")
    return dzu, dz1, part


def sdp_tail_fwd(noise, lens, zu, lz, lad_q, lad_p, af, bf, logs_q, logs_p, normalize=False):
    _f32(noise, zu, lz, lad_q, lad_p, af, bf, logs_q, logs_p)
    B, T = zu.shape
    out = torch.empty(B, dtype=torch.float32, device=zu.device)
    _lib.check(_lib.lib().s2svc_sdp_tail_fwd(B, T, ptr(noise), ptr(lens), ptr(zu), ptr(lz), ptr(lad_q), ptr(lad_p), ptr(af), ptr(bf),
                                             ptr(logs_q), ptr(logs_p), ptr(out), 1 if normalize else 0, stream()), "sdp_tail_fwd")
    return out


def sdp_tail_bwd(g, lens, zu, af, bf, normalize=False):
    _f32(g)
    B, T = zu.shape
    d_af, d_bf, d_lz, d_zu = (torch.empty_like(zu) for _ in range(4))
    neg_g = torch.empty(B, dtype=torch.float32, device=zu.device)
    part = torch.empty((B, 2), dtype=torch.float32, device=zu.device)
    _lib.check(_lib.lib().s2svc_sdp_tail_bwd(B, T, ptr(g), ptr(lens), ptr(zu), ptr(af), ptr(bf), ptr(d_af), ptr(d_bf), ptr(d_lz),
                                             ptr(d_zu), ptr(neg_g), ptr(part), 1 if normalize else 0, stream()), "sdp_tail_bwd")
    return d_af, d_bf, d_lz, d_zu, neg_g, part


def sdp_inverse_out(a, lens, m, logs):
    _f32(a, m, logs)
    B, T = a.shape
    dur = torch.empty_like(a)
    _lib.check(_lib.lib().s2svc_sdp_inverse_out(B, T, ptr(a), ptr(lens), ptr(m), ptr(logs), ptr(dur), stream()), "sdp_inverse_out")
    return dur
