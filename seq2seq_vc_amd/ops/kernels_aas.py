"""Launchers for the AAS-VC / Conformer specific kernels (csrc/dwconv.hip, align.hip, ctc.hip)."""
import math

import torch

from .. import _lib
from .kernels import _DT, dt, ptr, stream

_WS_CHUNKS = 64


def dwconv(x, w, bias, ks, dil=1, flip=False, add=None):
    """x (B,T,C) channel-last; w fp32 (C,1,k) contiguous; flip=True gives the data gradient of dy; add: a tensor of the output's
    shape added to the result (needs the vectorised kernel: dwconv_add_ok)."""
    B, T, C = x.shape
    y = torch.empty_like(x)
    _lib.check(_lib.lib().s2svc_dwconv_add(dt(x), B, T, C, ks, dil, ptr(x), ptr(w), ptr(bias), ptr(add), ptr(y), 1 if flip else 0,
                                           stream()), "dwconv")
    return y


def dwconv_add_ok(x, ks):
    return x.shape[-1] % (4 if x.dtype == torch.float32 else 8) == 0 and x.data_ptr() % 16 == 0 and ks <= 63


def dwconv_wgrad(x, dy, ks, dil=1, out=None):
    """dw (C, 1, ks) fp32; `out`: a gradient slot of that shape to ACCUMULATE into instead (no separate add launch)."""
    B, T, C = x.shape
    dw = torch.empty((C, 1, ks), dtype=torch.float32, device=x.device) if out is None else out
    ws = torch.empty(_WS_CHUNKS * C * ks, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().s2svc_dwconv_wgrad(dt(x), B, T, C, ks, dil, ptr(x), ptr(dy), ptr(dw), 0 if out is None else 1, ptr(ws),
                                             _WS_CHUNKS, stream()), "dwconv_wgrad")
    return dw


def convmod_supported(C, ks):
    return bool(_lib.lib().s2svc_convmod_supported(C, ks))


def convmod_fwd(y2, w, bias, ks, eps, momentum, run_mean=None, run_var=None, num_batches=None, vlens=None):
    """Conformer convolution module core, bf16 training: y2 (B,T,2C) -> z = dwconv(glu(y2)) (B,T,C) and its batch statistics
    (mean, rstd) in two launches (csrc/convmod.hip).  vlens (B int32, device): frames t >= vlens[b] are absent (include/s2svc_hip.h)."""
    B, T, C2 = y2.shape
    C = C2 // 2
    z = torch.empty((B, T, C), dtype=y2.dtype, device=y2.device)
    mean = torch.empty(C, dtype=torch.float32, device=y2.device)
    rstd = torch.empty(C, dtype=torch.float32, device=y2.device)
    ws = torch.empty(B * ((T + 63) // 64) * 2 * C, dtype=torch.float32, device=y2.device)
    _lib.check(_lib.lib().s2svc_convmod_fwd(B, T, C, ks, ptr(y2), ptr(w), ptr(bias), ptr(z), eps, momentum, ptr(mean), ptr(rstd),
                                            ptr(run_mean), ptr(run_var), ptr(num_batches), ptr(ws), ptr(vlens), stream()), "convmod_fwd")
    return z, mean, rstd


def bn_swish_apply(z, mean, rstd, gamma, beta, vlens=None):
    C = z.shape[-1]
    out = torch.empty_like(z)
    _lib.check(_lib.lib().s2svc_bn_swish_apply(z.numel() // C, C, ptr(z), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(out),
                                               z.shape[-2], ptr(vlens), stream()), "bn_swish_apply")
    return out


def convmod_bwd(da, z, y2, w, mean, rstd, gamma, beta, ks, dgamma_acc=None, dbeta_acc=None, vlens=None):
    """-> dy2 (B,T,2C), sdy, sdyx (C), (ws_w, chunks): the per-tile partial depthwise weight / bias gradients for
    convmod_wgrad_final.  dgamma_acc / dbeta_acc: fp32 (C) gradient slots to ADD the BatchNorm parameter gradients to."""
    B, T, C = z.shape
    dy2 = torch.empty_like(y2)
    sdy = torch.empty(C, dtype=torch.float32, device=z.device)
    sdyx = torch.empty(C, dtype=torch.float32, device=z.device)
    chunks = B * ((T + 63) // 64)
    ws_stats = torch.empty(((B * T + 63) // 64) * 2 * C, dtype=torch.float32, device=z.device)
    ws_w = torch.empty(chunks * C * (ks + 1), dtype=torch.float32, device=z.device)
    _lib.check(_lib.lib().s2svc_convmod_bwd(B, T, C, ks, ptr(da), ptr(z), ptr(y2), ptr(w), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta),
                                            ptr(dy2), ptr(sdy), ptr(sdyx), ptr(dgamma_acc), ptr(dbeta_acc), ptr(ws_stats), ptr(ws_w),
                                            ptr(vlens), stream()), "convmod_bwd")
    return dy2, sdy, sdyx, (ws_w, chunks)


def convmod_wgrad_final(ws_w, chunks, C, ks, dw=None, db=None, accumulate=False):
    """Sum the per-tile partials into dw (C,1,ks) / db (C) fp32 (allocated unless given; accumulate=True adds to them)."""
    if dw is None:
        dw = torch.empty((C, 1, ks), dtype=torch.float32, device=ws_w.device)
    if db is None:
        db = torch.empty(C, dtype=torch.float32, device=ws_w.device)
    _lib.check(_lib.lib().s2svc_convmod_wgrad_final(C, ks, chunks, ptr(ws_w), ptr(dw), ptr(db), 1 if accumulate else 0, stream()),
               "convmod_wgrad_final")
    return dw, db


def pairwise_l2_logsoftmax(feats, text, text_lens_i32):
    B, Tf, A = feats.shape
    Tx = text.shape[1]
    logp = torch.empty((B, Tf, Tx), dtype=torch.float32, device=feats.device)
    dist = torch.empty((B, Tf, Tx), dtype=torch.float32, device=feats.device)
    _lib.check(_lib.lib().s2svc_pairwise_l2_logsoftmax(dt(feats), B, Tf, Tx, A, ptr(feats), ptr(text), ptr(text_lens_i32),
                                                       ptr(logp), ptr(dist), stream()), "pairwise_l2_logsoftmax")
    return logp, dist


def pairwise_l2_bwd_g(logp, dist, dlogp, text_lens_i32, out_dtype):
    B, Tf, Tx = logp.shape
    G = torch.empty((B, Tf, Tx), dtype=out_dtype, device=logp.device)
    rowsum = torch.empty((B, Tf), dtype=torch.float32, device=logp.device)
    _lib.check(_lib.lib().s2svc_pairwise_l2_bwd_g(_DT[out_dtype], B, Tf, Tx, ptr(logp), ptr(dist), ptr(dlogp), ptr(text_lens_i32),
                                                  ptr(G), ptr(rowsum), stream()), "pairwise_l2_bwd_g")
    return G, rowsum


def rowscale(x, s):
    D = x.shape[-1]
    out = torch.empty_like(x)
    _lib.check(_lib.lib().s2svc_rowscale(dt(x), x.numel() // D, D, ptr(x), ptr(s), ptr(out), stream()), "rowscale")
    return out


def gauss_upsample_probs(ds, text_lens_i32, feat_lens_i32, Tf, out_dtype, delta=0.1):
    B, Tx = ds.shape
    P = torch.empty((B, Tf, Tx), dtype=out_dtype, device=ds.device)
    _lib.check(_lib.lib().s2svc_gauss_upsample_probs(_DT[out_dtype], B, Tf, Tx, ptr(ds), ptr(text_lens_i32), ptr(feat_lens_i32),
                                                     delta, ptr(P), stream()), "gauss_upsample_probs")
    return P


def betabinom_prior(B, Tf, Tx, text_lens_i32, feat_lens_i32, device):
    prior = torch.empty((B, Tf, Tx), dtype=torch.float32, device=device)
    _lib.check(_lib.lib().s2svc_betabinom_prior(B, Tf, Tx, ptr(text_lens_i32), ptr(feat_lens_i32), ptr(prior), stream()),
               "betabinom_prior")
    return prior


def forward_sum(log_p_attn, prior, text_lens_i32, feat_lens_i32, blank_prob=math.e ** -1):
    B, Tf, Tx = log_p_attn.shape
    dev = log_p_attn.device
    ws = torch.empty(_lib.lib().s2svc_forward_sum_ws_bytes(B, Tf, Tx) // 4 + 1, dtype=torch.float32, device=dev)
    loss_b = torch.empty((B,), dtype=torch.float32, device=dev)
    grad = torch.empty((B, Tf, Tx), dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().s2svc_forward_sum(B, Tf, Tx, ptr(log_p_attn), ptr(prior), ptr(text_lens_i32), ptr(feat_lens_i32),
                                            math.log(blank_prob), ptr(ws), ptr(loss_b), ptr(grad), stream()), "forward_sum")
    return loss_b, grad


def length_regulate_index(ds_i32, Tout):
    """ds (B,Tx) int32 -> start (B,Tx), idx (B,Tout), total (B) int32."""
    B, Tx = ds_i32.shape
    dev = ds_i32.device
    start = torch.empty((B, Tx), dtype=torch.int32, device=dev)
    idx = torch.empty((B, max(Tout, 1)), dtype=torch.int32, device=dev)
    total = torch.empty((B,), dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().s2svc_length_regulate_index(B, Tx, Tout, ptr(ds_i32), ptr(start), ptr(idx), ptr(total), stream()),
               "length_regulate_index")
    return start, idx[:, :Tout], total


def length_regulate_fwd(x, idx, Tout, pad_value=0.0):
    B, Tx, D = x.shape
    y = torch.empty((B, Tout, D), dtype=x.dtype, device=x.device)
    _lib.check(_lib.lib().s2svc_length_regulate_fwd(dt(x), B, Tx, Tout, D, ptr(x), ptr(idx), float(pad_value), ptr(y), stream()),
               "length_regulate_fwd")
    return y


def length_regulate_bwd(dy, start, ds_i32, Tx):
    B, Tout, D = dy.shape
    dx = torch.empty((B, Tx, D), dtype=dy.dtype, device=dy.device)
    _lib.check(_lib.lib().s2svc_length_regulate_bwd(dt(dy), B, Tx, Tout, D, ptr(dy), ptr(start), ptr(ds_i32), ptr(dx), stream()),
               "length_regulate_bwd")
    return dx


def attn_durations(att):
    """att (NH, Tf, Tx) fp32 -> (durations int64 (Tx,), focus rate scalar, head index)."""
    NH, Tf, Tx = att.shape
    dur = torch.empty((Tx,), dtype=torch.int64, device=att.device)
    focus = torch.empty((), dtype=torch.float32, device=att.device)
    head = torch.empty((), dtype=torch.int32, device=att.device)
    _lib.check(_lib.lib().s2svc_attn_durations(NH, Tf, Tx, ptr(att), ptr(dur), ptr(focus), ptr(head), stream()), "attn_durations")
    return dur, focus, head
