"""Launchers for the fused short-sequence attention kernels (csrc/attn_fused.hip; C-ABI in include/s2svc_hip.h)."""
import os

import torch

from .. import _lib
from .kernels import _DT, ptr, stream

_DISABLED = False         # tests flip it: fused vs separate attention kernels


def _strided_ok(t):
    """(B, T, D') view whose last dim is contiguous and whose row / batch strides and base keep 16-byte pieces aligned."""
    return t.stride(2) == 1 and t.stride(1) % 8 == 0 and t.stride(0) % 8 == 0 and t.data_ptr() % 16 == 0


view_ok = _strided_ok


def supported(q, k, v, H):
    if _DISABLED or q.dtype != torch.bfloat16:
        return False
    dk = q.shape[-1] // H
    if not _lib.lib().s2svc_attn_fused_supported(_DT[q.dtype], q.shape[1], k.shape[1], dk):
        return False
    return _strided_ok(q) and _strided_ok(k) and _strided_ok(v)


def fused_fwd(q, k, v, klen, causal, H, scale, p, seed):
    """q (B,T1,D), k/v (B,T2,D) (possibly column slices of packed projections) -> (out (B,T1,D), attn (B,H,T1,ld))."""
    B, T1, D = q.shape
    T2 = k.shape[1]
    dk = D // H
    ld = (T2 + 7) // 8 * 8
    attn = torch.empty((B, H, T1, ld), dtype=q.dtype, device=q.device)
    out = torch.empty((B, T1, D), dtype=q.dtype, device=q.device)
    _lib.check(_lib.lib().s2svc_attn_fused_fwd(B, H, T1, T2, dk, ptr(q), q.stride(1), q.stride(0), ptr(k), k.stride(1), k.stride(0),
                                               ptr(v), v.stride(1), v.stride(0), ptr(klen), 1 if causal else 0, scale, p, seed[0],
                                               seed[1], ptr(attn), ld, ptr(out), D, T1 * D, stream()), "attn_fused_fwd")
    return out, attn


def fused_bwd(q, k, v, dout, attn, dattn, H, scale, p, seed, dq, dk_out, dv):
    """Writes dq / dk / dv (views with last dim contiguous, e.g. slices of a packed gradient)."""
    B, T1, D = q.shape
    T2 = k.shape[1]
    dk = D // H
    ld = attn.shape[-1]
    for t in (dout, dq, dk_out, dv):
        if t.stride(2) != 1:
            raise ValueError("fused attention backward: last dim of gradients must be contiguous")
    _lib.check(_lib.lib().s2svc_attn_fused_bwd(B, H, T1, T2, dk, ptr(q), q.stride(1), q.stride(0), ptr(k), k.stride(1), k.stride(0),
                                               ptr(v), v.stride(1), v.stride(0), ptr(dout), dout.stride(1), dout.stride(0), ptr(attn),
                                               ptr(dattn), ld, scale, p, seed[0], seed[1], ptr(dq), dq.stride(1), dq.stride(0),
                                               ptr(dk_out), dk_out.stride(1), dk_out.stride(0), ptr(dv), dv.stride(1), dv.stride(0),
                                               stream()), "attn_fused_bwd")


# ----------------------------------------------------------------------------------------------
# attention map of plain attention in one launch, T2 <= 512 (csrc/attn_map.hip)
# ----------------------------------------------------------------------------------------------
_MAP_DISABLED = os.environ.get("S2SVC_NO_ATTNMAP", "0") == "1"       # (the switch: A/B against scores GEMM + softmax kernel)


def map_supported(q, k, H):
    if _MAP_DISABLED or q.dtype != torch.bfloat16:
        return False
    dk = q.shape[-1] // H
    if not _lib.lib().s2svc_attn_map_supported(_DT[q.dtype], q.shape[1], k.shape[1], dk):
        return False
    return _strided_ok(q) and _strided_ok(k)


def map_product_ok(m, H):
    """The map launches can also carry their tile's product with this (B, T2, D) view (v forward, k backward)."""
    return bool(_lib.lib().s2svc_attn_map_product_supported(m.shape[-1] // H)) and _strided_ok(m)


def map_fwd(q, k, klen, causal, H, scale, p, seed, v=None):
    """q (B,T1,D), k (B,T2,D) (possibly column slices of packed projections) -> attn, pdrop (B,H,T1,ld) (pdrop None when p == 0), and
    with v (map_product_ok) the context (B,T1,D) = (dropped map) . v of the same launch (else None)."""
    B, T1, D = q.shape
    T2 = k.shape[1]
    dk = D // H
    ld = (T2 + 7) // 8 * 8
    attn = torch.empty((B, H, T1, ld), dtype=q.dtype, device=q.device)
    pdrop = torch.empty_like(attn) if p > 0.0 else None
    ctx = torch.empty((B, T1, D), dtype=q.dtype, device=q.device) if v is not None else None
    _lib.check(_lib.lib().s2svc_attn_map_fwd(B, H, T1, T2, dk, ptr(q), q.stride(1), q.stride(0), ptr(k), k.stride(1), k.stride(0),
                                             ptr(klen), 1 if causal else 0, scale, p, seed[0], seed[1], ptr(attn), ptr(pdrop), ld,
                                             ptr(v), v.stride(1) if v is not None else 0, v.stride(0) if v is not None else 0,
                                             ptr(ctx), D, T1 * D, stream()), "attn_map_fwd")
    return attn, pdrop, ctx


def map_bwd(dctx, v, attn, dattn, H, scale, p, seed, ldb=0, k=None, dq=None):
    """dctx (B,T1,D), v (B,T2,D) views; attn (and dattn or None) (B,H,T1,ld) -> gradient of the scaled scores (B,H,T1,ld) and, with
    ldb > 0 (relative-position self-attention, "new" rel_shift), the gradient of the unshifted position term (B,H,T1,ldb).
    With k (map_product_ok) and dq (a (B,T1,D) view, last dim contiguous): dq = dS . k in the same launch."""
    B, T1, D = dctx.shape
    T2 = v.shape[1]
    ld = attn.shape[-1]
    if dattn is not None and (dattn.shape != attn.shape or not dattn.is_contiguous()):
        raise ValueError("attention map backward: dattn must have the stored map's padded layout")
    if (k is None) != (dq is None) or (dq is not None and (dq.stride(2) != 1 or dq.shape != dctx.shape)):
        raise ValueError("attention map backward: k and dq come together, dq (B,T1,D) with its last dim contiguous")
    ds = torch.empty_like(attn)
    dbd = torch.empty((B, H, T1, ldb), dtype=attn.dtype, device=attn.device) if ldb else None
    _lib.check(_lib.lib().s2svc_attn_map_bwd(B, H, T1, T2, D // H, ptr(dctx), dctx.stride(1), dctx.stride(0), ptr(v), v.stride(1), v.stride(0),
                                             ptr(attn), ptr(dattn), scale, p, seed[0], seed[1], ptr(ds), ld, ptr(dbd), ldb,
                                             ptr(k), k.stride(1) if k is not None else 0, k.stride(0) if k is not None else 0,
                                             ptr(dq), dq.stride(1) if dq is not None else 0, dq.stride(0) if dq is not None else 0,
                                             stream()), "attn_map_bwd")
    return ds, dbd


# ----------------------------------------------------------------------------------------------
# relative-position self-attention, T <= 256 (csrc/relattn.hip)
# ----------------------------------------------------------------------------------------------


def rel_supported(q, k, v, pos, H, rel_mode):
    if os.environ.get("S2SVC_NO_RELATTN", "0") == "1" or q.dtype != torch.bfloat16:        # (the switch: tests compare with the separate kernels)
        return False
    T, dk = q.shape[1], q.shape[-1] // H
    if not _lib.lib().s2svc_relattn_supported(_DT[q.dtype], T, dk, rel_mode) or pos.shape[1] != 2 * T - 1:
        return False
    return _strided_ok(q) and _strided_ok(k) and _strided_ok(v) and pos.is_contiguous() and pos.data_ptr() % 16 == 0


def rel_fwd(q, k, pos, u, v, klen, H, scale, p, seed):
    """q, k: (B,T,D) column blocks of the packed projection; pos (1,2T-1,D); u, v (H*dk) fp32 -> attn, pdrop (B,H,T,ld), qu, qv."""
    B, T, D = q.shape
    dk = D // H
    ld = (T + 7) // 8 * 8
    attn = torch.empty((B, H, T, ld), dtype=q.dtype, device=q.device)
    pdrop = torch.empty_like(attn) if p > 0.0 else None
    qu = torch.empty((B, T, D), dtype=q.dtype, device=q.device)
    qv = torch.empty((B, T, D), dtype=q.dtype, device=q.device)
    _lib.check(_lib.lib().s2svc_relattn_fwd(B, H, T, dk, ptr(q), q.stride(1), q.stride(0), ptr(k), k.stride(1), k.stride(0), ptr(pos),
                                            pos.stride(1), pos.shape[1], ptr(u), ptr(v), ptr(klen), scale, p, seed[0], seed[1],
                                            ptr(attn), ptr(pdrop), ld, ptr(qu), ptr(qv), stream()), "relattn_fwd")
    return attn, pdrop, qu, qv
