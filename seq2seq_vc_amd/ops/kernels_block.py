"""Launchers for the fused attention sub-layer kernels (csrc/attn_block.hip) and the GEMM with a row prologue
(csrc/gemm_rowpro.hip); C-ABI in include/s2svc_hip.h."""
import ctypes
import os

import torch

from .. import _lib
from .kernels import _DT, ACT, ptr, stream

# The fused layer functions are OPT-IN (S2SVC_FUSED_BLOCKS=1): they cut the VTN step from 395 to 310 launches but measured
# SLOWER on MI355X (4.98 vs 4.48 ms, profiles/r03_fused_layers_*): a workgroup that owns an utterance's 64 rows runs one wave per
# SIMD on cold code and cold caches, so its LayerNorm row loop (10-15 us), epilogue and attention core run at single-wave
# latency, where the modular kernels spread the same rows over the whole chip.  Kept tested (kernel- and
# model-level cases toggle `_DISABLED`) as the base for a 16-wave version.
_DISABLED = os.environ.get("S2SVC_FUSED_BLOCKS", "0") != "1"


def block_supported(dtype, T1, T2, D, H):
    if _DISABLED or dtype != torch.bfloat16:
        return False
    return bool(_lib.lib().s2svc_attn_block_supported(_DT[dtype], T1, T2, D, H))


def rowpro_supported(dtype, D):
    return (not _DISABLED) and dtype == torch.bfloat16 and bool(_lib.lib().s2svc_gemm_rowpro_supported(_DT[dtype], D))


def _view_ok(t):
    return t.stride(2) == 1 and t.stride(1) % 8 == 0 and t.stride(0) % 8 == 0 and t.data_ptr() % 16 == 0


def attn_block_fwd(x, norm, w, bias, H, klen, causal, p, seed, kv=None):
    """x (B,T1,D) dense bf16; norm = (gamma, beta, eps) or None; w = packed [Wq;Wk;Wv] (3D, D) (kv None) or Wq (D, D) with
    kv = (k, v) views (B,T2,D) of the memory's projection.
    -> (ctx (B,T1,D), attn (B,H,T1,ld), proj (B,T1,3D | D), y, mean, rstd)   (y / mean / rstd None without a norm)."""
    B, T1, D = x.shape
    nproj = 3 if kv is None else 1
    T2 = T1 if kv is None else kv[0].shape[1]
    ld = (T2 + 7) // 8 * 8
    dev = x.device
    attn = torch.empty((B, H, T1, ld), dtype=x.dtype, device=dev)
    out = torch.empty((B, T1, D), dtype=x.dtype, device=dev)
    proj = torch.empty((B, T1, nproj * D), dtype=x.dtype, device=dev)
    y = mean = rstd = None
    g = b = None
    eps = 0.0
    if norm is not None:
        g, b, eps = norm
        y = torch.empty_like(x)
        mean = torch.empty(B * T1, dtype=torch.float32, device=dev)
        rstd = torch.empty(B * T1, dtype=torch.float32, device=dev)
    k = v = None
    ldk = kbs = ldv = vbs = 0
    if kv is not None:
        k, v = kv
        if not (_view_ok(k) and _view_ok(v)):
            raise ValueError("attn_block_fwd: memory K / V must be 16-byte aligned views with a contiguous last dim")
        ldk, kbs, ldv, vbs = k.stride(1), k.stride(0), v.stride(1), v.stride(0)
    _lib.check(_lib.lib().s2svc_attn_block_fwd(B, H, T1, T2, D, nproj, ptr(x), ptr(g), ptr(b), eps, ptr(y), ptr(mean), ptr(rstd), ptr(w),
                                               ptr(bias), ptr(proj), ptr(k), ldk, kbs, ptr(v), ldv, vbs, ptr(klen), 1 if causal else 0,
                                               1.0 / (D // H) ** 0.5, p, seed[0], seed[1], ptr(attn), ld, ptr(out), stream()),
               "attn_block_fwd")
    return out, attn, proj, y, mean, rstd


def attn_block_bwd(g, ln, ds_extra, p_res, hscale, seed_res, wo_t, q, k, v, attn, dattn, H, p, seed, dq, dk, dv, want_da=True):
    """g (B,T1,D): gradient of the residual stream behind the output projection (mode 0), or of LayerNorm(s) with
    ln = (s, mean, rstd, gamma) (mode 1).  Writes dq / dk / dv (views); -> (ds | None, da | None)."""
    B, T1, D = g.shape
    T2 = k.shape[1]
    ld = attn.shape[-1]
    mode = 0 if ln is None else 1
    s = mean = rstd = gamma = ds = None
    if ln is not None:
        s, mean, rstd, gamma = ln
        ds = torch.empty_like(g)
    da = None
    if want_da and (mode == 1 or p_res > 0.0 or hscale != 1.0):
        da = torch.empty_like(g)
    for t in (q, k, v, dq, dk, dv):
        if not _view_ok(t):
            raise ValueError("attn_block_bwd: q / k / v / dq / dk / dv must be 16-byte aligned views with a contiguous last dim")
    _lib.check(_lib.lib().s2svc_attn_block_bwd(B, H, T1, T2, D, mode, ptr(g), ptr(s), ptr(mean), ptr(rstd), ptr(gamma), ptr(ds_extra),
                                               ptr(ds), ptr(da), p_res, hscale, seed_res[0], seed_res[1], ptr(wo_t),
                                               ptr(q), q.stride(1), q.stride(0), ptr(k), k.stride(1), k.stride(0), ptr(v), v.stride(1),
                                               v.stride(0), ptr(attn), ptr(dattn), ld, 1.0 / (D // H) ** 0.5, p, seed[0], seed[1],
                                               ptr(dq), dq.stride(1), dq.stride(0), ptr(dk), dk.stride(1), dk.stride(0),
                                               ptr(dv), dv.stride(1), dv.stride(0), stream()), "attn_block_bwd")
    if da is None and want_da:
        da = g                 # mode 0 without dropout: dA is the incoming gradient itself
    return ds, da


def gemm_rowpro(x2, w, N, out, *, mode=0, norm=None, p_a=0.0, hscale=1.0, seed_a=(None, 0), write_rows=False, bias=None, act=None,
                res=None, emask=None, emask_mode=0, drop_p=0.0, seed=(None, 0)):
    """out[M, N] = epilogue(prologue(x2)[M, D] . w[N, D]^T)  (s2svc_gemm_rowpro).
    mode 1: norm = (gamma, beta, eps) -> returns (y, mean, rstd); mode 2: rows * dropmask(p_a, seed_a) * hscale, returned when
    write_rows; mode 0: plain rows."""
    M, D = x2.shape
    d = _lib.GemmDesc()
    d.C, d.ldc, d.c_dtype = out.data_ptr(), N, _DT[out.dtype]
    d.bias, d.res, d.ldr = ptr(bias), ptr(res), N
    d.M, d.N, d.K, d.nb0, d.nb1 = M, N, D, 1, 1
    d.act, d.alpha, d.dtype, d.splitk = ACT[act], 1.0, _DT[x2.dtype], 1
    if emask is not None:
        d.emask, d.ldm, d.emask_mode = emask.data_ptr(), N, int(emask_mode)
    d.drop_p, d.seed_base, d.seed_off = drop_p, seed[0], seed[1]
    y = mean = rstd = None
    g = b = None
    eps = 0.0
    if mode == 1:
        g, b, eps = norm
        y = torch.empty_like(x2)
        mean = torch.empty(M, dtype=torch.float32, device=x2.device)
        rstd = torch.empty(M, dtype=torch.float32, device=x2.device)
    elif mode == 2 and write_rows and (p_a > 0.0 or hscale != 1.0):
        y = torch.empty_like(x2)
    _lib.check(_lib.lib().s2svc_gemm_rowpro(mode, D, ptr(x2), ptr(g), ptr(b), eps, ptr(y), ptr(mean), ptr(rstd), p_a, hscale, seed_a[0],
                                            seed_a[1], ptr(w), ctypes.byref(d), stream()), "s2svc_gemm_rowpro")
    if mode == 2 and write_rows and y is None:
        y = x2
    return y, mean, rstd
