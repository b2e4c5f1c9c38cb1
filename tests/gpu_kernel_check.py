"""Kernel-level parity sweep (runs on the GPU box): every launcher in seq2seq_vc_amd.ops.kernels is
compared with a plain fp32 torch formulation of the same op.  Used two ways:
  * `python tests/gpu_kernel_check.py` prints a PASS/FAIL table for all cases and never stops early
    (one gpurun round trip shows every broken kernel);
  * tests/test_gpu_kernels.py imports CASES and turns each into a `@pytest.mark.gpu` test.
"""
import math
import os
import sys
import traceback

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from seq2seq_vc_amd.ops import kernels as K  # noqa: E402

DEV = "cuda"
CASES = []


def case(fn):
    CASES.append(fn)
    return fn


def g(seed):
    return torch.Generator(device="cpu").manual_seed(seed)


def rnd(*shape, seed=0, dtype=torch.float32, scale=1.0):
    return (torch.randn(*shape, generator=g(seed)) * scale).to(DEV).to(dtype)


def tol(dtype):
    return (2e-5, 2e-5) if dtype == torch.float32 else (3e-2, 3e-2)


def check(name, got, ref, dtype, rtol=None, atol=None):
    r, a = tol(dtype)
    rtol = r if rtol is None else rtol
    atol = a if atol is None else atol
    got = got.float().cpu()
    ref = ref.float().cpu()
    if got.shape != ref.shape:
        return False, f"{name}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    err = (got - ref).abs()
    bound = atol + rtol * ref.abs()
    bad = (err > bound) | torch.isnan(got)
    if bad.any():
        idx = torch.nonzero(bad)[0].tolist()
        return False, (f"{name}: {int(bad.sum())}/{bad.numel()} bad, max_err={err.max():.3e} first_bad={idx} "
                       f"got={got[tuple(idx)]:.6f} ref={ref[tuple(idx)]:.6f}")
    return True, f"{name}: ok max_err={err.max():.3e}"


def both_dtypes(fn):
    def run():
        out = []
        for dtype in (torch.float32, torch.bfloat16):
            out += fn(dtype)
        return out
    run.__name__ = fn.__name__
    return run


# ------------------------------------------------------------------------------------------------
@case
@both_dtypes
def gemm_linear(dtype):
    res = []
    for (M, N, Kd, seed) in [(100, 70, 80, 1), (2016, 384, 384, 2), (64, 4, 384, 3), (130, 320, 256, 4),
                             (256, 1536, 384, 5), (37, 33, 7296, 6), (512, 512, 36, 7)]:
        x, w, b = rnd(M, Kd, seed=seed, dtype=dtype), rnd(N, Kd, seed=seed + 10, dtype=dtype, scale=0.05), rnd(N, seed=seed + 20)
        r = rnd(M, N, seed=seed + 30, dtype=dtype)
        out = torch.empty(M, N, dtype=dtype, device=DEV)
        K.gemm(K.operand(x, Kd), K.operand(w, Kd), M, N, Kd, out, in_dtype=dtype, bias=b, act="relu", res=r)
        ref = torch.relu(x.float() @ w.float().t() + b) + r.float()
        res.append(check(f"linear[{dtype}] {M}x{N}x{Kd}", out, ref, dtype))
    return res


@case
@both_dtypes
def gemm_epilogue_dropout_mask(dtype):
    """Epilogue stage `* dropmask * (emask > 0)`: the mask must be the one the standalone dropout kernel draws for the same
    (seed, element index), on every kernel family (LDS-DMA vector epilogue, scalar epilogue, skinny, split-K reduce)."""
    res = []
    K.manual_seed(77)
    for (M, N, Kd, sk, seed) in [(256, 192, 128, 1, 1), (2016, 1536, 384, 1, 2), (40, 96, 64, 1, 3), (130, 70, 80, 1, 4), (256, 192, 512, 2, 5)]:
        x, w, b = rnd(M, Kd, seed=seed, dtype=dtype), rnd(N, Kd, seed=seed + 10, dtype=dtype, scale=0.1), rnd(N, seed=seed + 20)
        em = rnd(M, N, seed=seed + 30, dtype=dtype)
        sd = K.new_seed(x.device)
        p = 0.3
        scale = K.act_dropout_fwd(torch.ones(M, N, dtype=torch.float32, device=DEV), None, p, sd)     # 0 or 1/(1-p)
        out = torch.empty(M, N, dtype=dtype, device=DEV)
        K.gemm(K.operand(x, Kd), K.operand(w, Kd), M, N, Kd, out, in_dtype=dtype, bias=b, act="relu", emask=em, drop_p=p, seed=sd,
               splitk=sk)
        ref = torch.relu(x.float() @ w.float().t() + b) * scale * (em.float() > 0)
        res.append(check(f"gemm epilogue dropout+mask[{dtype}] {M}x{N}x{Kd} splitk={sk}", out, ref, dtype))
        keep = (scale > 0).float().mean().item()
        res.append((abs(keep - (1 - p)) < 0.02, f"keep rate {keep:.3f} (p = {p})"))
    return res


@case
def attention_fused_vs_reference():
    """Fused short-sequence attention (bf16): forward / backward vs fp32 torch math, and -- with dropout on -- vs the unfused
    kernels (same seed => same dropout masks), for self / causal / source attention on packed and separate projections."""
    from seq2seq_vc_amd.ops import functional as Fn
    from seq2seq_vc_amd.ops import kernels_attn as KAT
    res = []
    dt_ = torch.bfloat16
    for (B, H, T1, T2, dk, causal, seed) in [(3, 4, 63, 63, 96, False, 1), (2, 4, 64, 64, 96, True, 2), (3, 2, 64, 63, 64, False, 3),
                                             (2, 2, 17, 40, 32, False, 4), (1, 1, 5, 9, 128, False, 5)]:
        D = H * dk
        q, k, v = (rnd(B, T1 if i == 0 else T2, D, seed=seed * 10 + i, dtype=dt_) for i in range(3))
        klen = torch.tensor([T2, max(1, T2 - 7), max(1, T2 // 2)][:B], dtype=torch.int32, device=DEV)
        dy = rnd(B, T1, D, seed=seed * 10 + 5, dtype=dt_)
        datt = rnd(B, H, T1, T2, seed=seed * 10 + 6, dtype=dt_) * 0.1
        scale = 1 / math.sqrt(dk)
        # fp32 reference
        qr, kr, vr = (t.float().clone().requires_grad_(True) for t in (q, k, v))
        qh = qr.view(B, T1, H, dk).transpose(1, 2)
        kh = kr.view(B, T2, H, dk).transpose(1, 2)
        vh = vr.view(B, T2, H, dk).transpose(1, 2)
        mask = torch.arange(T2, device=DEV)[None, None, None, :] < klen[:, None, None, None]
        if causal:
            mask = mask & torch.tril(torch.ones(T1, T2, dtype=torch.bool, device=DEV))[None, None]
        sc = (qh @ kh.transpose(-1, -2) * scale).masked_fill(~mask, torch.finfo(torch.float32).min)
        pr = torch.softmax(sc, -1).masked_fill(~mask, 0.0)
        outr = (pr @ vh).transpose(1, 2).reshape(B, T1, D)
        (outr * dy.float()).sum().backward(retain_graph=True)
        g_plain = [t.grad.clone() for t in (qr, kr, vr)]
        for t in (qr, kr, vr):
            t.grad = None
        ((outr * dy.float()).sum() + (pr * datt.float()).sum()).backward()
        g_att = [t.grad.clone() for t in (qr, kr, vr)]
        assert KAT.supported(q, k, v, H), "shape should take the fused kernel"
        for use_datt, gref in ((False, g_plain), (True, g_att)):
            qf, kf, vf = (t.clone().requires_grad_(True) for t in (q, k, v))
            out, att = Fn.attention_core(qf, kf, vf, klen, causal, H, 0.0)
            loss = (out.float() * dy.float()).sum()
            if use_datt:
                loss = loss + (att.float() * datt.float()).sum()
            loss.backward()
            tag = f"fused attn B{B} H{H} T{T1}x{T2} dk{dk} causal={causal} datt={use_datt}"
            if not use_datt:
                res.append(check(tag + " out", out, outr, dt_, atol=3e-2))
                res.append(check(tag + " attn", att, pr, dt_, atol=1e-2))
                res.append(check(tag + " attn rows sum to 1", att.float().sum(-1), torch.ones(B, H, T1), torch.float32, atol=2e-2))
            for nm, a, b_ in zip(("dq", "dk", "dv"), (qf.grad, kf.grad, vf.grad), gref):
                res.append(check(tag + " " + nm, a, b_, dt_, atol=0.15, rtol=5e-2))
        # dropout on: fused == unfused (identical masks), packed QKV path
        qkv = torch.cat([q[:, :min(T1, T2)], k[:, :min(T1, T2)], v[:, :min(T1, T2)]], dim=-1).contiguous() if T1 != T2 else torch.cat([q, k, v], -1)
        Tq = qkv.shape[1]
        kl2 = torch.clamp(klen, max=Tq)
        outs = []
        for disabled in (False, True):
            KAT._DISABLED = disabled
            try:
                K.manual_seed(123)
                K.reset_op_counter()
                x = qkv.clone().requires_grad_(True)
                o, a = Fn.attention_packed_qkv(x, kl2, causal, H, 0.3)
                (o.float() * dy[:, :Tq].float()).sum().backward()
                outs.append((o.detach(), a.detach(), x.grad.detach()))
            finally:
                KAT._DISABLED = False
        for nm, a, b_ in zip(("out", "attn", "dqkv"), outs[0], outs[1]):
            res.append(check(f"fused vs unfused (dropout 0.3) T{Tq} dk{dk} {nm}", a, b_, dt_, atol=0.12 if nm == "dqkv" else 4e-2, rtol=5e-2))
    return res


@case
def split_cols_block_consumed_twice():
    """One column block of split_cols feeding TWO source-attention calls (no model does; Fn._GradSink hands every block's in-place
    view to its first consumer only): the block's gradient must be g1 + g2, as with plain slicing."""
    from seq2seq_vc_amd.ops import functional as Fn
    res = []
    B, H, T1, T2, dk = 2, 4, 40, 48, 96
    D = H * dk
    dt_ = torch.bfloat16
    q1, q2 = rnd(B, T1, D, seed=1, dtype=dt_), rnd(B, T1, D, seed=2, dtype=dt_)
    kv_all = rnd(B, T2, 2 * 2 * D, seed=3, dtype=dt_)
    klen = torch.tensor([T2, T2 - 5], dtype=torch.int32, device=DEV)
    dy1, dy2, dy3 = rnd(B, T1, D, seed=4, dtype=dt_), rnd(B, T1, D, seed=5, dtype=dt_), rnd(B, T1, D, seed=6, dtype=dt_)
    grads = []
    for use_split in (True, False):
        x = kv_all.clone().requires_grad_(True)
        blocks = Fn.split_cols(x, 2) if use_split else (x[..., :2 * D], x[..., 2 * D:])
        o1, _ = Fn.attention_packed_kv(q1, blocks[0], klen, False, H, 0.0)
        o2, _ = Fn.attention_packed_kv(q2, blocks[0], klen, False, H, 0.0)        # block 0 a second time
        o3, _ = Fn.attention_packed_kv(q1, blocks[1], klen, False, H, 0.0)
        ((o1.float() * dy1.float()).sum() + (o2.float() * dy2.float()).sum() + (o3.float() * dy3.float()).sum()).backward()
        grads.append(x.grad.detach().clone())
    res.append(check("split_cols block used twice: gradient == plain slicing", grads[0], grads[1], dt_, atol=2e-2, rtol=2e-2))
    g0 = grads[1][..., :2 * D].float().abs().mean().item()
    res.append((g0 > 0, f"split_cols twice: block 0 gradient is non-trivial ({g0:.3e})"))
    return res


@case
def ffn_relu_fused_vs_unfused():
    """_FFNRelu (masks in GEMM epilogues) against the three-op composition, fp32, dropout off (exact) and on (same statistics)."""
    from seq2seq_vc_amd.ops import functional as Fn
    res = []
    M, D, H = 300, 64, 256
    x = rnd(3, 100, D, seed=1)
    w1, b1 = rnd(H, D, seed=2, scale=0.1).requires_grad_(True), rnd(H, seed=3, scale=0.1).requires_grad_(True)
    w2, b2 = rnd(D, H, seed=4, scale=0.1).requires_grad_(True), rnd(D, seed=5, scale=0.1).requires_grad_(True)
    dy = rnd(3, 100, D, seed=6)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya = Fn.ffn_relu(xa, w1, b1, w2, b2, 0.0)
    ya.backward(dy)
    ga = [t.grad.clone() for t in (xa, w1, b1, w2, b2)]
    for t in (w1, b1, w2, b2):
        t.grad = None
    yb = Fn.linear(Fn.dropout(Fn.linear(xb, w1, b1, act="relu"), 0.0), w2, b2)
    yb.backward(dy)
    gb = [t.grad for t in (xb, w1, b1, w2, b2)]
    res.append(check("ffn fused fwd", ya, yb, torch.float32, atol=1e-6))
    for nm, a, b_ in zip(("dx", "dw1", "db1", "dw2", "db2"), ga, gb):
        res.append(check(f"ffn fused {nm}", a, b_, torch.float32, atol=1e-4, rtol=1e-5))
    # dropout on: gradient of a linear functional of y wrt x must be consistent with the forward mask (finite-difference free check:
    # y is linear in x on the kept/active set, so <dy, J v> == <J^T dy, v> for a random direction v with the SAME mask)
    K.manual_seed(5)
    xa = x.clone().requires_grad_(True)
    K.reset_op_counter()
    y1 = Fn.ffn_relu(xa, w1, b1, w2, b2, 0.4)
    y1.backward(dy)
    v = xa.grad * (1e-3 / xa.grad.abs().mean())                                 # along the gradient: both sides are large and positive
    K.reset_op_counter()
    y2 = Fn.ffn_relu((x + v).requires_grad_(True), w1, b1, w2, b2, 0.4)        # same seed offset -> same dropout mask
    lhs = ((y2 - y1).detach() * dy).sum().item()
    rhs = (xa.grad * v).sum().item()
    res.append((abs(lhs - rhs) <= 2e-2 * max(abs(lhs), abs(rhs), 1e-6) + 1e-5, f"ffn fused dropout: <dy,Jv>={lhs:.6e} vs <J^T dy,v>={rhs:.6e}"))
    # Swish (the Conformer feed-forward): activation + dropout in the first GEMM's epilogue with the pre-activation as a second
    # output, swish' * dropmask in the epilogue of the data-gradient GEMM through w_2 -- against linear / act_dropout / linear,
    # fp32 (exact-fp32 kernels + stage pass) and bf16 (LDS-DMA kernels, vectorised epilogue; M = 4096 rows takes the 8-wave kernel)
    for dtype, (B, T, D, H), tolf, tolg in ((torch.float32, (3, 100, 64, 256), 2e-5, 2e-4), (torch.bfloat16, (16, 256, 384, 1536), 3e-2, 6e-2)):
        Fn.set_compute_dtype(dtype)
        try:
            x = rnd(B, T, D, seed=11)
            w1, b1 = rnd(H, D, seed=12, scale=0.1).requires_grad_(True), rnd(H, seed=13, scale=0.1).requires_grad_(True)
            w2, b2 = rnd(D, H, seed=14, scale=0.05).requires_grad_(True), rnd(D, seed=15, scale=0.1).requires_grad_(True)
            dy = rnd(B, T, D, seed=16).to(dtype)
            for p_drop in (0.0, 0.3):
                outs = []
                for fused in (True, False):
                    for t in (w1, b1, w2, b2):
                        t.grad = None
                    xi = x.clone().to(dtype).requires_grad_(True)
                    K.manual_seed(7)
                    K.reset_op_counter()
                    if fused:
                        y = Fn.ffn_act(xi, w1, b1, w2, b2, "swish", p_drop)
                    else:
                        y = Fn.linear(Fn.act_dropout(Fn.linear(xi, w1, b1), "swish", p_drop), w2, b2)
                    y.backward(dy)
                    outs.append([y.detach().float()] + [t.grad.detach().float().clone() for t in (xi, w1, b1, w2, b2)])
                names = ("y", "dx", "dw1", "db1", "dw2", "db2")
                for nm, a, b_ in zip(names, outs[0], outs[1]):
                    scale = float(b_.abs().max())
                    tol_ = tolf if nm == "y" else tolg
                    res.append(check(f"ffn swish fused vs composed {str(dtype)[6:]} p={p_drop} {nm}", a, b_, torch.float32, rtol=tol_, atol=tol_ * scale))
        finally:
            Fn.set_compute_dtype(torch.float32)
    return res


@case
@both_dtypes
def gemm_skinny(dtype):
    """M <= 64 dense projections (the decode-step shapes) take the weight-streaming kernel; every epilogue option,
    row counts off the 16-row MFMA tile, N off the 16-column tile, K off the k-step, fp32 output of bf16 inputs."""
    res = []
    for (M, N, Kd, seed) in [(1, 384, 384, 1), (16, 1152, 384, 2), (16, 384, 1536, 3), (17, 320, 384, 4), (40, 4, 384, 5),
                             (64, 100, 264, 6), (3, 37, 40, 7), (16, 1536, 384, 8)]:
        x, w, b = rnd(M, Kd, seed=seed, dtype=dtype), rnd(N, Kd, seed=seed + 10, dtype=dtype, scale=0.05), rnd(N, seed=seed + 20)
        r = rnd(M, N, seed=seed + 30, dtype=dtype)
        out = torch.empty(M, N, dtype=dtype, device=DEV)
        K.gemm(K.operand(x, Kd), K.operand(w, Kd), M, N, Kd, out, in_dtype=dtype, bias=b, act="relu", res=r, alpha=0.5)
        ref = torch.relu(0.5 * (x.float() @ w.float().t()) + b) + r.float()
        res.append(check(f"skinny[{dtype}] {M}x{N}x{Kd} bias+relu+res", out, ref, dtype))
        out32 = torch.full((M, N), 1.0, dtype=torch.float32, device=DEV)
        K.gemm(K.operand(x, Kd), K.operand(w, Kd), M, N, Kd, out32, in_dtype=dtype, accumulate=True)
        res.append(check(f"skinny[{dtype}] {M}x{N}x{Kd} fp32 accumulate", out32, x.float() @ w.float().t() + 1.0, dtype))
    # strided operands: a column slice of a packed activation (ld > K) against a row slice of a packed weight
    xp, wp = rnd(16, 3 * 96, seed=40, dtype=dtype), rnd(3 * 64, 96, seed=41, dtype=dtype, scale=0.1)
    out = torch.empty(16, 64, dtype=dtype, device=DEV)
    K.gemm(K.operand(xp, 3 * 96, offset=96), K.operand(wp, 96, offset=64 * 96), 16, 64, 96, out, in_dtype=dtype)
    res.append(check(f"skinny[{dtype}] strided slices", out, xp[:, 96:192].float() @ wp[64:128].float().t(), dtype))
    return res


@case
@both_dtypes
def gemm_dgrad_wgrad(dtype):
    res = []
    for (M, N, Kd, seed) in [(100, 70, 80, 1), (2016, 384, 384, 2), (64, 4, 384, 3), (4000, 256, 80, 4)]:
        x, w = rnd(M, Kd, seed=seed, dtype=dtype), rnd(N, Kd, seed=seed + 10, dtype=dtype, scale=0.05)
        dy = rnd(M, N, seed=seed + 40, dtype=dtype)
        # dgrad: dX[M,K] = dY[M,N] . W[N,K] : A = dY (KC), B(n=k', red=n') = W[n',k'] -> RC with ld=K
        dx = torch.empty(M, Kd, dtype=dtype, device=DEV)
        K.gemm(K.operand(dy, N), K.operand(w, Kd, layout=K.RC), M, Kd, N, dx, in_dtype=dtype)
        res.append(check(f"dgrad[{dtype}] {M}x{N}x{Kd}", dx, dy.float() @ w.float(), dtype))
        # wgrad: dW[N,K] = dY^T X : A(n, m) = dY[m,n] RC ld=N ; B(k, m) = X[m,k] RC ld=K ; reduction M
        for sk in (1, K.pick_splitk(N, Kd, M), 5):
            dw = torch.empty(N, Kd, dtype=torch.float32, device=DEV)
            K.gemm(K.operand(dy, N, layout=K.RC), K.operand(x, Kd, layout=K.RC), N, Kd, M, dw, in_dtype=dtype, splitk=sk)
            res.append(check(f"wgrad[{dtype}] {M}x{N}x{Kd} splitk={sk}", dw, dy.float().t() @ x.float(), dtype,
                             rtol=None if dtype == torch.float32 else 3e-2, atol=1e-4 * math.sqrt(M) if dtype == torch.float32 else 0.3))
    return res


@case
def gemm_grouped_wgrad():
    """Queued weight-gradient GEMMs run as grouped launches (no split-K) == fp32 torch, incl. ragged extents, fused bias
    row sums, accumulation into existing gradients, more problems than one launch holds (11), two problems writing the same
    gradient (must be serialised into successive launches) and an ineligible problem (launched directly while recording)."""
    res = []
    dtype = torch.bfloat16
    shapes = [(2016, 384, 384), (2016, 384, 1536), (2048, 1536, 384), (2016, 384, 80), (1000, 256, 384), (130, 72, 40), (2016, 7296, 384)]
    shapes += [(512, 128, 64 + 8 * i) for i in range(7)]           # 14 problems: two launches per flush
    probs = []
    for i, (rows, fin, fout) in enumerate(shapes):
        x, dy = rnd(rows, fin, seed=i, dtype=dtype), rnd(rows, fout, seed=100 + i, dtype=dtype)
        dw0, db0 = rnd(fout, fin, seed=200 + i), rnd(fout, seed=300 + i)
        probs.append((x, dy, dw0, db0))
    saved, saved_max = K._GROUP_TILE, K._GROUP_MAX_TILES
    big = []
    with K.record_grouped(big):      # default policy: an output with >= _GROUP_MAX_TILES 128x128 tiles is launched directly
        x, dy, dw0 = rnd(512, 3072, seed=50, dtype=dtype), rnd(512, 1664, seed=51, dtype=dtype), rnd(1664, 3072, seed=52)
        dw = dw0.clone()
        K.gemm(K.operand(dy, 1664, layout=K.RC), K.operand(x, 3072, layout=K.RC), 1664, 3072, 512, dw, in_dtype=dtype, accumulate=True)
        x8, dy8, dw8 = rnd(512, 3072, seed=53, dtype=dtype), rnd(512, 1536, seed=54, dtype=dtype), torch.zeros(1536, 3072, device=DEV)
        K.gemm(K.operand(dy8, 1536, layout=K.RC), K.operand(x8, 3072, layout=K.RC), 1536, 3072, 512, dw8, in_dtype=dtype, accumulate=True)
    res.append((len(big) == 1, "a chip-filling problem (312 tiles of 128x128) is not queued; one of exact 256x256 tiles is (8-wave grouped kernel)"))
    K.flush_grouped(big)
    res.append(check("queued 256-multiple problem after the flush", dw8, dy8.float().t() @ x8.float(), torch.float32, rtol=1e-3, atol=0.05))
    res.append(check("chip-filling problem launched directly", dw, dw0 + dy.float().t() @ x.float(), torch.float32, rtol=1e-3, atol=0.05))
    K._GROUP_MAX_TILES = 1 << 30
    for tile in (64, 128):
        K._GROUP_TILE = tile
        outs = [(p[2].clone(), p[3].clone()) for p in probs]
        queue = []
        with K.record_grouped(queue):
            for (x, dy, _, _), (dw, db) in zip(probs, outs):
                rows, fin, fout = x.shape[0], x.shape[1], dy.shape[1]
                K.gemm(K.operand(dy, fout, layout=K.RC), K.operand(x, fin, layout=K.RC), fout, fin, rows, dw, in_dtype=dtype, splitk=4,
                       accumulate=True, a_rowsum=db, a_rowsum_accumulate=True)
            x, dy = probs[0][0], probs[0][1]       # the first problem once more into the SAME gradient (a shared weight)
            K.gemm(K.operand(dy, 384, layout=K.RC), K.operand(x, 384, layout=K.RC), 384, 384, 2016, outs[0][0], in_dtype=dtype,
                   accumulate=True, a_rowsum=outs[0][1], a_rowsum_accumulate=True)
            y_now = torch.empty(2016, 384, dtype=dtype, device=DEV)      # K-contiguous operands: not eligible, runs immediately
            w_kc = rnd(384, 384, seed=9, dtype=dtype)
            K.gemm(K.operand(x, 384), K.operand(w_kc, 384), 2016, 384, 384, y_now, in_dtype=dtype)
        res.append((len(queue) == len(probs) + 1, f"tile={tile}: {len(queue)} problems queued"))
        res.append((bool(torch.equal(outs[1][0], probs[1][2])), "nothing is written before the flush"))
        res.append(check(f"tile={tile}: ineligible problem ran directly", y_now, x.float() @ w_kc.float().t(), dtype, atol=0.3))
        K.flush_grouped(queue)
        res.append((len(queue) == 0, "queue empty after the flush"))
        for i, ((x, dy, dw0, db0), (dw, db)) in enumerate(zip(probs, outs)):
            mult = 2.0 if i == 0 else 1.0
            dw_ref = dw0 + mult * (dy.float().t() @ x.float())
            db_ref = db0 + mult * dy.float().sum(0)
            res.append(check(f"grouped wgrad tile={tile} #{i} {tuple(dw.shape)} dW", dw, dw_ref, torch.float32, rtol=1e-3,
                             atol=2e-4 * float(dw_ref.abs().max())))
            res.append(check(f"grouped wgrad tile={tile} #{i} db", db, db_ref, torch.float32, rtol=1e-3, atol=2e-4 * float(db_ref.abs().max())))
    K._GROUP_TILE, K._GROUP_MAX_TILES = saved, saved_max
    return res


@case
@both_dtypes
def gemm_batched_attention(dtype):
    res = []
    B, H, T1, T2, dk = 3, 4, 37, 45, 96
    D = H * dk
    q, k, v = rnd(B, T1, D, seed=1, dtype=dtype), rnd(B, T2, D, seed=2, dtype=dtype), rnd(B, T2, D, seed=3, dtype=dtype)
    scores = torch.empty(B, H, T1, T2, dtype=torch.float32, device=DEV)
    K.gemm(K.operand(q, D, bs0=T1 * D, bs1=dk), K.operand(k, D, bs0=T2 * D, bs1=dk), T1, T2, dk, scores, in_dtype=dtype,
           nb0=B, nb1=H, cbs=(H * T1 * T2, T1 * T2))
    qh = q.float().view(B, T1, H, dk).transpose(1, 2)
    kh = k.float().view(B, T2, H, dk).transpose(1, 2)
    vh = v.float().view(B, T2, H, dk).transpose(1, 2)
    ref = qh @ kh.transpose(-1, -2)
    res.append(check(f"QK^T[{dtype}]", scores, ref, dtype, atol=1e-4 if dtype == torch.float32 else 0.3))
    p = torch.softmax(ref / math.sqrt(dk), -1).to(dtype)
    ctx = torch.empty(B, T1, D, dtype=dtype, device=DEV)
    # ctx[b, t1, h*dk + d] = sum_t2 p[b,h,t1,t2] v[b,t2,h*dk+d] : B(n=d, red=t2) = v -> RC ld=D
    K.gemm(K.operand(p, T2, bs0=H * T1 * T2, bs1=T1 * T2), K.operand(v, D, layout=K.RC, bs0=T2 * D, bs1=dk), T1, dk, T2, ctx,
           in_dtype=dtype, nb0=B, nb1=H, ldc=D, cbs=(T1 * D, dk))
    refc = (p.float() @ vh).transpose(1, 2).reshape(B, T1, D)
    res.append(check(f"PV[{dtype}]", ctx, refc, dtype))
    return res


@case
@both_dtypes
def gemm_conv1d(dtype):
    res = []
    for (B, T, Cin, Cout, ks, seed) in [(3, 50, 80, 256, 5, 1), (2, 33, 256, 80, 5, 2), (4, 64, 96, 96, 3, 3), (2, 40, 24, 40, 1, 4)]:
        pad = (ks - 1) // 2
        x = rnd(B, T, Cin, seed=seed, dtype=dtype)
        w = rnd(Cout, Cin, ks, seed=seed + 1, dtype=torch.float32, scale=0.05)
        b = rnd(Cout, seed=seed + 2)
        wp = K.gather3(w, (Cout, ks, Cin), (Cin * ks, 1, ks), 0, dtype)          # (O, k, I)
        y = torch.empty(B, T, Cout, dtype=dtype, device=DEV)
        K.gemm(K.operand(x, Cin, mode=K.CONV1D, C=Cin, T=T, pad=pad), K.operand(wp, ks * Cin), B * T, Cout, ks * Cin, y,
               in_dtype=dtype, bias=b)
        xr = x.float().transpose(1, 2).requires_grad_(True)
        wr = w.to(dtype).float().requires_grad_(True)
        yr = F.conv1d(xr, wr, b, padding=pad)
        res.append(check(f"conv1d fwd[{dtype}] {B}x{T}x{Cin}->{Cout} k{ks}", y, yr.transpose(1, 2), dtype))
        dy = rnd(B, T, Cout, seed=seed + 3, dtype=dtype)
        yr.backward(dy.float().transpose(1, 2))
        # dgrad: conv of dY with flipped taps; Wd[c][j'][o] = W[o, c, k-1-j']
        wd = K.gather3(w, (Cin, ks, Cout), (ks, -1, Cin * ks), ks - 1, dtype)
        dx = torch.empty(B, T, Cin, dtype=dtype, device=DEV)
        K.gemm(K.operand(dy, Cout, mode=K.CONV1D, C=Cout, T=T, pad=pad), K.operand(wd, ks * Cout), B * T, Cin, ks * Cout, dx,
               in_dtype=dtype)
        res.append(check(f"conv1d dgrad[{dtype}] k{ks}", dx, xr.grad.transpose(1, 2), dtype))
        # wgrad: dWp[o, (j,c)] = sum_m dY[m,o] X[m+j-pad, c]
        dwp = torch.empty(Cout, ks * Cin, dtype=torch.float32, device=DEV)
        K.gemm(K.operand(dy, Cout, layout=K.RC), K.operand(x, Cin, layout=K.RC, mode=K.CONV1D, C=Cin, T=T, pad=pad), Cout,
               ks * Cin, B * T, dwp, in_dtype=dtype, splitk=3)
        dw = K.gather3(dwp, (Cout, Cin, ks), (ks * Cin, 1, Cin), 0, torch.float32)
        res.append(check(f"conv1d wgrad[{dtype}] k{ks}", dw, wr.grad, dtype, atol=1e-4 if dtype == torch.float32 else 0.3))
    return res


@case
def gemm_conv1d_on_8wave_kernel():
    """Conv1d (stride 1, 'same') as an implicit GEMM on the 8-wave kernel (bf16; csrc/gemm_8ph.hip P8_CONV1D): the three tile
    geometries (256 x 128, 256 x 256, 512 x 128), kernel widths 3 and 5, utterance boundaries inside a tile (T = 96 / 160), bias +
    ReLU and the plain data-gradient form -- against fp32 torch on the same stored operands and against the 4-wave kernel."""
    res = []
    dtype = torch.bfloat16
    L = K._lib.lib()
    for (B, T, Cin, Cout, ks, act, seed) in [(16, 256, 128, 1024, 3, "relu", 1), (64, 128, 64, 1536, 3, None, 2), (128, 128, 64, 640, 5, "relu", 3),
                                             (32, 96, 192, 1152, 3, None, 4), (40, 160, 128, 1024, 5, "relu", 5)]:
        pad = (ks - 1) // 2
        x = rnd(B, T, Cin, seed=seed, dtype=dtype)
        w = rnd(Cout, Cin, ks, seed=seed + 1, dtype=torch.float32, scale=0.05)
        b = rnd(Cout, seed=seed + 2) if act else None
        wp = K.gather3(w, (Cout, ks, Cin), (Cin * ks, 1, ks), 0, dtype)          # (O, k, I)
        yr = F.conv1d(x.float().transpose(1, 2), w.to(dtype).float(), b, padding=pad).transpose(1, 2)
        if act:
            yr = torch.relu(yr)
        outs = []
        for on in (1, 0):
            prev = L.s2svc_gemm_set_8ph(on)
            y = torch.empty(B, T, Cout, dtype=dtype, device=DEV)
            K.gemm(K.operand(x, Cin, mode=K.CONV1D, C=Cin, T=T, pad=pad), K.operand(wp, ks * Cin), B * T, Cout, ks * Cin, y,
                   in_dtype=dtype, bias=b, act=act)
            L.s2svc_gemm_set_8ph(prev)
            res.append(check(f"conv1d on 8-wave={on} {B}x{T}x{Cin}->{Cout} k{ks} act={act}", y, yr, dtype))
            outs.append(y)
        res.append(check(f"conv1d 8-wave vs 4-wave {B}x{T}x{Cin}->{Cout} k{ks}", outs[0], outs[1].float(), dtype, rtol=1e-2, atol=1e-2))
    return res


@case
def tconv2d_weight_matrices():
    """The four parity-class weight matrices of the stride-2 transposed convolution (LDS-tiled kernel for O, C % 32 == 0, element-wise
    otherwise) against an index-by-index torch construction: bit-exact (a bf16 rounding of the same fp32 values)."""
    res = []
    for (O, C, seed) in [(64, 64, 1), (384, 384, 2), (72, 64, 3), (96, 160, 4)]:
        w = rnd(O, C, 3, 3, seed=seed, scale=0.3)
        got = K.tconv2d_weights(w)
        for cls in range(4):
            pt, pf = cls >> 1, cls & 1
            taps = [(pt + 2 * ta, pf + 2 * fb) for ta in range(2 - pt) for fb in range(2 - pf)]
            ref = torch.stack([w[:, :, kh, kw].t() for kh, kw in taps], 1).reshape(C, len(taps) * O).to(torch.bfloat16)   # [c][tap * O + o]
            ok = bool(torch.equal(got[cls], ref))
            res.append((ok, f"tconv2d weights O{O} C{C} class {cls}: bit-exact={ok}"))
    return res


@case
def permute_inner_accumulate():
    """dst[o][b][a] (+)= src[o][a][b] through LDS (a convolution weight gradient into the parameter's layout): bit-exact against torch."""
    res = []
    for (n, A, Bn, seed) in [(384, 9, 384, 1), (384, 19, 384, 2), (1536, 3, 1536, 3), (7, 5, 33, 4), (3, 1, 8, 5)]:
        src = rnd(n, A, Bn, seed=seed)
        base = rnd(n, Bn, A, seed=seed + 1)
        ref = src.permute(0, 2, 1).contiguous()
        out = K.permute_inner(src, n, A, Bn).view(n, Bn, A)
        res.append((bool(torch.equal(out, ref)), f"permute_inner n{n} A{A} B{Bn}: bit-exact={bool(torch.equal(out, ref))}"))
        acc = base.clone()
        K.permute_inner(src, n, A, Bn, out=acc, accumulate=True)
        res.append((bool(torch.equal(acc, base + ref)), f"permute_inner accumulate n{n} A{A} B{Bn}: bit-exact={bool(torch.equal(acc, base + ref))}"))
    # a slot of the wrong dtype / size / layout is refused, never written through
    src = rnd(3, 2, 8, seed=9)
    for nm, bad in (("bf16 slot", torch.zeros(48, dtype=torch.bfloat16, device=DEV)), ("short slot", torch.zeros(40, device=DEV)),
                    ("strided slot", torch.zeros(96, device=DEV)[::2])):
        try:
            K.permute_inner(src, 3, 2, 8, out=bad, accumulate=True)
            res.append((False, f"permute_inner accepted a {nm}"))
        except TypeError:
            res.append((True, f"permute_inner refuses a {nm}"))
    return res


@case
def conv1d_wgrad_on_w8():
    """Conv1d weight gradient on the ragged 8-wave weight-gradient kernel (implicit im2col B operand, kind 1): kernel widths 3 / 5,
    utterance boundaries inside K tiles (T = 96), a reduction that ends inside a K tile, bias row sums -- against fp32 torch."""
    res = []
    dtype = torch.bfloat16
    for (B, T, Cin, Cout, ks, seed) in [(16, 256, 1536, 1536, 3, 1), (5, 96, 256, 2048, 5, 2), (3, 50, 128, 4096, 3, 3)]:
        pad = (ks - 1) // 2
        x = rnd(B, T, Cin, seed=seed, dtype=dtype)
        dy = rnd(B, T, Cout, seed=seed + 3, dtype=dtype)
        wr = torch.zeros(Cout, Cin, ks, device=DEV, requires_grad=True)
        yr = F.conv1d(x.float().transpose(1, 2), wr, None, padding=pad)
        yr.backward(dy.float().transpose(1, 2))
        dbr = dy.float().sum((0, 1))
        for w8 in (True, False):
            dwp = torch.full((Cout, ks * Cin), 7.0, dtype=torch.float32, device=DEV)
            db = torch.full((Cout,), 3.0, dtype=torch.float32, device=DEV)
            K.gemm(K.operand(dy, Cout, layout=K.RC), K.operand(x, Cin, layout=K.RC, mode=K.CONV1D, C=Cin, T=T, pad=pad), Cout,
                   ks * Cin, B * T, dwp, in_dtype=dtype, splitk=1, a_rowsum=db, a_rowsum_accumulate=False, wgrad=w8)
            dw = K.gather3(dwp, (Cout, Cin, ks), (ks * Cin, 1, Cin), 0, torch.float32)
            sc = max(float(wr.grad.abs().max()), 1.0)
            tag = f"conv1d wgrad w8={int(w8)} {B}x{T} {Cin}->{Cout} k{ks}"
            res.append(check(tag, dw, wr.grad, torch.float32, rtol=1e-4, atol=2e-4 * sc))
            res.append(check(tag + " bias", db, dbr, torch.float32, rtol=1e-4, atol=2e-4 * max(float(dbr.abs().max()), 1.0)))
    return res


@case
@both_dtypes
def gemm_conv2d(dtype):
    res = []
    B, T1, F1, C, O = 2, 31, 19, 32, 48
    T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
    x = rnd(B, T1, F1, C, seed=1, dtype=dtype)          # NHWC
    w = rnd(O, C, 3, 3, seed=2, scale=0.05)
    b = rnd(O, seed=3)
    wp = K.gather3(w, (O, 9, C), (C * 9, 1, 9), 0, dtype)  # (O, tap, C)
    y = torch.empty(B, T2, F2, O, dtype=dtype, device=DEV)
    K.gemm(K.operand(x, C, mode=K.CONV2D_S2, C=C, T1=T1, F1=F1, T2=T2, F2=F2), K.operand(wp, 9 * C), B * T2 * F2, O, 9 * C, y,
           in_dtype=dtype, bias=b, act="relu")
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    wr = w.to(dtype).float().requires_grad_(True)
    yr = torch.relu(F.conv2d(xr, wr, b, stride=2))
    res.append(check(f"conv2d fwd[{dtype}]", y, yr.permute(0, 2, 3, 1), dtype))
    dy = rnd(B, T2, F2, O, seed=4, dtype=dtype)
    dyr = dy.float() * (yr.permute(0, 2, 3, 1) > 0)
    yr.backward(dy.float().permute(0, 3, 1, 2))
    dym = dyr.to(dtype).contiguous()
    dwp = torch.empty(O, 9 * C, dtype=torch.float32, device=DEV)
    K.gemm(K.operand(dym, O, layout=K.RC), K.operand(x, C, layout=K.RC, mode=K.CONV2D_S2, C=C, T1=T1, F1=F1, T2=T2, F2=F2), O,
           9 * C, B * T2 * F2, dwp, in_dtype=dtype, splitk=2)
    dw = K.gather3(dwp, (O, C, 9), (9 * C, 1, C), 0, torch.float32).view(O, C, 3, 3)
    res.append(check(f"conv2d wgrad[{dtype}]", dw, wr.grad, dtype, atol=1e-4 if dtype == torch.float32 else 0.3))
    return res


@case
def gemm_conv2d_dma_addressing():
    """The LDS-DMA GEMM's implicit 3x3 stride-2 convolution operands (C >= 64) on geometries that take each addressing
    path: scalar-base rows with the division-free (tap, channel) walk; the conv2d weight gradient's per-lane pixel walk
    (wraps over f2 / t2 / utterance, F2 > 64, a split-K start in the middle of the image); the per-lane fallbacks
    (C % 64 != 0, T2 too small for the single-wrap walk); row / column tails.  Forward and weight gradient vs torch."""
    res = []
    dtype = torch.bfloat16
    for (B, T1, F1, C, O, seed) in [(3, 31, 23, 64, 64, 1), (2, 9, 135, 64, 64, 2), (5, 5, 23, 64, 64, 3), (2, 33, 21, 72, 64, 4),
                                    (2, 127, 39, 128, 72, 5)]:
        T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
        x = rnd(B, T1, F1, C, seed=seed, dtype=dtype)
        w = rnd(O, C, 3, 3, seed=seed + 10, scale=0.05)
        wp = K.gather3(w, (O, 9, C), (C * 9, 1, 9), 0, dtype)
        y = torch.empty(B, T2, F2, O, dtype=dtype, device=DEV)
        K.gemm(K.operand(x, C, mode=K.CONV2D_S2, C=C, T1=T1, F1=F1, T2=T2, F2=F2), K.operand(wp, 9 * C), B * T2 * F2, O, 9 * C, y,
               in_dtype=dtype)
        wr = w.to(dtype).float().requires_grad_(True)
        yr = F.conv2d(x.float().permute(0, 3, 1, 2), wr, None, stride=2)
        tag = f"B{B} {T1}x{F1} C{C} O{O}"
        res.append(check(f"conv2d fwd (DMA GEMM) {tag}", y, yr.permute(0, 2, 3, 1), dtype))
        dy = rnd(B, T2, F2, O, seed=seed + 20, dtype=dtype)
        yr.backward(dy.float().permute(0, 3, 1, 2))
        for sk in (1, 3):
            dwp = torch.empty(O, 9 * C, dtype=torch.float32, device=DEV)
            K.gemm(K.operand(dy, O, layout=K.RC), K.operand(x, C, layout=K.RC, mode=K.CONV2D_S2, C=C, T1=T1, F1=F1, T2=T2, F2=F2), O,
                   9 * C, B * T2 * F2, dwp, in_dtype=dtype, splitk=sk)
            dw = K.gather3(dwp, (O, C, 9), (9 * C, 1, C), 0, torch.float32).view(O, C, 3, 3)
            res.append(check(f"conv2d wgrad (DMA GEMM) {tag} splitk={sk}", dw, wr.grad, dtype, rtol=3e-2, atol=0.3))
    # dense operands: K tail (the last k tile takes the per-lane path), row tails, a row-contiguous operand whose row count is
    # not a multiple of 8 inside a padded buffer
    for (M, N, Kd, seed) in [(200, 136, 200, 6), (515, 384, 448, 7)]:
        a, b = rnd(M, Kd, seed=seed, dtype=dtype), rnd(N, Kd, seed=seed + 1, dtype=dtype, scale=0.05)
        c = torch.empty(M, N, dtype=dtype, device=DEV)
        K.gemm(K.operand(a, Kd), K.operand(b, Kd), M, N, Kd, c, in_dtype=dtype)
        res.append(check(f"dense fwd (DMA GEMM) {M}x{N}x{Kd}", c, a.float() @ b.float().t(), dtype))
    M, N, Kd = 1000, 20, 384
    buf = rnd(M, 24, seed=8, dtype=dtype)
    dy, x = buf[:, :N], rnd(M, Kd, seed=9, dtype=dtype)
    dw = torch.empty(N, Kd, dtype=torch.float32, device=DEV)
    K.gemm(K.operand(dy, 24, layout=K.RC), K.operand(x, Kd, layout=K.RC), N, Kd, M, dw, in_dtype=dtype)
    res.append(check("wgrad with a 20-row operand in a 24-wide buffer", dw, dy.float().t() @ x.float(), dtype, rtol=3e-2, atol=0.3))
    return res


@case
def conv2d_wgrad_on_w8():
    """Conv2d 3x3 stride 2 weight gradient on the ragged 8-wave weight-gradient kernel (implicit im2col B operand, C % 128 == 0):
    multi-chunk reductions (K tiles beyond one chunk), a reduction that ends inside a K tile, ragged O, bias row sums -- against
    fp32 torch and against the 4-wave split-K kernel on the same operands."""
    res = []
    dtype = torch.bfloat16
    for (B, T1, F1, C, O, seed) in [(2, 127, 39, 128, 72, 1), (5, 127, 39, 384, 384, 2), (3, 33, 21, 256, 200, 3), (1, 9, 7, 128, 64, 4)]:
        T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
        M2 = B * T2 * F2
        x = rnd(B, T1, F1, C, seed=seed, dtype=dtype)
        dy = rnd(B, T2, F2, O, seed=seed + 20, dtype=dtype)
        wr = torch.zeros(O, C, 3, 3, device=DEV, requires_grad=True)
        yr = F.conv2d(x.float().permute(0, 3, 1, 2), wr, None, stride=2)
        yr.backward(dy.float().permute(0, 3, 1, 2))
        dbr = dy.float().sum((0, 1, 2))
        tag = f"B{B} {T1}x{F1} C{C} O{O}"
        outs = []
        for w8 in (True, False):
            dwp = torch.full((O, 9 * C), 7.0, dtype=torch.float32, device=DEV)      # (not accumulated into: the fill must vanish)
            db = torch.full((O,), 3.0, dtype=torch.float32, device=DEV)
            K.gemm(K.operand(dy, O, layout=K.RC), K.operand(x, C, layout=K.RC, mode=K.CONV2D_S2, C=C, T1=T1, F1=F1, T2=T2, F2=F2), O,
                   9 * C, M2, dwp, in_dtype=dtype, splitk=1, a_rowsum=db, a_rowsum_accumulate=False, wgrad=w8)
            dw = K.gather3(dwp, (O, C, 9), (9 * C, 1, C), 0, torch.float32).view(O, C, 3, 3)
            sc = max(float(wr.grad.abs().max()), 1.0)
            res.append(check(f"conv2d wgrad w8={int(w8)} {tag}", dw, wr.grad, torch.float32, rtol=1e-4, atol=2e-4 * sc))
            res.append(check(f"conv2d wgrad w8={int(w8)} {tag} bias", db, dbr, torch.float32, rtol=1e-4, atol=2e-4 * max(float(dbr.abs().max()), 1.0)))
            outs.append(dw)
    return res


@case
def conv2d_dgrad_transposed():
    """Data gradient of the 3x3 stride-2 Conv2d as four implicit transposed-convolution GEMMs (one per parity class of
    input pixels, stored through the c_map) vs torch's conv2d input gradient and vs the dcols GEMM + col2im path, for
    odd / even input extents (rows the convolution never touched must come out 0)."""
    res = []
    dtype = torch.bfloat16
    for (B, T1, F1, C, O, seed) in [(2, 31, 19, 64, 64, 1), (3, 32, 20, 64, 128, 2), (1, 9, 8, 128, 64, 3), (2, 127, 39, 64, 64, 4)]:
        T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
        w = rnd(O, C, 3, 3, seed=seed, scale=0.05)
        dy = rnd(B, T2, F2, O, seed=seed + 10, dtype=dtype)
        dx = torch.full((B, T1, F1, C), float("nan"), dtype=dtype, device=DEV)       # every pixel must be written
        for cls, wt in enumerate(K.tconv2d_weights(w)):
            pt, pf = cls >> 1, cls & 1
            Tc, Fc = (T1 - pt + 1) // 2, (F1 - pf + 1) // 2
            K.gemm(K.operand(dy, O, mode=K.TCONV2D_S2, C=O, T1=Tc, F1=Fc, T2=T2, F2=F2, pad=cls), K.operand(wt, wt.shape[1]),
                   B * Tc * Fc, C, wt.shape[1], dx, in_dtype=dtype, c_map=(T1, F1, Tc, Fc, pt, pf))
        dxg = torch.full((B, T1, F1, C), float("nan"), dtype=dtype, device=DEV)      # the four classes as ONE grid
        descs, wts = [], K.tconv2d_weights(w)
        for cls, wt in enumerate(wts):
            pt, pf = cls >> 1, cls & 1
            Tc, Fc = (T1 - pt + 1) // 2, (F1 - pf + 1) // 2
            K.gemm(K.operand(dy, O, mode=K.TCONV2D_S2, C=O, T1=Tc, F1=Fc, T2=T2, F2=F2, pad=cls), K.operand(wt, wt.shape[1]),
                   B * Tc * Fc, C, wt.shape[1], dxg, in_dtype=dtype, c_map=(T1, F1, Tc, Fc, pt, pf), group=descs)
        K.launch_group(descs)
        res.append((bool(torch.equal(dxg, dx)), f"tconv2d dgrad B{B} {T1}x{F1}: grouped launch == four launches (bit-exact)"))
        xr = torch.zeros(B, C, T1, F1, device=DEV, requires_grad=True)
        F.conv2d(xr, w.to(dtype).float(), None, stride=2).backward(dy.float().permute(0, 3, 1, 2))
        ref = xr.grad.permute(0, 2, 3, 1)
        res.append(check(f"tconv2d dgrad B{B} {T1}x{F1} C{C} O{O} vs torch", dx, ref, dtype, atol=3e-2 * max(1.0, float(ref.abs().max()))))
        wp = K.gather3(w, (O, 9, C), (C * 9, 1, 9), 0, dtype)
        dcols = torch.empty(B * T2 * F2, 9 * C, dtype=dtype, device=DEV)
        K.gemm(K.operand(dy, O), K.operand(wp, 9 * C, layout=K.RC), B * T2 * F2, 9 * C, O, dcols, in_dtype=dtype)
        old = K.col2im_s2(dcols, B, T1, F1, C, T2, F2)
        res.append(check(f"tconv2d dgrad B{B} {T1}x{F1} vs dcols+col2im", dx, old, dtype, atol=3e-2 * max(1.0, float(ref.abs().max()))))
    # the VTN front-end at its recipe size: the four class GEMMs run on the 8-wave kernel (csrc/gemm_8ph.hip: masked DMA for the
    # taps that fall outside the output-gradient image, c_map stores) -- against the 128 x 128 kernel and torch
    L = K._lib.lib()
    prev = L.s2svc_gemm_set_8ph(-1)
    try:
        B, T1, F1, C, O = 32, 127, 39, 384, 384
        T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
        w = rnd(O, C, 3, 3, seed=7, scale=0.02)
        dy = rnd(B, T2, F2, O, seed=17, dtype=dtype)
        outs = {}
        for mode in (0, 1, 2):
            L.s2svc_gemm_set_8ph(mode)
            dx = torch.full((B, T1, F1, C), float("nan"), dtype=dtype, device=DEV)
            for cls, wt in enumerate(K.tconv2d_weights(w)):
                pt, pf = cls >> 1, cls & 1
                Tc, Fc = (T1 - pt + 1) // 2, (F1 - pf + 1) // 2
                K.gemm(K.operand(dy, O, mode=K.TCONV2D_S2, C=O, T1=Tc, F1=Fc, T2=T2, F2=F2, pad=cls), K.operand(wt, wt.shape[1]),
                       B * Tc * Fc, C, wt.shape[1], dx, in_dtype=dtype, c_map=(T1, F1, Tc, Fc, pt, pf))
            outs[mode] = dx
        xr = torch.zeros(B, C, T1, F1, device=DEV, requires_grad=True)
        F.conv2d(xr, w.to(dtype).float(), None, stride=2).backward(dy.float().permute(0, 3, 1, 2))
        ref = xr.grad.permute(0, 2, 3, 1)
        tol = 3e-2 * max(1.0, float(ref.abs().max()))
        res.append(check("tconv2d dgrad B32 127x39 C384 (8-wave kernel) vs torch", outs[1], ref, dtype, atol=tol))
        res.append((bool(torch.equal(outs[1], outs[2])), "tconv2d dgrad on the 8-wave kernel: skewed and lockstep wave halves agree bit for bit"))
        same = float((outs[1].float() - outs[0].float()).abs().max())
        res.append((same <= 0.02 * max(1.0, float(ref.abs().max())), f"tconv2d dgrad 8-wave vs 128x128 kernel: max diff {same:.3e}"))
    finally:
        L.s2svc_gemm_set_8ph(prev)
    return res


@case
def conv2d_subsampling_frontend():
    """Conv2d(1,C,3,2)+ReLU -> Conv2d(C,C,3,2)+ReLU -> Linear front-end in fp32, forward and every gradient, against torch
    (direct C_in=1 kernels, implicit-GEMM conv, col2im, relu' fused into the consumers).  The bf16 legs of the same
    kernels are checked one by one below (a composite bf16 run differs from torch by relu-mask flips of near-zero
    activations, which is noise, not a defect)."""
    from seq2seq_vc_amd.modules import Conv2dSubsampling, Lens
    res = []
    dtype = torch.float32
    B, T, idim, C = 3, 67, 80, 64
    torch.manual_seed(5)
    m = Conv2dSubsampling(idim, C, 0.0, use_pos_enc=False).to(DEV)
    x = rnd(B, T, idim, seed=1, dtype=dtype)
    params = list(m.parameters())
    y, _ = m(x, Lens([T, 50, 33], DEV))
    dy = rnd(*y.shape, seed=2, dtype=dtype)
    y.backward(dy)
    got = [p.grad.clone() for p in params]
    ws = [p.detach().clone().requires_grad_(True) for p in params]
    h = torch.relu(F.conv2d(x.unsqueeze(1), ws[0], ws[1], stride=2))
    h = torch.relu(F.conv2d(h, ws[2], ws[3], stride=2))
    b_, c_, t_, f_ = h.shape
    yr = F.linear(h.transpose(1, 2).contiguous().view(b_, t_, c_ * f_), ws[4], ws[5])
    yr.backward(dy)
    res.append(check("frontend fwd[fp32]", y, yr, dtype))
    for n, a_, r in zip(["conv0.w", "conv0.b", "conv2.w", "conv2.b", "out.w", "out.b"], got, ws):
        res.append(check(f"frontend d{n}[fp32]", a_, r.grad, dtype, rtol=1e-4, atol=2e-5 * max(float(r.grad.abs().max()), 1.0)))
    return res


@case
def linear_fc_permuted_backward_bf16():
    """The Linear behind the Conv2d front-end (subsampling.py:64-70) at VTN's width, bf16: forward, data gradient (with relu' of the
    input in the epilogue) and weight / bias gradients against fp32 math on the same stored operands -- on the default kernels
    (transposed permuted weight copy on the 8-wave kernel; weight gradient on the ragged 8-wave kernel) and with both switched off."""
    from seq2seq_vc_amd.ops import functional as Fn
    res = []
    dtype = torch.bfloat16
    for (M, C, Fd, D, seed) in [(2016, 384, 19, 384, 1), (300, 64, 19, 256, 2)]:
        x = torch.relu(rnd(M, Fd * C, seed=seed, dtype=dtype))                       # (f, c) channel-last columns
        w0 = rnd(D, C * Fd, seed=seed + 1, scale=0.02)
        b0 = rnd(D, seed=seed + 2, scale=0.1)
        dy = rnd(M, D, seed=seed + 3, dtype=dtype)
        wq = w0.to(dtype).float()
        wperm = wq.view(D, C, Fd).permute(0, 2, 1).reshape(D, Fd * C)               # columns in (f, c) order
        yr = x.float() @ wperm.t() + b0
        dxr = (dy.float() @ wperm) * (x.float() > 0)
        dwr = (dy.float().t() @ x.float()).view(D, Fd, C).permute(0, 2, 1).reshape(D, C * Fd)
        dbr = dy.float().sum(0)
        xx = x.clone().requires_grad_(True)
        w, b = w0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        y = Fn.linear_fc_permuted(xx, w, b, C, Fd, input_is_relu=True)
        y.backward(dy)
        tag = f"fc_permuted[bf16] M{M} C{C} D{D}"
        res.append(check(f"{tag} fwd", y, yr, dtype))
        res.append(check(f"{tag} dx", xx.grad, dxr, dtype))
        sc = max(float(dwr.abs().max()), 1.0)
        res.append(check(f"{tag} dw", w.grad, dwr, torch.float32, rtol=1e-4, atol=2e-4 * sc))
        res.append(check(f"{tag} db", b.grad, dbr, torch.float32, rtol=1e-4, atol=2e-4 * max(float(dbr.abs().max()), 1.0)))
    return res


@case
@both_dtypes
def conv_in1_and_col2im(dtype):
    """The C_in = 1 streaming kernels and the stride-2 col2im gather, each against fp32 math on the SAME stored operands
    (so bf16 differs only by the output rounding / fp32 summation order)."""
    res = []
    for (B, T, Fd, O, seed) in [(3, 67, 80, 64, 1), (2, 40, 80, 384, 2), (1, 9, 11, 8, 3)]:
        x = rnd(B, T, Fd, seed=seed, dtype=dtype)
        w, b = rnd(O, 1, 3, 3, seed=seed + 1, scale=0.3), rnd(O, seed=seed + 2, scale=0.1)
        y = K.conv_in1_fwd(x, w, b)
        yr = torch.relu(F.conv2d(x.float().unsqueeze(1), w, b, stride=2)).permute(0, 2, 3, 1)
        res.append(check(f"conv_in1 fwd[{dtype}] B{B} T{T} F{Fd} O{O}", y, yr, dtype))
        dy = rnd(*y.shape, seed=seed + 3, dtype=dtype)
        for use_y in (True, False):
            dw, db = torch.zeros(O, 1, 3, 3, device=DEV), torch.zeros(O, device=DEV)
            K.conv_in1_wgrad(x, dy, dw, db, False, y=y if use_y else None)
            gm = dy.float() * (y.float() > 0) if use_y else dy.float()
            cols = F.unfold(x.float().unsqueeze(1), 3, stride=2)                      # (B, 9, T1*F1)
            dwr = torch.einsum("bkp,bpo->ok", cols, gm.reshape(B, -1, O)).view(O, 1, 3, 3)
            dbr = gm.sum((0, 1, 2))
            sc = max(float(dwr.abs().max()), 1.0)
            res.append(check(f"conv_in1 wgrad[{dtype}] O{O} mask={use_y} dw", dw, dwr, torch.float32, rtol=1e-4, atol=1e-4 * sc))
            res.append(check(f"conv_in1 wgrad[{dtype}] O{O} mask={use_y} db", db, dbr, torch.float32, rtol=1e-4, atol=1e-4 * sc))
            K.conv_in1_wgrad(x, dy, dw, db, True, y=y if use_y else None)         # accumulate: exactly twice
            res.append(check(f"conv_in1 wgrad[{dtype}] O{O} accumulate", dw, 2 * dwr, torch.float32, rtol=1e-4, atol=2e-4 * sc))
    for (B, T1, F1, C, seed) in [(2, 33, 39, 64, 4), (1, 10, 8, 24, 5), (2, 12, 7, 20, 6)]:
        T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
        dcols = rnd(B * T2 * F2, 9 * C, seed=seed, dtype=dtype)
        dx = K.col2im_s2(dcols, B, T1, F1, C, T2, F2)
        # fold wants (B, C*9, L) with channel-major rows; ours is tap-major (9, C)
        cr = dcols.float().view(B, T2 * F2, 9, C).permute(0, 3, 2, 1).reshape(B, C * 9, T2 * F2)
        dxr = F.fold(cr, (T1, F1), 3, stride=2).permute(0, 2, 3, 1)
        res.append(check(f"col2im_s2[{dtype}] B{B} {T1}x{F1} C{C}", dx, dxr, dtype, atol=None if dtype == torch.float32 else 5e-2))
    return res


@case
def layernorm_bwd_with_partial_gradients():
    """s2svc_layernorm_bwd_pg (ln_bwd_vec_pg_kernel + colreduce mode 7): the LayerNorm backward pass of big bf16 sites that also
    writes the first reduction stage of d gamma / d beta -- ds / dh bit-identical to the plain kernel (same row code), the parameter
    gradients (partials summed by the grouped second stage, accumulating into slots) against the separate mode-1 reduction and fp32
    torch; AAS-VC's decoder / encoder sites, ragged rows, D = 1000, residual-stream gradient added, dropped copy."""
    res = []
    dtype = torch.bfloat16
    for (rows, D, seed, extra, p) in [(4096, 1536, 1, True, 0.2), (4096, 384, 2, False, 0.0), (4100, 1000, 3, True, 0.0), (2048, 1536, 4, False, 0.1)]:
        s_in = rnd(rows, D, seed=seed, dtype=dtype)
        dy = rnd(rows, D, seed=seed + 1, dtype=dtype)
        gm = 1 + 0.1 * rnd(D, seed=seed + 2)
        ex = rnd(rows, D, seed=seed + 3, dtype=dtype) if extra else None
        sf = s_in.float()
        mean, var = sf.mean(1), sf.var(1, unbiased=False)
        rstd = torch.rsqrt(var + 1e-12)
        sd = K.new_seed(s_in.device) if p > 0 else (None, 0)
        ds0, dh0 = K.layernorm_bwd(dy, s_in, mean, rstd, gm, ds_extra=ex, p=p, seed=sd, want_dh=p > 0, hscale=1.0)
        ds1, dh1, part = K.layernorm_bwd(dy, s_in, mean, rstd, gm, ds_extra=ex, p=p, seed=sd, want_dh=p > 0, hscale=1.0, want_partials=True)
        res.append((part is not None, f"layernorm_bwd_pg takes {rows} x {D}"))
        if part is None:
            continue
        res.append((bool(torch.equal(ds0, ds1)) and (dh0 is None or bool(torch.equal(dh0, dh1))), f"layernorm_bwd_pg {rows} x {D}: ds / dh bit-identical to the plain kernel"))
        ws, chunks = part
        db0, dg0 = rnd(D, seed=seed + 5), rnd(D, seed=seed + 6)
        db, dg = db0.clone(), dg0.clone()
        K.colreduce_partials(ws, chunks, D, db, dg)
        xh = (sf - mean[:, None]) * rstd[:, None]
        rb, rg = dy.float().sum(0), (dy.float() * xh).sum(0)
        res.append(check(f"layernorm_bwd_pg {rows} x {D}: d beta (accumulated into its slot)", db, db0 + rb, torch.float32, rtol=1e-4, atol=1e-4 * float(rb.abs().max())))
        res.append(check(f"layernorm_bwd_pg {rows} x {D}: d gamma", dg, dg0 + rg, torch.float32, rtol=1e-4, atol=1e-4 * float(rg.abs().max())))
        sb, sg = K.colreduce(1, dy, s_in, mean, rstd, want_dot=True)
        d = max(float((sb - (db - db0)).abs().max()) / float(rb.abs().max()), float((sg - (dg - dg0)).abs().max()) / float(rg.abs().max()))
        res.append((d <= 1e-4, f"layernorm_bwd_pg {rows} x {D} vs the separate mode-1 reduction: largest relative difference {d:.2e} (<= 1e-4)"))
        db2, dg2 = db0.clone(), dg0.clone()
        _, _, part2 = K.layernorm_bwd(dy, s_in, mean, rstd, gm, ds_extra=ex, p=p, seed=sd, want_dh=p > 0, hscale=1.0, want_partials=True)
        K.colreduce_partials(part2[0], part2[1], D, db2, dg2)
        res.append((bool(torch.equal(db, db2) and torch.equal(dg, dg2)), f"layernorm_bwd_pg {rows} x {D}: repeated launches agree bit for bit"))
    _, _, none = K.layernorm_bwd(rnd(2016, 384, seed=9, dtype=dtype), rnd(2016, 384, seed=10, dtype=dtype), torch.zeros(2016, device=DEV),
                                 torch.ones(2016, device=DEV), torch.ones(384, device=DEV), want_partials=True)
    res.append((none is None, "layernorm_bwd_pg declines the small (VTN-sized) sites: they keep the queued mode-1 reduction"))
    return res


@case
@both_dtypes
def layernorm(dtype):
    res = []
    for (rows, D, seed) in [(37, 384, 1), (2016, 384, 2), (100, 1536, 3), (9, 50, 4), (16, 768, 5), (5, 512, 6)]:
        x = rnd(rows, D, seed=seed, dtype=dtype)
        r = rnd(rows, D, seed=seed + 1, dtype=dtype)
        gm, bt = 1 + 0.1 * rnd(D, seed=seed + 2), 0.1 * rnd(D, seed=seed + 3)
        dy = rnd(rows, D, seed=seed + 4, dtype=dtype)
        for use_res in (False, True):
            xr = x.float().requires_grad_(True)
            rr = r.float().requires_grad_(True)
            s_ref = xr + rr if use_res else xr
            s_ref_q = s_ref.to(dtype).float() if use_res else s_ref
            y_ref = F.layer_norm(s_ref_q, (D,), gm, bt, 1e-12)
            y, s, mean, rstd = K.layernorm_fwd(x, gm, bt, 1e-12, res=r if use_res else None)
            res.append(check(f"ln fwd[{dtype}] {rows}x{D} res={use_res}", y, y_ref, dtype, atol=1e-4 if dtype == torch.float32 else 5e-2))
            sin = (s if use_res else x)
            # backward against autograd on the same (rounded) LN input
            sq = sin.float().detach().requires_grad_(True)
            gmr, btr = gm.clone().requires_grad_(True), bt.clone().requires_grad_(True)
            F.layer_norm(sq, (D,), gmr, btr, 1e-12).backward(dy.float())
            extra = rnd(rows, D, seed=seed + 5, dtype=dtype) if use_res else None
            ds, dh = K.layernorm_bwd(dy, sin, mean, rstd, gm, ds_extra=extra, want_dh=use_res)
            ds_ref = sq.grad + (extra.float() if use_res else 0)
            res.append(check(f"ln bwd[{dtype}] {rows}x{D} res={use_res}", ds, ds_ref, dtype, atol=1e-4 if dtype == torch.float32 else 5e-2))
            if use_res:
                res.append(check(f"ln bwd dh[{dtype}]", dh, ds_ref, dtype, atol=1e-4 if dtype == torch.float32 else 5e-2))
            dbeta, dgamma = K.colreduce(1, dy, sin, mean, rstd, want_dot=True)
            res.append(check(f"ln dgamma[{dtype}] {rows}x{D}", dgamma, gmr.grad, dtype, atol=2e-4 * math.sqrt(rows) if dtype == torch.float32 else 0.5))
            res.append(check(f"ln dbeta[{dtype}] {rows}x{D}", dbeta, btr.grad, dtype, atol=2e-4 * math.sqrt(rows) if dtype == torch.float32 else 0.5))
    return res


@case
@both_dtypes
def batchnorm(dtype):
    res = []
    rows, C = 3 * 50, 40
    x = rnd(rows, C, seed=1, dtype=dtype) * 2 + 0.5
    gm, bt = 1 + 0.1 * rnd(C, seed=2), 0.1 * rnd(C, seed=3)
    dz = rnd(rows, C, seed=4, dtype=dtype)
    mean, _ = K.colreduce(0, x, scale=1.0 / rows)
    var, _ = K.colreduce(3, None, x=x, mean=mean, scale=1.0 / rows, rows=rows, D=C)
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    nb = torch.zeros((), dtype=torch.int64, device=DEV)
    rstd = K.bn_finalize(mean, var, rows, 1e-5, 0.1, rm, rv, nb)
    y, _ = K.bn_apply(x, mean, rstd, gm, bt, act="tanh")
    xr = x.float().requires_grad_(True)
    gmr, btr = gm.clone().requires_grad_(True), bt.clone().requires_grad_(True)
    rm2, rv2 = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    yr = torch.tanh(F.batch_norm(xr, rm2, rv2, gmr, btr, True, 0.1, 1e-5))
    res.append(check(f"bn fwd[{dtype}]", y, yr, dtype, atol=1e-4 if dtype == torch.float32 else 5e-2))
    res.append(check(f"bn running_mean[{dtype}]", rm, rm2, torch.float32, atol=1e-5))
    res.append(check(f"bn running_var[{dtype}]", rv, rv2, torch.float32, atol=1e-4))
    # one-pass moments (colreduce mode 6 + var_is_ex2; the bf16 training path) == the two-pass statistics
    mean1, ex2 = K.colreduce(6, None, x=x, scale=1.0 / rows, rows=rows, D=C, want_dot=True)
    rm3, rv3 = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    rstd1 = K.bn_finalize(mean1, ex2, rows, 1e-5, 0.1, rm3, rv3, None, var_is_ex2=True)
    res.append(check(f"bn one-pass mean[{dtype}]", mean1, mean, torch.float32, atol=1e-5))
    res.append(check(f"bn one-pass rstd[{dtype}]", rstd1, rstd, torch.float32, rtol=1e-4, atol=1e-5))
    res.append(check(f"bn one-pass running_var[{dtype}]", rv3, rv2, torch.float32, atol=1e-4))
    res.append(check(f"bn num_batches[{dtype}]", nb.float().view(1), torch.ones(1), torch.float32))
    yr.backward(dz.float())
    dyp = K.act_dropout_bwd(dz, y, act="tanh")
    sdy, sdyx = K.colreduce(2, dyp, x, mean, rstd, want_dot=True)
    dx = K.bn_bwd(dyp, x, mean, rstd, gm, sdy, sdyx)
    a = 2e-4 if dtype == torch.float32 else 8e-2
    res.append(check(f"bn bwd dx[{dtype}]", dx, xr.grad, dtype, atol=a))
    res.append(check(f"bn dgamma[{dtype}]", sdyx, gmr.grad, dtype, atol=a * 10))
    res.append(check(f"bn dbeta[{dtype}]", sdy, btr.grad, dtype, atol=a * 10))
    return res


@case
def batchnorm_act_dropout_vectorised():
    """The 16-byte BatchNorm + activation + dropout kernels of csrc/convmod.hip (bf16 training path of Fn.batch_norm_act: statistics,
    apply, and a backward pass that recomputes the activation / dropout derivative) against torch fp32 (no dropout) and against the
    scalar kernels of norm.hip with the same seeds => same dropout masks."""
    import os
    from seq2seq_vc_amd.ops import functional as Fn
    res = []
    bf = torch.bfloat16

    def rel(a, b):
        a, b = a.detach().float().cpu().reshape(-1), b.detach().float().cpu().reshape(-1)
        return float((a - b).norm() / b.norm().clamp_min(1e-30))
    for (rows, C, act, seed) in [(150, 40, "tanh", 1), (12800, 512, "tanh", 2), (4096, 80, None, 3), (333, 1536, "swish", 4), (64, 8, "relu", 5)]:
        x = rnd(rows, C, seed=seed, dtype=bf) * 2 + 0.5
        gm, bt = 1 + 0.1 * rnd(C, seed=seed + 1), 0.1 * rnd(C, seed=seed + 2)
        dz = rnd(rows, C, seed=seed + 3, dtype=bf)
        xr = x.float().requires_grad_(True)
        gmr, btr = gm.clone().requires_grad_(True), bt.clone().requires_grad_(True)
        rm2, rv2 = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
        pre = F.batch_norm(xr, rm2, rv2, gmr, btr, True, 0.1, 1e-5)
        yr = {"tanh": torch.tanh, "swish": lambda t_: t_ * torch.sigmoid(t_), "relu": torch.relu, None: lambda t_: t_}[act](pre)
        yr.backward(dz.float())

        def run(vec, p):
            os.environ["S2SVC_NO_BN_VEC"] = "0" if vec else "1"
            K.manual_seed(77)
            K.reset_op_counter()
            xx = x.clone().requires_grad_(True)
            g2, b2 = gm.clone().requires_grad_(True), bt.clone().requires_grad_(True)
            rm, rv, nb = torch.zeros(C, device=DEV), torch.ones(C, device=DEV), torch.zeros((), dtype=torch.int64, device=DEV)
            y = Fn.batch_norm_act(xx, g2, b2, rm, rv, nb, True, act, p, 1e-5, 0.1)
            y.backward(dz)
            return y.detach(), xx.grad, g2.grad, b2.grad, rm, rv, nb
        try:
            v0, s0, v1, s1 = run(True, 0.0), run(False, 0.0), run(True, 0.5), run(False, 0.5)
        finally:
            os.environ.pop("S2SVC_NO_BN_VEC", None)
        tag = f"bn-vec {rows}x{C} {act}"
        for nm, got, sca, ref in zip(("y", "dx", "dgamma", "dbeta"), v0, s0, (yr, xr.grad, gmr.grad, btr.grad)):
            e, es = rel(got, ref), rel(sca, ref)
            res.append((e <= 2e-2 and e <= 1.5 * es + 1e-3, f"{tag} {nm}: rel-L2 vs fp32 torch {e:.2e} (scalar kernels {es:.2e})"))
        res.append(check(f"{tag} running_mean", v0[4], rm2, torch.float32, atol=2e-3, rtol=1e-2))
        res.append(check(f"{tag} running_var", v0[5], rv2, torch.float32, atol=2e-3, rtol=1e-2))
        res.append((int(v0[6]) == 1, f"{tag} num_batches_tracked = {int(v0[6])}"))
        for nm, got, sca in zip(("y", "dx", "dgamma", "dbeta"), v1, s1):
            e = rel(got, sca)
            res.append((e <= 1e-2, f"{tag} dropout 0.5, same masks, {nm}: rel-L2 vectorised vs scalar {e:.2e}"))
        res.append((bool(((v1[0] == 0) == (s1[0] == 0)).all()), f"{tag} dropped positions agree"))
    return res


def _rel_shift_new(x):
    b, h, t, l = x.shape
    zp = torch.zeros((b, h, t, 1), device=x.device, dtype=x.dtype)
    xp = torch.cat([zp, x], dim=-1).view(b, h, l + 1, t)
    return xp[:, :, 1:].view_as(x)[:, :, :, : l // 2 + 1]


def _rel_shift_legacy(x):
    b, h, t, l = x.shape
    zp = torch.zeros((b, h, t, 1), device=x.device, dtype=x.dtype)
    xp = torch.cat([zp, x], dim=-1).view(b, h, l + 1, t)
    return xp[:, :, 1:].view_as(x)


@case
@both_dtypes
def attn_softmax(dtype):
    res = []
    B, H, T1, T2 = 3, 2, 21, 29
    sc = rnd(B, H, T1, T2, seed=1) * 3
    klen = torch.tensor([29, 17, 5], dtype=torch.int32, device=DEV)
    scale = 1 / math.sqrt(24)
    for causal in (False, True):
        T2c = T1 if causal else T2
        s = sc[..., :T2c].contiguous()
        kl = torch.clamp(klen, max=T2c)
        sr = s.clone().requires_grad_(True)
        mask = (torch.arange(T2c, device=DEV)[None, None, None, :] < kl[:, None, None, None])
        if causal:
            mask = mask & torch.tril(torch.ones(T1, T2c, dtype=torch.bool, device=DEV))[None, None]
        z = (sr * scale).masked_fill(~mask, torch.finfo(torch.float32).min)
        pr = torch.softmax(z, -1).masked_fill(~mask, 0.0)
        attn, pdrop = K.attn_softmax_fwd(s, dtype, scale, klen=kl, causal=causal)
        res.append(check(f"softmax fwd[{dtype}] causal={causal}", attn, pr, dtype, atol=1e-6 if dtype == torch.float32 else 1e-2))
        dp = rnd(B, H, T1, T2c, seed=5)
        pr.backward(dp)
        dsc, _ = K.attn_softmax_bwd(attn, dp, scale)
        res.append(check(f"softmax bwd[{dtype}] causal={causal}", dsc, sr.grad, dtype, atol=1e-6 if dtype == torch.float32 else 2e-2))
        # rows padded to a multiple of 8 (ld > T2): pad columns hold garbage on input and must come back zero
        ld = (T2c + 7) // 8 * 8
        sp = torch.full((B, H, T1, ld), 1e30, dtype=torch.float32, device=DEV)
        sp[..., :T2c] = s
        attn_p, _ = K.attn_softmax_fwd(sp, dtype, scale, klen=kl, causal=causal, T2=T2c)
        res.append(check(f"softmax fwd padded[{dtype}] causal={causal}", attn_p[..., :T2c], pr, dtype,
                         atol=1e-6 if dtype == torch.float32 else 1e-2))
        res.append(check(f"softmax fwd pad cols zero[{dtype}]", attn_p[..., T2c:].float(), torch.zeros_like(attn_p[..., T2c:]).float(),
                         torch.float32, atol=0.0))
        dpp = torch.full((B, H, T1, ld), 1e30, dtype=torch.float32, device=DEV)
        dpp[..., :T2c] = dp
        dsp, _ = K.attn_softmax_bwd(attn_p, dpp, scale, T2=T2c)
        res.append(check(f"softmax bwd padded[{dtype}] causal={causal}", dsp[..., :T2c], sr.grad, dtype,
                         atol=1e-6 if dtype == torch.float32 else 2e-2))
        res.append(check(f"softmax bwd pad cols zero[{dtype}]", dsp[..., T2c:].float(), torch.zeros_like(dsp[..., T2c:]).float(),
                         torch.float32, atol=0.0))
    # relative-position variants (T1 == T2)
    T = 19
    ac = rnd(B, H, T, T, seed=7) * 2
    kl = torch.tensor([19, 11, 3], dtype=torch.int32, device=DEV)
    mask = (torch.arange(T, device=DEV)[None, None, None, :] < kl[:, None, None, None]).expand(B, H, T, T)
    for mode, Lp, shift in ((1, 2 * T - 1, _rel_shift_new), (2, T, _rel_shift_legacy)):
        bd = rnd(B, H, T, Lp, seed=8 + mode) * 2
        acr, bdr = ac.clone().requires_grad_(True), bd.clone().requires_grad_(True)
        z = ((acr + shift(bdr)) * scale).masked_fill(~mask, torch.finfo(torch.float32).min)
        pr = torch.softmax(z, -1).masked_fill(~mask, 0.0)
        attn, _ = K.attn_softmax_fwd(ac, dtype, scale, klen=kl, bd=bd, rel_mode=mode)
        res.append(check(f"softmax relpos{mode} fwd[{dtype}]", attn, pr, dtype, atol=1e-6 if dtype == torch.float32 else 1e-2))
        dp = rnd(B, H, T, T, seed=20 + mode)
        pr.backward(dp)
        dsc, dbd = K.attn_softmax_bwd(attn, dp, scale, Lp=Lp, rel_mode=mode)
        res.append(check(f"softmax relpos{mode} dac[{dtype}]", dsc, acr.grad, dtype, atol=1e-6 if dtype == torch.float32 else 2e-2))
        res.append(check(f"softmax relpos{mode} dbd[{dtype}]", dbd, bdr.grad, dtype, atol=1e-6 if dtype == torch.float32 else 2e-2))
        # bd / dbd rows padded to a multiple of 8 (ldb > Lp): garbage in the bd pad, zeros expected in the dbd pad
        ldb = (Lp + 7) // 8 * 8 + 8
        bdp = torch.full((B, H, T, ldb), 1e30, dtype=torch.float32, device=DEV)
        bdp[..., :Lp] = bd
        attn_p, _ = K.attn_softmax_fwd(ac, dtype, scale, klen=kl, bd=bdp, rel_mode=mode, Lp=Lp)
        res.append(check(f"softmax relpos{mode} fwd padded bd[{dtype}]", attn_p, attn, dtype, atol=0.0, rtol=0.0))
        dsc_p, dbd_p = K.attn_softmax_bwd(attn, dp, scale, Lp=Lp, rel_mode=mode, ldb=ldb)
        res.append(check(f"softmax relpos{mode} dbd padded[{dtype}]", dbd_p[..., :Lp], dbd, dtype, atol=0.0, rtol=0.0))
        res.append(check(f"softmax relpos{mode} dbd pad zero[{dtype}]", dbd_p[..., Lp:].float(),
                         torch.zeros_like(dbd_p[..., Lp:]).float(), torch.float32, atol=0.0))
    return res


@case
@both_dtypes
def elementwise(dtype):
    res = []
    x = rnd(7, 33, 48, seed=1, dtype=dtype)
    for act, f in (("relu", torch.relu), ("tanh", torch.tanh), ("swish", lambda t: t * torch.sigmoid(t)),
                   ("gelu", F.gelu), ("sigmoid", torch.sigmoid)):
        xr = x.float().clone().requires_grad_(True)
        yr = f(xr)
        y = K.act_dropout_fwd(x, act=act)
        res.append(check(f"act {act} fwd[{dtype}]", y, yr, dtype, atol=1e-6 if dtype == torch.float32 else 2e-2))
        dz = rnd(7, 33, 48, seed=2, dtype=dtype)
        yr.backward(dz.float())
        saved = x if act in ("swish", "gelu") else y
        dx = K.act_dropout_bwd(dz, saved, act=act)
        res.append(check(f"act {act} bwd[{dtype}]", dx, xr.grad, dtype, atol=1e-5 if dtype == torch.float32 else 5e-2))
    # dropout statistics + fwd/bwd mask consistency
    big = torch.ones(1 << 20, dtype=dtype, device=DEV)
    seed = K.new_seed(big.device)
    for p in (0.1, 0.5):
        y = K.act_dropout_fwd(big, p=p, seed=seed)
        keep = (y != 0).float().mean().item()
        ok = abs(keep - (1 - p)) < 5e-3
        res.append((ok, f"dropout keep-rate[{dtype}] p={p}: {keep:.4f}"))
        val = y[y != 0].float().mean().item()
        res.append((abs(val - 1 / (1 - p)) < 2e-2, f"dropout scale[{dtype}] p={p}: {val:.4f}"))
        dxm = K.act_dropout_bwd(big, y, p=p, seed=seed)
        res.append(check(f"dropout fwd/bwd same mask[{dtype}] p={p}", dxm, y, dtype))
    # the 16-byte kernels (n % 8 == 0) and the scalar ones (any n) draw the same masks and compute the same values
    xs = rnd(4099, seed=7, dtype=dtype)
    for act in ("relu", "swish"):
        y_s = K.act_dropout_fwd(xs, act=act, p=0.3, seed=seed)            # n = 4099: scalar kernel
        y_v = K.act_dropout_fwd(xs[:4096].clone(), act=act, p=0.3, seed=seed)
        res.append((bool(torch.equal(y_s[:4096], y_v)), f"act+dropout fwd[{dtype}] {act}: vector == scalar kernel"))
        sv_s = xs if act == "swish" else y_s
        d_s = K.act_dropout_bwd(xs, sv_s, act=act, p=0.3, seed=seed)
        d_v = K.act_dropout_bwd(xs[:4096].clone(), sv_s[:4096].clone(), act=act, p=0.3, seed=seed)
        res.append((bool(torch.equal(d_s[:4096], d_v)), f"act+dropout bwd[{dtype}] {act}: vector == scalar kernel"))
    # positional encodings
    B, T, D = 3, 17, 32
    xx = rnd(B, T, D, seed=3, dtype=dtype)
    pe = rnd(40, D, seed=4)
    alpha = torch.tensor(0.7, device=DEV)
    y = K.posenc_fwd(xx, 1.0, alpha, pe)
    res.append(check(f"scaled posenc fwd[{dtype}]", y, xx.float() + 0.7 * pe[:T], dtype))
    y = K.posenc_fwd(xx, math.sqrt(D), None, pe)
    res.append(check(f"abs posenc fwd[{dtype}]", y, xx.float() * math.sqrt(D) + pe[:T], dtype, atol=1e-5 if dtype == torch.float32 else 0.2))
    dy = rnd(B, T, D, seed=5, dtype=dtype)
    dx, dalpha = K.posenc_bwd(dy, 1.0, pe, want_dalpha=True)
    res.append(check(f"posenc bwd dx[{dtype}]", dx, dy.float(), dtype))
    res.append(check(f"posenc bwd dalpha[{dtype}]", dalpha.view(1), (dy.float() * pe[:T]).sum().view(1), dtype, atol=1e-3 if dtype == torch.float32 else 0.5))
    # glu
    xg = rnd(50, 2 * 24, seed=6, dtype=dtype)
    xr = xg.float().requires_grad_(True)
    yr = F.glu(xr, dim=-1)
    y = K.glu_fwd(xg)
    res.append(check(f"glu fwd[{dtype}]", y, yr, dtype, atol=1e-6 if dtype == torch.float32 else 2e-2))
    dyg = rnd(50, 24, seed=7, dtype=dtype)
    yr.backward(dyg.float())
    res.append(check(f"glu bwd[{dtype}]", K.glu_bwd(xg, dyg), xr.grad, dtype, atol=1e-6 if dtype == torch.float32 else 3e-2))
    # head bias, axpby, cast
    q = rnd(10, 4 * 8, seed=8, dtype=dtype)
    u, v = rnd(32, seed=9), rnd(32, seed=10)
    qu, qv = K.add_head_bias(q, u, v)
    res.append(check(f"add_head_bias u[{dtype}]", qu, q.float() + u, dtype))
    res.append(check(f"add_head_bias v[{dtype}]", qv, q.float() + v, dtype))
    res.append(check(f"axpby[{dtype}]", K.axpby(0.5, q, 2.0, q), 2.5 * q.float(), dtype))
    f32 = rnd(1000, seed=11)
    res.append(check("cast f32->bf16", K.cast(f32, torch.bfloat16), f32.to(torch.bfloat16), torch.float32, atol=0, rtol=0))
    return res


@case
def colreduce_grouped():
    """Queued column reductions (parameter gradients of LayerNorm / bias / BatchNorm vectors) run as grouped launches ==
    the same reductions launched one by one: mixed dtypes, modes, shapes, more items than one launch holds (24), and two
    items accumulating into the same gradient (serialised)."""
    res = []
    items = []
    for i in range(30):
        dtype = torch.bfloat16 if i % 3 else torch.float32
        rows, D = [(2016, 384), (100, 1536), (37, 80), (4096, 384), (9, 50)][i % 5]
        mode = (0, 1, 2)[i % 3]
        dy, x = rnd(rows, D, seed=i, dtype=dtype), rnd(rows, D, seed=100 + i, dtype=dtype)
        mean = rnd(rows if mode == 1 else D, seed=200 + i)
        rstd = rnd(rows if mode == 1 else D, seed=300 + i).abs() + 0.5
        s0, d0 = rnd(D, seed=400 + i), rnd(D, seed=500 + i)
        items.append((mode, dy, x, mean, rstd, s0, d0))
    ref = []
    for mode, dy, x, mean, rstd, s0, d0 in items:
        a, b = s0.clone(), d0.clone()
        K.colreduce(mode, dy, x if mode else None, mean if mode else None, rstd if mode else None, out_sum=a,
                    out_dot=b if mode else None, accumulate=True)
        ref.append((a, b))
    outs = [(s0.clone(), d0.clone()) for _, _, _, _, _, s0, d0 in items]
    queue = []
    with K.record_colreduce(queue):
        for (mode, dy, x, mean, rstd, _, _), (a, b) in zip(items, outs):
            K.colreduce(mode, dy, x if mode else None, mean if mode else None, rstd if mode else None, out_sum=a,
                        out_dot=b if mode else None, accumulate=True)
        mode, dy, x, mean, rstd, _, _ = items[1]          # once more into the same outputs: a shared parameter
        K.colreduce(mode, dy, x, mean, rstd, out_sum=outs[1][0], out_dot=outs[1][1], accumulate=True)
        now, _ = K.colreduce(0, items[0][1])              # no output slot: not queued, result available immediately
    res.append((len(queue) == 31, f"{len(queue)} reductions queued"))
    res.append(check("unqueued reduction", now, items[0][1].float().sum(0), torch.float32, rtol=1e-4, atol=1e-2))
    res.append((bool(torch.equal(outs[0][0], items[0][5])), "nothing is written before the flush"))
    K.flush_colreduce(queue)
    for i, ((a, b), (ra, rb)) in enumerate(zip(outs, ref)):
        if i == 1:
            ra, rb = 2 * ra - items[1][5], 2 * rb - items[1][6]
        tol = dict(rtol=1e-5, atol=1e-3 if i == 1 else 1e-5)
        res.append(check(f"grouped colreduce #{i} sum", a, ra, torch.float32, **tol))
        if items[i][0]:
            res.append(check(f"grouped colreduce #{i} dot", b, rb, torch.float32, **tol))
    # the 16-byte stage-1 variant (bf16 rows of >= 768 columns: a lane owns 8 columns) against fp32 torch, all modes
    for j, (rows, D) in enumerate([(4096, 1536), (300, 3072), (65, 776)]):
        dy, x = rnd(rows, D, seed=700 + j, dtype=torch.bfloat16), rnd(rows, D, seed=710 + j, dtype=torch.bfloat16)
        fy, fx = dy.float(), x.float()
        for mode in (0, 1, 2, 3, 4, 6):
            per_row = mode == 1
            mean = rnd(rows if per_row else D, seed=720 + j)
            rstd = rnd(rows if per_row else D, seed=730 + j).abs() + 0.5
            s_, d_ = K.colreduce(mode, None if mode in (3, 6) else dy, None if mode == 0 else x,
                                 mean if mode in (1, 2, 3) else None, rstd if mode in (1, 2) else None, want_dot=mode in (1, 2, 6))
            mb, rb_ = (mean[:, None], rstd[:, None]) if per_row else (mean[None, :], rstd[None, :])
            want = {0: (fy.sum(0), None), 1: (fy.sum(0), (fy * (fx - mb) * rb_).sum(0)), 2: (fy.sum(0), (fy * (fx - mb) * rb_).sum(0)),
                    3: (((fx - mb) ** 2).sum(0), None), 4: ((fy * fx).sum(0), None), 6: (fx.sum(0), (fx * fx).sum(0))}[mode]
            sc = float(want[0].abs().max()) + 1.0
            res.append(check(f"colreduce 16-byte {rows}x{D} mode {mode} sum", s_, want[0], torch.float32, rtol=1e-4, atol=1e-4 * sc))
            if want[1] is not None:
                sc = float(want[1].abs().max()) + 1.0
                res.append(check(f"colreduce 16-byte {rows}x{D} mode {mode} dot", d_, want[1], torch.float32, rtol=1e-4, atol=1e-4 * sc))
    return res


@case
def gather3_grouped_refresh():
    """s2svc_gather3_grouped (the one-launch refresh of every permuted convolution weight after an optimiser step) == one
    s2svc_gather3 per copy, bit for bit: forward and data-gradient layouts of Conv1d weights (the latter takes the LDS-tile path:
    innermost output index with the largest source stride), Conv2d taps, ragged extents, both output dtypes, > 24 jobs."""
    res = []
    jobs = []
    shapes = [(1536, 512, 3), (512, 80, 5), (80, 512, 5), (384, 384, 3), (100, 37, 3), (65, 130, 1), (33, 31, 7)]
    for i, (O, I, ks) in enumerate(shapes * 4):
        w = rnd(O, I, ks, seed=900 + i)
        odt = torch.bfloat16 if i % 2 else torch.float32
        jobs.append((w, ((O, ks, I), (I * ks, 1, ks), 0, odt)))                       # (O, k, I): forward
        jobs.append((w, ((I, ks, O), (ks, -1, I * ks), ks - 1, odt)))                 # (I, k flipped, O): data gradient
    w2 = rnd(96, 64, 3, 3, seed=990)
    jobs.append((w2, ((96, 9, 64), (64 * 9, 1, 9), 0, torch.bfloat16)))
    registry = [(w, key, torch.full(key[0], float("nan"), dtype=key[3], device=DEV)) for w, key in jobs]
    K.gather3_refresh(registry)
    bad = []
    for (w, (n, st, off, odt), buf) in registry:
        ref = K.gather3(w, n, st, off, odt)
        if not torch.equal(buf, ref):
            bad.append((tuple(w.shape), n, st))
    res.append((not bad, f"grouped refresh of {len(registry)} permuted copies == one gather each: {len(bad)} differ {bad[:3]}"))
    # and the gather itself against torch indexing for one of each kind
    w = jobs[0][0]
    O, I, ks = w.shape
    res.append(check("gather3 (O, k, I)", registry[0][2], w.permute(0, 2, 1).contiguous(), registry[0][2].dtype, atol=1e-2))
    res.append(check("gather3 (I, k flipped, O)", registry[1][2], w.flip(2).permute(1, 2, 0).contiguous(), registry[1][2].dtype, atol=1e-2))
    return res


@case
def transposed_weight_copies():
    """s2svc_transpose_tiles: every matrix of a table transposed in one launch (64x64 tiles; 16-byte path and the
    element-wise path for extents / offsets that are not multiples of 8), bit-exact."""
    res = []
    shapes = [(384, 384), (1536, 384), (384, 1536), (80, 384), (384, 7296), (320, 384), (5, 7), (129, 66), (64, 72), (1, 16)]
    offs, mats, n = [], [], 0
    for i, (r, c) in enumerate(shapes):
        if i == 7:
            n += 3                                   # an offset that is not a multiple of 8
        offs.append(n)
        mats.append(rnd(r, c, seed=i, dtype=torch.bfloat16))
        n += r * c
        n = (n + 63) // 64 * 64 if i != 6 else n
    src = torch.zeros(n, dtype=torch.bfloat16, device=DEV)
    dst = torch.full((n,), -1.0, dtype=torch.bfloat16, device=DEV)
    tiles = []
    for (r, c), o, m in zip(shapes, offs, mats):
        src[o:o + r * c] = m.reshape(-1)
        nt = ((r + 63) // 64) * ((c + 63) // 64)
        tiles += [(o, o, (r << 32) | c, t) for t in range(nt)]
    K.transpose_tiles(torch.tensor(tiles, dtype=torch.int64, device=DEV), src, dst)
    for (r, c), o, m in zip(shapes, offs, mats):
        got = dst[o:o + r * c].view(c, r)
        res.append((bool(torch.equal(got, m.t().contiguous())), f"transpose {r}x{c} at offset {o}"))
    return res


@case
def dropout_mask_statistics():
    """The counter-based dropout masks (csrc/common.h): keep rate within 4 sigma for several p, masks of neighbouring
    seeds (consecutive op offsets, consecutive steps) and of neighbouring elements uncorrelated, same seed => same mask."""
    res = []
    n = 1 << 22
    ones = torch.ones(n, dtype=torch.float32, device=DEV)
    K.manual_seed(1234)

    def mask(p, seed):
        return (K.act_dropout_fwd(ones, None, p, seed) > 0).float()
    for p in (0.1, 0.5, 0.9, 0.05):
        sd = K.new_seed(ones.device)
        m = mask(p, sd)
        keep = m.mean().item()
        sigma = math.sqrt(p * (1 - p) / n)
        res.append((abs(keep - (1 - p)) < 4 * sigma + 2.0 ** -16, f"keep rate p={p}: {keep:.5f} (4 sigma = {4 * sigma:.5f})"))
        res.append((torch.equal(m, mask(p, sd)), f"p={p}: same (seed, index) -> same mask"))
        c = m - m.mean()
        var = (c * c).mean().item()
        for lag in (1, 2, 3, 4, 5, 8, 64, 1536):
            r = (c[:-lag] * c[lag:]).mean().item() / var
            res.append((abs(r) < 5 / math.sqrt(n), f"p={p}: lag-{lag} autocorrelation {r:+.2e}"))
        for d in (1, 2, 0x10001):                           # next op in the step / the same op one step later
            m2 = mask(p, (sd[0], sd[1] + d))
            c2 = m2 - m2.mean()
            r = (c * c2).mean().item() / var
            res.append((abs(r) < 5 / math.sqrt(n), f"p={p}: correlation with seed+{d:#x} {r:+.2e}"))
    return res


@case
def mas_kernel():
    from oracle import mas as omas
    res = []
    rng = np.random.default_rng(0)
    for (B, Tf, Tx, seed) in [(2, 6, 3, 7), (4, 40, 12, 1), (16, 256, 64, 2), (3, 70, 100, 3), (2, 300, 130, 4)]:
        gen = torch.Generator().manual_seed(seed)
        lp = torch.log_softmax(torch.randn(B, Tf, Tx, generator=gen), dim=-1)
        tl = torch.randint(max(1, Tx // 2), Tx + 1, (B,), generator=gen)
        fl = torch.randint(max(1, Tf // 2), Tf + 1, (B,), generator=gen)
        tl[0], fl[0] = Tx, Tf
        if (B, Tf, Tx) == (2, 6, 3):
            tl, fl = torch.tensor([3, 2]), torch.tensor([6, 4])
        ds_ref, bl_ref, paths, margin = omas.viterbi_decode(lp.numpy(), tl.numpy(), fl.numpy())
        ds, path, binmean = K.mas(lp.to(DEV), tl.to(DEV).int(), fl.to(DEV).int())
        ok = torch.equal(ds.cpu(), torch.from_numpy(ds_ref))
        res.append((ok, f"mas ds B{B} Tf{Tf} Tx{Tx} bit-exact={ok} (min margin {margin:.2e})"))
        okp = all(np.array_equal(path[b, : int(fl[b])].cpu().numpy(), paths[b]) for b in range(B))
        res.append((okp, f"mas path B{B} Tf{Tf} Tx{Tx} bit-exact={okp}"))
        bl = -(binmean.sum() / B).item()
        res.append((abs(bl - bl_ref) <= 1e-5 * max(1, abs(bl_ref)), f"mas bin_loss {bl:.6f} vs {bl_ref:.6f}"))
    # KAT2: all ties
    lp = torch.full((1, 6, 3), math.log(1 / 3))
    ds, path, _ = K.mas(lp.to(DEV), torch.tensor([3], dtype=torch.int32, device=DEV), torch.tensor([6], dtype=torch.int32, device=DEV))
    ok = path[0].cpu().tolist() == [0, 0, 0, 0, 1, 2]
    res.append((ok, f"mas KAT2 all-ties path={path[0].cpu().tolist()}"))
    # KAT3: T_inp > T_mel
    lp3 = torch.log_softmax(torch.from_numpy(np.random.default_rng(0).standard_normal((3, 5)).astype(np.float32)), -1)[None]
    ds, path, _ = K.mas(lp3.to(DEV), torch.tensor([5], dtype=torch.int32, device=DEV), torch.tensor([3], dtype=torch.int32, device=DEV))
    ok = path[0].cpu().tolist() == [2, 3, 4] and ds[0].cpu().tolist() == [0, 0, 1, 1, 1]
    res.append((ok, f"mas KAT3 path={path[0].cpu().tolist()} ds={ds[0].cpu().tolist()}"))
    return res



@case
def gemm_fp32_small_tiles():
    """fp32 GEMMs that underfill the chip on 64 x 64 tiles run on 32 x 32 tiles with K tiles of 128 (gemm_fast.hip; the duration
    predictor's Linear layers, data gradients and its 29-column spline projection): against torch fp32 and BIT FOR BIT against the
    64 x 64 kernel (tile hint 64) -- k ascends in the same order in both; K-contiguous and row-contiguous operands, bias / ReLU /
    residual / accumulate, split-K partials, row / column / K tails."""
    res = []
    f32 = torch.float32
    prev = torch.backends.cuda.matmul.allow_tf32
    torch.backends.cuda.matmul.allow_tf32 = False
    try:
        for (M, N, Kd, seed) in [(1024, 384, 384, 1), (1024, 29, 384, 2), (1000, 200, 136, 3), (70, 8, 640, 4), (1024, 192, 768, 5)]:
            a, w = rnd(M, Kd, seed=seed), rnd(N, Kd, seed=seed + 50, scale=0.05)
            bias, r = rnd(N, seed=seed + 60), rnd(M, N, seed=seed + 70)
            ref = torch.relu(a.double() @ w.double().t() + bias.double()) + r.double()
            outs = []
            for tile in (64, 0):
                c = torch.full((M, N), float("nan"), device=DEV)
                K.gemm(K.operand(a, Kd), K.operand(w, Kd), M, N, Kd, c, in_dtype=f32, bias=bias, act="relu", res=r, tile=tile)
                outs.append(c)
            res.append(check(f"fp32 32x32 tiles {M}x{N}x{Kd} bias + relu + residual", outs[1], ref.float(), f32, rtol=1e-4, atol=1e-4))
            res.append((bool(torch.equal(outs[0], outs[1])), f"fp32 32x32 tiles {M}x{N}x{Kd} == 64x64 kernel bit for bit"))
            # data gradient: B row-contiguous, accumulate into C
            if N % 4 == 0:
                dy = rnd(M, N, seed=seed + 80)
                outs = []
                for tile in (64, 0):
                    dx = rnd(M, Kd, seed=seed + 90)
                    K.gemm(K.operand(dy, N), K.operand(w, Kd, layout=K.RC), M, Kd, N, dx, in_dtype=f32, accumulate=True, tile=tile)
                    outs.append(dx)
                ref = rnd(M, Kd, seed=seed + 90).double() + dy.double() @ w.double()
                res.append(check(f"fp32 32x32 tiles dgrad {M}x{Kd}x{N} (RC weights, accumulate)", outs[1], ref.float(), f32, rtol=1e-4, atol=1e-4))
                res.append((bool(torch.equal(outs[0], outs[1])), f"fp32 32x32 tiles dgrad {M}x{Kd}x{N} == 64x64 kernel bit for bit"))
                # weight gradient: both operands row-contiguous, split-K 2 (partials + reduction kernel)
                if Kd % 4 == 0 and M % 4 == 0:
                    outs = []
                    for tile in (64, 0):
                        dw = torch.full((N, Kd), float("nan"), device=DEV)
                        K.gemm(K.operand(dy, N, layout=K.RC), K.operand(a, Kd, layout=K.RC), N, Kd, M, dw, in_dtype=f32, splitk=2, tile=tile)
                        outs.append(dw)
                    res.append(check(f"fp32 32x32 tiles wgrad {N}x{Kd}x{M} split-K 2", outs[1], (dy.double().t() @ a.double()).float(), f32, rtol=1e-4, atol=1e-4))
                    res.append((bool(torch.equal(outs[0], outs[1])), f"fp32 32x32 tiles wgrad {N}x{Kd}x{M} == 64x64 kernel bit for bit"))
                    # the tile hint of ops.kernels.plan_gemm (32): unsplit and split, with the fused bias gradient (row sums of dY^T)
                    for sk in (1, 2):
                        dw = torch.full((N, Kd), float("nan"), device=DEV)
                        db = torch.full((N,), float("nan"), device=DEV)
                        K.gemm(K.operand(dy, N, layout=K.RC), K.operand(a, Kd, layout=K.RC), N, Kd, M, dw, in_dtype=f32, splitk=sk, tile=32, a_rowsum=db)
                        res.append(check(f"fp32 tile hint 32 wgrad {N}x{Kd}x{M} split-K {sk}", dw, (dy.double().t() @ a.double()).float(), f32, rtol=1e-4, atol=1e-4))
                        res.append(check(f"fp32 tile hint 32 wgrad {N}x{Kd}x{M} split-K {sk}: fused bias gradient", db, dy.double().sum(0).float(), f32, rtol=1e-4, atol=1e-4))
    finally:
        torch.backends.cuda.matmul.allow_tf32 = prev
    return res


@case
def gemm_8phase_n96():
    """The one-round geometry of the 8-wave kernel (gemm_8ph_kernel_n96: 256 x 96 p tiles, p = 2 / 3 phases per K tile, 4 x 2
    waves, B units with pad rows): against torch fp32 matmul of the same bf16 inputs and BIT FOR BIT against the 256 x 128 kernel
    (same K order per output element); odd / even / minimal numbers of K tiles (two LDS stages alternate), row and
    column tails, the common epilogue's options (bias, relu, residual, dropout), repeated launches, and what the policy picks."""
    res = []
    dtype = torch.bfloat16
    L = K._lib.lib()
    prev = L.s2svc_gemm_set_8ph(-1)
    OFF, P = 1 | (3 << 4) | (1 << 8), {2: 1 | (5 << 8), 3: 1 | (6 << 8)}
    try:
        for (M, N, Kd, seed) in [(4096, 4608, 1536, 1), (4096, 3072, 1536, 2), (4096, 1536, 384, 4), (1030, 288, 640, 5),
                                 (2050, 192, 128, 6), (300, 384, 192, 7), (4096, 576, 320, 8), (515, 864, 128, 9), (256, 288, 4608, 10)]:
            a, b = rnd(M, Kd, seed=seed, dtype=dtype), rnd(N, Kd, seed=seed + 100, dtype=dtype, scale=0.05)
            bias = rnd(N, seed=seed + 200)
            ref = a.float() @ b.float().t() + bias
            L.s2svc_gemm_set_8ph(OFF)
            c0 = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
            K.gemm(K.operand(a, Kd), K.operand(b, Kd), M, N, Kd, c0, in_dtype=dtype, bias=bias)
            for ph in (2, 3):
                if N % (96 * ph):
                    continue
                L.s2svc_gemm_set_8ph(P[ph])
                c = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
                K.gemm(K.operand(a, Kd), K.operand(b, Kd), M, N, Kd, c, in_dtype=dtype, bias=bias)
                res.append(check(f"8-phase n96 {M}x{N}x{Kd} tile 256x{96 * ph}", c, ref, dtype))
                res.append((bool(torch.equal(c, c0)), f"8-phase n96 {M}x{N}x{Kd} 256x{96 * ph} == 256x128 kernel bit for bit"))
        # epilogue options of the common epilogue
        M, N, Kd = 4096, 4608, 256
        a, b = rnd(M, Kd, seed=11, dtype=dtype), rnd(N, Kd, seed=12, dtype=dtype, scale=0.05)
        r, bias = rnd(M, N, seed=13, dtype=dtype), rnd(N, seed=14)
        for ph in (2, 3):
            outs = []
            for mode in (OFF, P[ph]):
                L.s2svc_gemm_set_8ph(mode)
                K.reset_op_counter()
                c = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
                K.gemm(K.operand(a, Kd), K.operand(b, Kd), M, N, Kd, c, in_dtype=dtype, bias=bias, act="relu", res=r, drop_p=0.1)
                outs.append(c)
            res.append((bool(torch.equal(outs[0], outs[1])) and not bool(torch.isnan(outs[1].float()).any()),
                        f"8-phase n96 256x{96 * ph}: bias + relu + dropout + residual == 256x128 kernel bit for bit"))
        # repeated launches give identical bits
        for ph, (M, N, Kd) in ((2, (4096, 3072, 1536)), (3, (4096, 4608, 1536))):
            L.s2svc_gemm_set_8ph(P[ph])
            a, b = rnd(M, Kd, seed=31, dtype=dtype), rnd(N, Kd, seed=32, dtype=dtype, scale=0.05)
            first, bad = None, 0
            for it in range(30):
                c = torch.empty(M, N, dtype=dtype, device=DEV)
                K.gemm(K.operand(a, Kd), K.operand(b, Kd), M, N, Kd, c, in_dtype=dtype)
                if first is None:
                    first = c
                elif not torch.equal(first, c):
                    bad += 1
            res.append((bad == 0, f"8-phase n96 256x{96 * ph} {M}x{N}x{Kd}: 30 launches identical ({bad} differ)"))
    finally:
        L.s2svc_gemm_set_8ph(prev)
    return res


@case
def gemm_8phase():
    """The 256-row / 8-wave / phase-interleaved bf16 GEMM (csrc/gemm_8ph.hip) in each of its three tile geometries
    (256 x 256, 512 x 128, 256 x 128) and both wave-half schedules (skewed / lockstep), on shapes with odd and even numbers
    of K tiles (two LDS buffers alternate), a single K-tile pair, row and column tails (clamped source rows, masked
    stores), the implicit Conv2d-3x3-s2 operand with its division-free (tap, channel) walk -- against torch fp32 matmul /
    conv2d of the same bf16 inputs and against the 128 x 128 kernel; epilogue variants; bit-reproducibility over repeated
    launches (no race between the DMA stream and the fragment reads)."""
    res = []
    dtype = torch.bfloat16
    L = K._lib.lib()
    prev = L.s2svc_gemm_set_8ph(-1)
    GEO = {1: "256x256", 2: "512x128", 3: "256x128"}
    try:
        for (M, N, Kd, seed) in [(4096, 1536, 1536, 1), (8200, 520, 448, 2), (4096, 4096, 512, 3), (7300, 2048, 320, 4),
                                 (2048, 3072, 128, 5), (33000, 128, 192, 6), (1030, 264, 640, 7)]:
            a, b = rnd(M, Kd, seed=seed, dtype=dtype), rnd(N, Kd, seed=seed + 100, dtype=dtype, scale=0.05)
            bias = rnd(N, seed=seed + 200)
            ref = a.float() @ b.float().t() + bias
            L.s2svc_gemm_set_8ph(0)
            c0 = torch.empty((M, N), dtype=dtype, device=DEV)
            K.gemm(K.operand(a, Kd), K.operand(b, Kd), M, N, Kd, c0, in_dtype=dtype, bias=bias)
            for geo in (1, 2, 3):
                outs = {}
                for mode in (1, 2):
                    L.s2svc_gemm_set_8ph(mode | (geo << 4))
                    c = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
                    K.gemm(K.operand(a, Kd), K.operand(b, Kd), M, N, Kd, c, in_dtype=dtype, bias=bias)
                    outs[mode] = c
                res.append(check(f"8-phase dense {M}x{N}x{Kd} tile {GEO[geo]}", outs[1], ref, dtype))
                res.append((bool(torch.equal(outs[1], outs[2])), f"8-phase {M}x{N}x{Kd} {GEO[geo]}: skewed and lockstep wave halves agree bit for bit"))
                same = float((outs[1].float() - c0.float()).abs().max())
                res.append((same <= 0.02 * float(ref.abs().max()), f"8-phase {GEO[geo]} vs 128x128 kernel {M}x{N}x{Kd}: max diff {same:.3e}"))
        # the policy takes the kernel for these shapes (a NaN-filled output would survive a silent fall-through to ... nothing)
        L.s2svc_gemm_set_8ph(1)
        # epilogue variants: relu + residual, accumulate into fp32 C
        M, N, Kd = 4096, 1536, 384
        a, b = rnd(M, Kd, seed=11, dtype=dtype), rnd(N, Kd, seed=12, dtype=dtype, scale=0.05)
        r = rnd(M, N, seed=13, dtype=dtype)
        for geo in (1, 2, 3):
            L.s2svc_gemm_set_8ph(1 | (geo << 4))
            c = torch.empty(M, N, dtype=dtype, device=DEV)
            K.gemm(K.operand(a, Kd), K.operand(b, Kd), M, N, Kd, c, in_dtype=dtype, act="relu", res=r)
            res.append(check(f"8-phase {GEO[geo]} relu + residual", c, torch.relu(a.float() @ b.float().t()) + r.float(), dtype))
            c32 = rnd(M, N, seed=14)
            c0 = c32.clone()
            K.gemm(K.operand(a, Kd), K.operand(b, Kd), M, N, Kd, c32, in_dtype=dtype, accumulate=True)
            res.append(check(f"8-phase {GEO[geo]} accumulate into fp32 C", c32, c0 + a.float() @ b.float().t(), torch.float32, rtol=2e-2, atol=2e-2))
        # implicit Conv2d 3x3 stride 2 (the VTN / TTS front-end's second convolution; row tails; C = 64 / 128)
        for (B, T1, F1, C, O, seed) in [(32, 127, 39, 64, 384, 21), (7, 127, 39, 128, 384, 22), (9, 61, 39, 64, 256, 23)]:
            T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
            x = rnd(B, T1, F1, C, seed=seed, dtype=dtype)
            w = rnd(O, C, 3, 3, seed=seed + 10, scale=0.05)
            bb = rnd(O, seed=seed + 20)
            wp = K.gather3(w, (O, 9, C), (C * 9, 1, 9), 0, dtype)
            yr = torch.relu(F.conv2d(x.float().permute(0, 3, 1, 2), w.to(dtype).float(), bb, stride=2)).permute(0, 2, 3, 1)
            for geo in (1, 2, 3):
                for mode in (1, 2):
                    L.s2svc_gemm_set_8ph(mode | (geo << 4))
                    y = torch.full((B, T2, F2, O), float("nan"), dtype=dtype, device=DEV)
                    K.gemm(K.operand(x, C, mode=K.CONV2D_S2, C=C, T1=T1, F1=F1, T2=T2, F2=F2), K.operand(wp, 9 * C), B * T2 * F2, O, 9 * C, y,
                           in_dtype=dtype, bias=bb, act="relu")
                    res.append(check(f"8-phase conv2d fwd B{B} {T1}x{F1} C{C} O{O} tile {GEO[geo]} mode {mode}", y, yr, dtype))
        # repeated launches give identical bits
        for geo, (M, N, Kd) in ((1, (4096, 4096, 1024)), (2, (8192, 384, 3456)), (3, (4096, 1536, 1536))):
            L.s2svc_gemm_set_8ph(1 | (geo << 4))
            a, b = rnd(M, Kd, seed=31, dtype=dtype), rnd(N, Kd, seed=32, dtype=dtype, scale=0.05)
            first, bad = None, 0
            for it in range(30):
                c = torch.empty(M, N, dtype=dtype, device=DEV)
                K.gemm(K.operand(a, Kd), K.operand(b, Kd), M, N, Kd, c, in_dtype=dtype)
                if first is None:
                    first = c
                elif not torch.equal(first, c):
                    bad += 1
            res.append((bad == 0, f"8-phase {GEO[geo]}: {bad} of 29 repeated launches differ from the first"))
    finally:
        L.s2svc_gemm_set_8ph(prev)
    return res


@case
def gemm_8phase_swish_epilogues():
    """The Swish epilogue pair of the 8-wave 256 x 128 kernel (epilogue_flush_swish: forward = bias, pre-activation output, Swish,
    dropout; data gradient = dropout mask x swish'(pre-activation)) at the Conformer feed-forward shapes of AAS-VC (4096 x 1536 x 1536
    decoder, 4096 x 1536 x 384 encoder): values against fp32 torch, the dropout mask of the backward launch equal to the forward's
    (same seed), kept elements scaled by 1 / (1 - p)."""
    res = []
    dtype = torch.bfloat16
    for (M, N, Kd, seed) in ((4096, 1536, 1536, 1), (4096, 1536, 384, 2)):
        a, w = rnd(M, Kd, seed=seed, dtype=dtype), rnd(N, Kd, seed=seed + 10, dtype=dtype, scale=0.05)
        b = rnd(N, seed=seed + 20, scale=0.1)
        z = a.float() @ w.float().t() + b
        sd = K.new_seed(a.device)
        outs = {}
        for p_drop in (0.0, 0.25):
            y = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
            pre = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
            K.gemm(K.operand(a, Kd), K.operand(w, Kd), M, N, Kd, y, in_dtype=dtype, bias=b, act="swish", drop_p=p_drop, seed=sd, pre_out=pre)
            dy = rnd(M, Kd, seed=seed + 30, dtype=dtype)
            du = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
            K.gemm(K.operand(dy, Kd), K.operand(w, Kd), M, N, Kd, du, in_dtype=dtype, emask=pre, emask_mode=1, drop_p=p_drop, seed=sd)
            outs[p_drop] = (y, pre, du)
        y0, pre0, du0 = outs[0.0]
        res.append(check(f"8-phase swish fwd {M}x{N}x{Kd}: pre-activation", pre0, z, dtype))
        res.append(check(f"8-phase swish fwd {M}x{N}x{Kd}: output", y0, z * torch.sigmoid(z), dtype))
        u = pre0.float()
        sg = torch.sigmoid(u)
        g = dy.float() @ w.float().t()
        res.append(check(f"8-phase swish' data gradient {M}x{N}x{Kd}", du0, g * sg * (1 + u * (1 - sg)), dtype))
        y1, pre1, du1 = outs[0.25]
        res.append((bool(torch.equal(pre1, pre0)), f"8-phase swish {M}x{N}x{Kd}: the pre-activation does not depend on the dropout"))
        keep = y1 != 0
        frac = 1.0 - float(keep.float().mean())
        res.append((abs(frac - 0.25) < 0.01 + float((y0 == 0).float().mean()), f"8-phase swish {M}x{N}x{Kd}: dropped fraction {frac:.4f} (p = 0.25)"))
        res.append(check(f"8-phase swish {M}x{N}x{Kd}: kept outputs scaled by 1 / (1 - p)", y1, torch.where(keep, y0.float() / 0.75, torch.zeros_like(z)), dtype))
        both = (du0 != 0) & (y0 != 0)
        same_mask = float(((du1 != 0) == keep)[both].float().mean())
        res.append((same_mask == 1.0, f"8-phase swish {M}x{N}x{Kd}: backward mask == forward mask on {same_mask:.6f} of the elements"))
        res.append(check(f"8-phase swish' {M}x{N}x{Kd}: kept gradients scaled by 1 / (1 - p)", du1, torch.where(du1 != 0, du0.float() / 0.75, torch.zeros_like(z)), dtype))
    return res


@case
def gemm_8phase_weight_gradients():
    """The 8-wave kernel for row-contiguous operands (gemm_8ph_tr_kernel: C (+)= dY^T . X on 256 x 128 tiles, transpose reads,
    bias row-sums as MFMA products with a ones fragment): single launches (odd / even K-tile counts, a single K-tile pair,
    fp32 C with and without accumulation, bf16 C), the grouped launch, both wave-half schedules bit for bit, repeated
    launches bit for bit -- against torch fp32 and against the 4-wave kernels (set_8ph(0))."""
    res = []
    dtype = torch.bfloat16
    L = K._lib.lib()
    prev = L.s2svc_gemm_set_8ph(-1)
    try:
        for (rows, fin, fout, seed) in [(4096, 1536, 4608, 1), (320, 2048, 2048, 2), (128, 1536, 3072, 3), (1088, 4096, 1024, 4)]:
            x, dy = rnd(rows, fin, seed=seed, dtype=dtype), rnd(rows, fout, seed=seed + 50, dtype=dtype)
            dw0, db0 = rnd(fout, fin, seed=seed + 100), rnd(fout, seed=seed + 150)
            ref_w = dy.float().t() @ x.float()
            ref_b = dy.float().sum(0)
            outs = {}
            for mode in (0, 1 | (1 << 4), 2 | (1 << 4), 1 | (3 << 4), 2 | (3 << 4)):      # off; 256x256 / 256x128 tiles, skewed / lockstep
                L.s2svc_gemm_set_8ph(mode)
                dw, db = dw0.clone(), db0.clone()
                K.gemm(K.operand(dy, fout, layout=K.RC), K.operand(x, fin, layout=K.RC), fout, fin, rows, dw, in_dtype=dtype,
                       accumulate=True, a_rowsum=db, a_rowsum_accumulate=True)
                outs[mode] = (dw, db)
            sc = float(ref_w.abs().max())
            for geo, nm in ((1, "256x256"), (3, "256x128")):
                o1, o2 = outs[1 | (geo << 4)], outs[2 | (geo << 4)]
                res.append(check(f"8-wave wgrad {fout}x{fin}x{rows} tile {nm} dW (+=)", o1[0], dw0 + ref_w, torch.float32, rtol=1e-3, atol=2e-4 * sc))
                res.append(check(f"8-wave wgrad {fout}x{fin}x{rows} tile {nm} db (+=)", o1[1], db0 + ref_b, torch.float32, rtol=1e-3,
                                 atol=2e-4 * float(ref_b.abs().max())))
                res.append((bool(torch.equal(o1[0], o2[0]) and torch.equal(o1[1], o2[1])),
                            f"8-wave wgrad {fout}x{fin}x{rows} {nm}: skewed and lockstep wave halves agree bit for bit"))
                dmax = float((o1[0] - outs[0][0]).abs().max())
                res.append((dmax <= 1e-3 * sc, f"8-wave {nm} vs 4-wave wgrad {fout}x{fin}x{rows}: max diff {dmax:.3e} (scale {sc:.1f})"))
            L.s2svc_gemm_set_8ph(1)
            dwn = torch.full((fout, fin), float("nan"), device=DEV)
            dbn = torch.full((fout,), float("nan"), device=DEV)
            K.gemm(K.operand(dy, fout, layout=K.RC), K.operand(x, fin, layout=K.RC), fout, fin, rows, dwn, in_dtype=dtype, a_rowsum=dbn)
            res.append(check(f"8-wave wgrad {fout}x{fin}x{rows} dW (=)", dwn, ref_w, torch.float32, rtol=1e-3, atol=2e-4 * sc))
            res.append(check(f"8-wave wgrad {fout}x{fin}x{rows} db (=)", dbn, ref_b, torch.float32, rtol=1e-3, atol=2e-4 * float(ref_b.abs().max())))
            dwb = torch.empty((fout, fin), dtype=dtype, device=DEV)
            K.gemm(K.operand(dy, fout, layout=K.RC), K.operand(x, fin, layout=K.RC), fout, fin, rows, dwb, in_dtype=dtype)
            res.append(check(f"8-wave wgrad {fout}x{fin}x{rows} bf16 C", dwb, ref_w, dtype, atol=0.02 * sc))
        # grouped: five AAS-VC decoder layers' worth of shapes in one launch (the queue policy of training)
        L.s2svc_gemm_set_8ph(1)
        shapes = [(2048, 1536, 1536), (2048, 1536, 3072), (2048, 3072, 1536), (2048, 1536, 1536), (2048, 1536, 256)]
        probs = []
        for i, (rows, fin, fout) in enumerate(shapes):
            probs.append((rnd(rows, fin, seed=60 + i, dtype=dtype), rnd(rows, fout, seed=70 + i, dtype=dtype), rnd(fout, fin, seed=80 + i),
                          rnd(fout, seed=90 + i)))
        saved = (K._GROUP_TILE, K._GROUP_MAX_TILES, K._GROUP_BIG_TILES)
        K._GROUP_MAX_TILES = 1 << 30
        results = {}
        for mode in (0, 1, 1 | (3 << 4)):          # 4-wave; policy (256x256: 396 tiles... the cost model decides); forced 256x128
            L.s2svc_gemm_set_8ph(mode)
            outs = [(p[2].clone(), p[3].clone()) for p in probs]
            queue = []
            with K.record_grouped(queue):
                for (x, dy, _, _), (dw, db) in zip(probs, outs):
                    K.gemm(K.operand(dy, dy.shape[1], layout=K.RC), K.operand(x, x.shape[1], layout=K.RC), dy.shape[1], x.shape[1], x.shape[0],
                           dw, in_dtype=dtype, accumulate=True, a_rowsum=db, a_rowsum_accumulate=True)
            K.flush_grouped(queue)
            results[mode] = outs
        K._GROUP_TILE, K._GROUP_MAX_TILES, K._GROUP_BIG_TILES = saved
        for mode in (1, 1 | (3 << 4)):
            for i, ((x, dy, dw0, db0), (dw, db)) in enumerate(zip(probs, results[mode])):
                ref_w, ref_b = dw0 + dy.float().t() @ x.float(), db0 + dy.float().sum(0)
                res.append(check(f"8-wave grouped wgrad (mode {mode}) #{i} {tuple(dw.shape)} dW", dw, ref_w, torch.float32, rtol=1e-3,
                                 atol=2e-4 * float(ref_w.abs().max())))
                res.append(check(f"8-wave grouped wgrad (mode {mode}) #{i} db", db, ref_b, torch.float32, rtol=1e-3,
                                 atol=2e-4 * float(ref_b.abs().max())))
        # repeated launches give identical bits
        x, dy = rnd(4096, 1536, seed=41, dtype=dtype), rnd(4096, 4608, seed=42, dtype=dtype)
        first, bad = None, 0
        for it in range(20):
            dw, db = torch.empty(4608, 1536, device=DEV), torch.empty(4608, device=DEV)
            K.gemm(K.operand(dy, 4608, layout=K.RC), K.operand(x, 1536, layout=K.RC), 4608, 1536, 4096, dw, in_dtype=dtype, a_rowsum=db)
            if first is None:
                first = (dw, db)
            elif not (torch.equal(first[0], dw) and torch.equal(first[1], db)):
                bad += 1
        res.append((bad == 0, f"8-wave wgrad: {bad} of 19 repeated launches differ from the first"))
    finally:
        L.s2svc_gemm_set_8ph(prev)
    return res


@case
def gemm_w8_ragged_weight_gradients():
    """The ragged 8-wave weight-gradient kernel (csrc/gemm_8ph.hip "W8": (problem, K chunk, 256 x 128 tile) units, zero-filled DMA
    sources past M / N / K, idle wave halves skip their MFMAs): VTN's layer shapes (K = 2016 = 31.5 K tiles), shapes that are ragged
    in every dimension, a reduction shorter than one K tile, reductions cut into 2 and 16 chunks (partial tiles through the
    workspace + the ordered reduction launch) -- dW and the bias row sums, accumulating and not, against torch fp32 and the 4-wave
    kernels; a group of all problems gives the bits of one launch per problem; repeated launches agree bit for bit."""
    res = []
    dtype = torch.bfloat16
    L = K._lib.lib()
    prev = L.s2svc_gemm_set_w8(-1, 0)
    shapes = [(2016, 384, 384), (2016, 384, 1152), (2016, 1536, 384), (2048, 384, 1536), (2016, 7296, 384), (2048, 384, 320),
              (777, 72, 200), (100, 8, 8), (63, 320, 256), (4096, 384, 1536), (4160, 256, 136), (40000, 128, 264)]
    try:
        L.s2svc_gemm_set_w8(1, 32)
        probs = []
        for i, (rows, fin, fout) in enumerate(shapes):
            x, dy = rnd(rows, fin, seed=300 + i, dtype=dtype), rnd(rows, fout, seed=330 + i, dtype=dtype)
            probs.append((x, dy, rnd(fout, fin, seed=360 + i), rnd(fout, seed=390 + i)))

        def descs_for(outs, accumulate):
            ds = []
            for (x, dy, _, _), (dw, db) in zip(probs, outs):
                K.gemm(K.operand(dy, dy.shape[1], layout=K.RC), K.operand(x, x.shape[1], layout=K.RC), dy.shape[1], x.shape[1], x.shape[0], dw,
                       in_dtype=dtype, accumulate=accumulate, group=ds)
                ds[-1].a_rowsum, ds[-1].a_rowsum_accumulate = db.data_ptr(), 1 if accumulate else 0
            return ds

        takes = [bool(L.s2svc_gemm_wgrad_ok(K.ctypes.addressof(d))) for d in descs_for([(p[2], p[3]) for p in probs], True)]
        res.append((all(takes), f"W8 takes all {len(shapes)} shapes: {takes}"))
        for accumulate in (True, False):
            outs = [(p[2].clone(), p[3].clone()) if accumulate else (torch.full_like(p[2], float("nan")), torch.full_like(p[3], float("nan")))
                    for p in probs]
            K.launch_wgrad_group(descs_for(outs, accumulate))                      # one group
            singles = [(p[2].clone(), p[3].clone()) if accumulate else (torch.full_like(p[2], float("nan")), torch.full_like(p[3], float("nan")))
                       for p in probs]
            for d in descs_for(singles, accumulate):                               # one launch per problem
                K.launch_wgrad_group([d])
            for (rows, fin, fout), (x, dy, dw0, db0), (dw, db), (dw1, db1) in zip(shapes, probs, outs, singles):
                ref_w, ref_b = dy.float().t() @ x.float(), dy.float().sum(0)
                if accumulate:
                    ref_w, ref_b = ref_w + dw0, ref_b + db0
                sc, scb = float(ref_w.abs().max()), float(ref_b.abs().max())
                tag = f"W8 wgrad {fout}x{fin}x{rows} {'(+=)' if accumulate else '(=)'}"
                res.append(check(tag + " dW", dw, ref_w, torch.float32, rtol=1e-3, atol=3e-4 * sc))
                res.append(check(tag + " db", db, ref_b, torch.float32, rtol=1e-3, atol=3e-4 * scb))
                res.append((bool(torch.equal(dw, dw1) and torch.equal(db, db1)), tag + ": the grouped launch gives the bits of the single launch"))
        # against the 4-wave grouped kernel (the path these problems took until round 3) through the training-time queue
        results = {}
        for on in (0, 1):
            L.s2svc_gemm_set_w8(on, 0)
            outs = [(p[2].clone(), p[3].clone()) for p in probs]
            queue = []
            with K.record_grouped(queue):
                for (x, dy, _, _), (dw, db) in zip(probs, outs):
                    K.gemm(K.operand(dy, dy.shape[1], layout=K.RC), K.operand(x, x.shape[1], layout=K.RC), dy.shape[1], x.shape[1], x.shape[0], dw,
                           in_dtype=dtype, accumulate=True, a_rowsum=db, a_rowsum_accumulate=True)
            nq = len(queue)
            K.flush_grouped(queue)
            results[on] = outs
        res.append((nq >= len(shapes) - 2, f"W8: {nq} of {len(shapes)} problems were queued for the grouped launch"))
        worst = 0.0
        for (dw_a, db_a), (dw_b, db_b), p in zip(results[0], results[1], probs):
            sc = float((dw_a - p[2]).abs().max()) + 1e-6
            worst = max(worst, float((dw_a - dw_b).abs().max()) / sc, float((db_a - db_b).abs().max()) / (float((db_a - p[3]).abs().max()) + 1e-6))
        res.append((worst <= 1e-3, f"W8 vs the 4-wave grouped kernel through flush_grouped: largest relative difference {worst:.3e} (<= 1e-3)"))
        L.s2svc_gemm_set_w8(1, 0)
        bad = 0
        for _ in range(10):
            outs = [(p[2].clone(), p[3].clone()) for p in probs]
            K.launch_wgrad_group(descs_for(outs, True))
            bad += sum(0 if (torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])) else 1 for a, b in zip(outs, results[1]))
        res.append((bad == 0, f"W8: {bad} outputs of 10 repeated grouped launches differ from the first"))
        # chunk length 8: every VTN reduction is cut into 4 chunks -- same values within fp32 rounding of a different summation order
        L.s2svc_gemm_set_w8(1, 8)
        outs = [(p[2].clone(), p[3].clone()) for p in probs]
        K.launch_wgrad_group(descs_for(outs, True))
        worst = max(float((a[0] - b[0]).abs().max()) / (float((b[0] - p[2]).abs().max()) + 1e-6) for a, b, p in zip(outs, results[1], probs))
        res.append((worst <= 1e-4, f"W8 with 8-tile chunks vs 32-tile chunks: largest relative difference {worst:.3e} (<= 1e-4)"))
    finally:
        L.s2svc_gemm_set_w8(prev & 1, prev >> 8)
    return res


@case
def bn_two_launch_statistics():
    """s2svc_bn_stats (the second reduction stage folded into the finalisation: 2 launches instead of 3) against the
    three-launch composition it replaces -- colreduce(mode 6) + bn_finalize -- incl. the running statistics, and against
    torch, on the shapes of the VTN postnet (8192 x 512), the AAS-VC convolution module (4096 x 1536), a ragged one and fp32."""
    res = []
    for (rows, C, dtype, seed) in [(8192, 512, torch.bfloat16, 1), (4096, 1536, torch.bfloat16, 2), (1000, 200, torch.bfloat16, 3),
                                   (777, 96, torch.float32, 4)]:
        x = rnd(rows, C, seed=seed, dtype=dtype, scale=2.0) + 0.3
        rm0, rv0 = rnd(C, seed=seed + 30), rnd(C, seed=seed + 40).abs() + 0.5
        rm_a, rv_a, nb_a = rm0.clone(), rv0.clone(), torch.zeros((), dtype=torch.int64, device=DEV)
        rm_b, rv_b, nb_b = rm0.clone(), rv0.clone(), torch.zeros((), dtype=torch.int64, device=DEV)
        mean_a, ex2 = K.colreduce(6, None, x=x, scale=1.0 / rows, rows=rows, D=C, want_dot=True)
        rstd_a = K.bn_finalize(mean_a, ex2, rows, 1e-5, 0.1, rm_a, rv_a, nb_a, var_is_ex2=True)
        mean_b, rstd_b = K.bn_stats(x, rows, C, 1e-5, 0.1, rm_b, rv_b, nb_b)
        for nm, a, b in (("mean", mean_a, mean_b), ("rstd", rstd_a, rstd_b), ("running mean", rm_a, rm_b), ("running var", rv_a, rv_b)):
            res.append(check(f"bn_stats {rows}x{C} {dtype} {nm} vs colreduce + bn_finalize", b, a, torch.float32, rtol=2e-6, atol=1e-7))
        res.append((int(nb_b) == 1, f"bn_stats {rows}x{C}: num_batches_tracked advanced once"))
        xf = x.float()
        res.append(check(f"bn_stats {rows}x{C} mean vs torch", mean_b, xf.mean(0), torch.float32, rtol=1e-4, atol=1e-4))
        res.append(check(f"bn_stats {rows}x{C} rstd vs torch", rstd_b, 1.0 / torch.sqrt(xf.var(0, unbiased=False) + 1e-5), torch.float32, rtol=1e-3, atol=1e-4))
    return res


@case
def decode_emit_advance_kernel():
    """csrc/decode_fused.hip: frames / stop probabilities / stop test of one position from ONE packed feat_out | prob_out projection, the
    step counter and the dropout seed advanced by the last workgroup, the ticket reset, and alpha * pe[pos + 1] left for the next step."""
    from seq2seq_vc_amd.ops import kernels_decode as KD
    res = []
    for dtype in (torch.bfloat16, torch.float32):
        B, r, odim, D = 16, 4, 80, 384
        out = rnd(B, r * odim + r, seed=80, dtype=dtype)
        pos = torch.tensor([6], dtype=torch.int32, device=DEV)
        outs, probs = torch.zeros(B, 40, odim, device=DEV), torch.zeros(B, 40, device=DEV)
        prev, stop_at = torch.zeros(B, odim, dtype=dtype, device=DEV), torch.zeros(B, dtype=torch.int32, device=DEV)
        minlen, maxlen = torch.zeros(B, dtype=torch.int32, device=DEV), torch.full((B,), 7, dtype=torch.int32, device=DEV)
        maxlen[3] = 100
        ticket = torch.zeros(1, dtype=torch.int32, device=DEV)
        pe, alpha = rnd(10, D, seed=81), torch.tensor([0.7], device=DEV)
        pe_next = torch.zeros(1, D, dtype=dtype, device=DEV)
        seedt = K.SEED.tensor(torch.device(DEV))
        s0 = int(seedt.item())
        KD.decode_emit_advance(out, r, odim, 2.0, minlen, maxlen, pos, outs, probs, prev, stop_at, seedt.data_ptr(), 0x10001, ticket,
                               pe=pe, alpha=alpha, pe_next=pe_next)
        ok = (int(pos) == 7 and int(ticket) == 0 and int(seedt.item()) == s0 + 0x10001
              and torch.equal(outs[:, 24:28].reshape(B, -1), out[:, :r * odim].float())
              and torch.equal(prev, out[:, (r - 1) * odim:r * odim]) and torch.allclose(probs[:, 24:28], torch.sigmoid(out[:, r * odim:].float()), atol=1e-6)
              and stop_at.tolist() == [7, 7, 7, 0] + [7] * 12 and torch.equal(pe_next[0], (0.7 * pe[7]).to(dtype)))
        res.append((bool(ok), f"decode_emit_advance[{dtype}]: frames / probs / prev written, pos 6 -> {int(pos)}, seed advanced, ticket reset, "
                              f"stop test {stop_at.tolist()[:5]}, next positional row"))
        # at the last row of the table the positional row is left alone (the host never replays past the capacity)
        pos.fill_(9)
        keep = pe_next.clone()
        KD.decode_emit_advance(out, r, odim, 2.0, minlen, maxlen, pos, outs[:, :40], probs, prev, stop_at, None, 0, ticket, pe=pe, alpha=alpha, pe_next=pe_next)
        res.append((bool(torch.equal(pe_next, keep)) and int(pos) == 10, f"decode_emit_advance[{dtype}]: no read past the positional table"))
    return res


@case
def debug_exec_build_runs_clean():
    """The library built with -DS2SVC_DEBUG_EXEC (the DPP / v_permlane16/32_swap reductions of csrc/common.h trap unless all 64 lanes
    are active) runs the reduction-heavy kernel cases in a child process without trapping and with every check green."""
    import subprocess
    from seq2seq_vc_amd import _lib
    if os.environ.get("S2SVC_LIB"):
        return [(True, "debug-exec build: skipped inside a child that already runs an alternative library")]
    if not os.path.exists(_lib.DEBUG_EXEC_LIB_PATH):
        return [(False, f"{_lib.DEBUG_EXEC_LIB_PATH} is missing: __graft_entry__.build() makes it")]
    only = "layernorm,layernorm_bwd_with_partial_gradients,attn_softmax,attention_fused_vs_reference,batchnorm,batchnorm_act_dropout_vectorised,colreduce_grouped,mas_kernel,bn_two_launch_statistics"
    have = {c.__name__ for c in CASES}
    only = ",".join(n for n in only.split(",") if n in have)
    env = dict(os.environ, S2SVC_LIB=_lib.DEBUG_EXEC_LIB_PATH)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--only", only], env=env, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-600:]
    npass = r.stdout.count("PASS ")
    return [(r.returncode == 0 and npass > 20, f"debug-exec build, cases [{only}]: rc={r.returncode}, {npass} checks passed\n{tail if r.returncode else ''}")]


def main():
    torch.manual_seed(0)
    nfail = 0
    only = None
    if "--only" in sys.argv:
        only = set(sys.argv[sys.argv.index("--only") + 1].split(","))
    for fn in CASES:
        if only is not None and fn.__name__ not in only:
            continue
        try:
            results = fn()
        except Exception:
            results = [(False, f"{fn.__name__}: EXCEPTION\n{traceback.format_exc()}")]
        for ok, msg in results:
            print(("PASS " if ok else "FAIL ") + msg)
            nfail += 0 if ok else 1
        torch.cuda.synchronize()
    print(f"== {nfail} failures")
    return nfail


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
