"""Kernel-level parity for the AAS-VC / Conformer specific kernels (runs on the GPU box)."""
import math
import os
import sys
import traceback

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from gpu_kernel_check import DEV, both_dtypes, check, rnd  # noqa: E402
from seq2seq_vc_amd.ops import functional_aas as FA  # noqa: E402
from seq2seq_vc_amd.ops import kernels as K  # noqa: E402
from seq2seq_vc_amd.ops import kernels_aas as KA  # noqa: E402

CASES = []


def case(fn):
    CASES.append(fn)
    return fn


@case
@both_dtypes
def depthwise_conv(dtype):
    res = []
    for (B, T, C, ks, dil, seed) in [(3, 50, 48, 7, 1, 1), (2, 64, 96, 15, 1, 2), (2, 40, 32, 3, 3, 3), (2, 40, 32, 3, 9, 4), (1, 33, 130, 31, 1, 5),
                                      (2, 300, 384, 15, 1, 6), (1, 5, 64, 3, 1, 7), (2, 131, 72, 3, 3, 8)]:
        x = rnd(B, T, C, seed=seed, dtype=dtype)
        w = rnd(C, 1, ks, seed=seed + 1, scale=0.3)
        b = rnd(C, seed=seed + 2)
        pad = (ks * dil - dil) // 2
        xr = x.float().transpose(1, 2).clone().requires_grad_(True)
        wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        yr = F.conv1d(xr, wr, br, padding=pad, dilation=dil, groups=C)
        y = KA.dwconv(x, w, b, ks, dil)
        a = 2e-5 if dtype == torch.float32 else 5e-2
        res.append(check(f"dwconv fwd[{dtype}] k{ks} d{dil}", y, yr.transpose(1, 2), dtype, atol=a))
        dy = rnd(B, T, C, seed=seed + 3, dtype=dtype)
        yr.backward(dy.float().transpose(1, 2))
        res.append(check(f"dwconv dgrad[{dtype}] k{ks} d{dil}", KA.dwconv(dy, w, None, ks, dil, flip=True), xr.grad.transpose(1, 2), dtype, atol=a))
        res.append(check(f"dwconv wgrad[{dtype}] k{ks} d{dil}", KA.dwconv_wgrad(x, dy, ks, dil), wr.grad, dtype,
                         atol=2e-4 if dtype == torch.float32 else 0.3))
    return res


@case
@both_dtypes
def pairwise_distance_logsoftmax(dtype):
    res = []
    B, Tf, Tx, A = 3, 37, 21, 96
    f = rnd(B, Tf, A, seed=1, dtype=dtype)
    t = rnd(B, Tx, A, seed=2, dtype=dtype)
    tl = torch.tensor([21, 13, 5], dtype=torch.int32, device=DEV)
    fr, tr = f.float().clone().requires_grad_(True), t.float().clone().requires_grad_(True)
    dist = torch.norm(fr.unsqueeze(2) - tr.unsqueeze(1), p=2, dim=3)
    mask = torch.arange(Tx, device=DEV)[None, None, :] >= tl[:, None, None]
    ref = F.log_softmax((-dist).masked_fill(mask, -np.inf), dim=-1)
    logp = FA.pairwise_logsoftmax(f.clone().requires_grad_(True), t.clone().requires_grad_(True), tl)
    res.append(check(f"pairwise logp[{dtype}]", logp, ref, torch.float32, atol=2e-5 if dtype == torch.float32 else 1e-4))
    g = rnd(B, Tf, Tx, seed=3)
    g = g.masked_fill(mask, 0.0)
    (ref.masked_fill(mask, 0.0) * g).sum().backward()
    fi, ti = f.clone().requires_grad_(True), t.clone().requires_grad_(True)
    lp = FA.pairwise_logsoftmax(fi, ti, tl)
    (lp.masked_fill(mask, 0.0) * g).sum().backward()
    a = 5e-5 if dtype == torch.float32 else 5e-2
    res.append(check(f"pairwise dfeats[{dtype}]", fi.grad, fr.grad, dtype, atol=a))
    res.append(check(f"pairwise dtext[{dtype}]", ti.grad, tr.grad, dtype, atol=a * 3))
    return res


@case
@both_dtypes
def gaussian_upsampling(dtype):
    res = []
    B, Tx, A, Tf = 3, 12, 40, 50
    hs = rnd(B, Tx, A, seed=1, dtype=dtype)
    tl = torch.tensor([12, 9, 4], dtype=torch.int32, device=DEV)
    fl = torch.tensor([50, 41, 22], dtype=torch.int32, device=DEV)
    ds = torch.zeros(B, Tx, device=DEV)
    g = torch.Generator().manual_seed(5)
    for b in range(B):
        n = int(tl[b])
        d = torch.randint(1, 6, (n,), generator=g).float()
        ds[b, :n] = d.to(DEV)
    hr = hs.float().clone().requires_grad_(True)
    hm = torch.arange(Tf, device=DEV)[None, :] < fl[:, None]
    dm = torch.arange(Tx, device=DEV)[None, :] < tl[:, None]
    tt = torch.arange(0, Tf, device=DEV).unsqueeze(0).repeat(B, 1).float() * hm.float()
    c = ds.cumsum(dim=-1) - ds / 2
    energy = -0.1 * (tt.unsqueeze(-1) - c.unsqueeze(1)) ** 2
    energy = energy.masked_fill(~(dm.unsqueeze(1).repeat(1, Tf, 1)), -float("inf"))
    ref = torch.matmul(torch.softmax(energy, dim=2), hr)
    hi = hs.clone().requires_grad_(True)
    out = FA.gaussian_upsample(hi, ds, tl, fl, Tf, 0.1)
    a = 2e-5 if dtype == torch.float32 else 5e-2
    res.append(check(f"gauss upsample fwd[{dtype}]", out, ref, dtype, atol=a))
    dy = rnd(B, Tf, A, seed=2, dtype=dtype)
    ref.backward(dy.float())
    out.backward(dy)
    res.append(check(f"gauss upsample dhs[{dtype}]", hi.grad, hr.grad, dtype, atol=a * 4))
    return res


@case
def forward_sum_ctc():
    from scipy.stats import betabinom
    res = []
    for (B, Tf, Tx, seed) in [(3, 30, 9, 1), (4, 64, 16, 2), (2, 12, 12, 3), (2, 10, 14, 4), (16, 256, 64, 5)]:
        gen = torch.Generator().manual_seed(seed)
        lp = torch.log_softmax(torch.randn(B, Tf, Tx, generator=gen), dim=-1)
        tl = torch.randint(max(1, Tx // 2), Tx + 1, (B,), generator=gen)
        fl = torch.randint(max(1, Tf // 2), Tf + 1, (B,), generator=gen)
        tl[0], fl[0] = Tx, Tf
        if (B, Tf, Tx) == (2, 10, 14):
            tl, fl = torch.tensor([14, 3]), torch.tensor([10, 8])   # utterance 0 is infeasible (N > T): zero_infinity
        # CPU reference: exactly the reference's forward (losses/forward_sum_loss.py:58-76) with scipy's prior
        prior = torch.full((B, Tf, Tx), -np.inf)
        for b in range(B):
            T, N = int(fl[b]), int(tl[b])
            a = np.arange(1, T + 1, dtype=float)
            bb = np.array([T - t + 1 for t in a])
            prior[b, :T, :N] = torch.from_numpy(betabinom.logpmf(np.arange(N)[..., None], N, a, bb)).transpose(0, 1)
        lpr = lp.clone().requires_grad_(True)
        x = F.pad(lpr + prior, (1, 0, 0, 0, 0, 0), value=float(np.log(np.e ** -1)))
        loss = 0
        for b in range(B):
            tgt = torch.arange(1, int(tl[b]) + 1).unsqueeze(0)
            cur = x[b, : int(fl[b]), : int(tl[b]) + 1].unsqueeze(1)
            loss = loss + F.ctc_loss(cur, tgt, input_lengths=fl[b:b + 1], target_lengths=tl[b:b + 1], zero_infinity=True)
        loss = loss / B
        loss.backward()
        tli, fli = tl.to(DEV).int(), fl.to(DEV).int()
        pr = KA.betabinom_prior(B, Tf, Tx, tli, fli, DEV)
        res.append(check(f"betabinom prior B{B} T{Tf} N{Tx}", pr, prior, torch.float32, atol=2e-5))
        lpi = lp.to(DEV).requires_grad_(True)
        out = FA.forward_sum_loss(lpi, pr, tli, fli, math.e ** -1)
        (out * 1.0).backward()
        res.append(check(f"forward-sum loss B{B} T{Tf} N{Tx}", out.view(1), loss.detach().view(1), torch.float32, atol=2e-5, rtol=2e-5))
        res.append(check(f"forward-sum grad B{B} T{Tf} N{Tx}", lpi.grad, lpr.grad, torch.float32, atol=2e-6, rtol=1e-3))
    return res


@case
@both_dtypes
def interpolate_nearest(dtype):
    res = []
    for (B, Tin, Tout, C) in [(2, 15, 16, 8), (3, 63, 64, 32), (2, 40, 13, 5), (1, 7, 30, 3)]:
        x = rnd(B, Tin, C, seed=Tin, dtype=dtype)
        xr = x.float().clone().requires_grad_(True)
        ref = torch.stack([F.interpolate(xr[i][None].permute(0, 2, 1), size=Tout).permute(0, 2, 1)[0] for i in range(B)])
        res.append(check(f"interp fwd[{dtype}] {Tin}->{Tout}", K.interp_nearest(x, Tout), ref, dtype, atol=0, rtol=0))
        dy = rnd(B, Tout, C, seed=Tout, dtype=dtype)
        ref.backward(dy.float())
        res.append(check(f"interp bwd[{dtype}] {Tin}->{Tout}", K.interp_nearest_bwd(dy, Tin), xr.grad, dtype,
                         atol=1e-6 if dtype == torch.float32 else 5e-2))
    return res


@case
def stft_logmel_frontend():
    from oracle import logmel as OL
    from seq2seq_vc_amd.frontend import logmelfilterbank
    res = []
    rng = np.random.default_rng(3)
    for n in (4000, 16000, 48000 + 137, 255):
        t = np.arange(n) / 16000.0
        x = (0.3 * np.sin(2 * np.pi * 220 * t) + 0.2 * np.sin(2 * np.pi * 3100 * t) + 0.05 * rng.standard_normal(n)).astype(np.float32)
        ref = OL.logmelfilterbank(x, 16000, fft_size=1024, hop_size=256, num_mels=80, fmin=80, fmax=7600)
        got = logmelfilterbank(torch.from_numpy(x).to(DEV), 16000, fft_size=1024, hop_size=256, num_mels=80, fmin=80, fmax=7600)
        res.append(check(f"logmel N={n} frames={ref.shape[0]}", got, torch.from_numpy(ref), torch.float32, atol=2e-4, rtol=1e-4))
    mean, scale = rng.standard_normal(80).astype(np.float32), (1 + rng.random(80)).astype(np.float32)
    got = logmelfilterbank(torch.from_numpy(x).to(DEV), 16000, fmin=80, fmax=7600, mean=mean, scale=scale)
    res.append(check("logmel + fused normalisation", got, torch.from_numpy((ref - mean) / scale), torch.float32, atol=3e-4, rtol=1e-4))
    return res


def main():
    nfail = 0
    for fn in CASES:
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
