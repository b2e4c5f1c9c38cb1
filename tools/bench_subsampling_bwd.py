#!/usr/bin/env python3
"""Stand-alone timings of the backward kernels of the Conv2d front-end (subsampling.py:58-70) at VTN vc1's shapes
(B = 32, T = 256, 80 mel bins, C = 384): the tail of the backward pass, where every kernel fills the chip and the step
pays the SUM of their times.  Each line: `--iters` launches replayed from one hipGraph, HIP events, median of `--rounds`.

    python tools/bench_subsampling_bwd.py [--iters 20] [--rounds 3]
"""
import argparse
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from seq2seq_vc_amd.ops import kernels as K  # noqa: E402
from tools.gemm_bench import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    L = K._lib.lib()
    dt = torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(1)
    u = lambda *s: (torch.rand(*s, device="cuda", generator=g) * 2 - 1).to(dt)
    B, T, Fm, C, D = a.batch, 256, 80, 384, 384
    T1, F1 = (T - 3) // 2 + 1, (Fm - 3) // 2 + 1
    T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
    M2, M = B * T2 * F2, B * T2
    Kd = C * F2

    def t(fn):
        return statistics.median(bench(fn, a.iters) for _ in range(a.rounds))

    def line(name, us, gf=None, mb=None):
        s = f"{name:62s} {us:8.1f} us"
        if gf:
            s += f"  {gf / us * 1e3:7.1f} TF/s = {gf / us * 1e3 / 2500 * 100:5.1f} % of the bf16 peak"
        if mb:
            s += f"  {mb / us:6.2f} TB/s = {mb / us / 8 * 100:5.1f} % of 8 TB/s"
        print(s, flush=True)

    # ---- Linear(19 C -> D) behind the convolutions
    x = torch.relu(u(M, Kd))
    w = (torch.rand(D, Kd, device="cuda", generator=g) * 0.04 - 0.02)
    dy = u(M, D)
    wp = K.gather3(w, (D, F2, C), (C * F2, 1, F2), 0, dt)
    wd = K.gather3(w, (F2, C, D), (1, F2, C * F2), 0, dt)
    dx = torch.empty(M, Kd, dtype=dt, device="cuda")
    gf = 2.0 * M * Kd * D * 1e-9
    line("embed Linear dgrad, row-contiguous weight (4-wave kernel)",
         t(lambda: K.gemm(K.operand(dy, D), K.operand(wp, Kd, layout=K.RC), M, Kd, D, dx, in_dtype=dt, emask=x)), gf)
    line("embed Linear dgrad, transposed permuted copy (8-wave kernel)",
         t(lambda: K.gemm(K.operand(dy, D), K.operand(wd, D), M, Kd, D, dx, in_dtype=dt, emask=x)), gf)
    dwp = torch.empty(D, Kd, dtype=torch.float32, device="cuda")
    db = torch.zeros(D, device="cuda")
    tile, sk = K.plan_gemm(D, Kd, M)
    for w8 in (False, True):
        line(f"embed Linear wgrad, w8={int(w8)} (plan tile {tile} split {sk})",
             t(lambda: K.gemm(K.operand(dy, D, layout=K.RC), K.operand(x, Kd, layout=K.RC), D, Kd, M, dwp, in_dtype=dt, splitk=sk, tile=tile,
                              a_rowsum=db, a_rowsum_accumulate=True, wgrad=w8)), gf)

    # ---- Conv2d(C -> C, 3, 2)
    xin = torch.relu(u(B, T1, F1, C))
    dy2 = u(B, T2, F2, C)
    wc = (torch.rand(C, C, 3, 3, device="cuda", generator=g) * 0.04 - 0.02)
    gf = 2.0 * M2 * C * 9 * C * 1e-9
    dwc = torch.empty(C, 9 * C, dtype=torch.float32, device="cuda")
    tile, sk = K.plan_gemm(C, 9 * C, M2)
    for w8 in (False, True):
        line(f"conv2 wgrad, w8={int(w8)} (plan tile {tile} split {sk})",
             t(lambda: K.gemm(K.operand(dy2, C, layout=K.RC), K.operand(xin, C, layout=K.RC, mode=K.CONV2D_S2, C=C, T1=T1, F1=F1, T2=T2, F2=F2),
                              C, 9 * C, M2, dwc, in_dtype=dt, splitk=sk, tile=tile, a_rowsum=db, a_rowsum_accumulate=True, wgrad=w8)), gf)
    wts = K.tconv2d_weights(wc)
    dxin = torch.empty(B, T1, F1, C, dtype=dt, device="cuda")

    def dgrad(mask=True, only=None, tile=0):
        for cls, wt in enumerate(wts):
            if only is not None and cls != only:
                continue
            pt, pf = cls >> 1, cls & 1
            Tc, Fc = (T1 - pt + 1) // 2, (F1 - pf + 1) // 2
            Kc = wt.shape[1]
            K.gemm(K.operand(dy2, C, mode=K.TCONV2D_S2, C=C, T1=Tc, F1=Fc, T2=T2, F2=F2, pad=cls), K.operand(wt, Kc), B * Tc * Fc, C, Kc,
                   dxin, in_dtype=dt, c_map=(T1, F1, Tc, Fc, pt, pf), emask=xin if mask else None, tile=tile)

    for geo, name in ((0, "by policy"), (1, "256 x 256"), (2, "512 x 128"), (3, "256 x 128")):
        prev = L.s2svc_gemm_set_8ph(1 | (geo << 4))
        line(f"conv2 dgrad (4 parity classes, relu' mask), tiles {name}", t(dgrad), gf)
        if geo in (0, 3):
            line(f"conv2 dgrad (4 parity classes, relu' mask, GENERAL flush), tiles {name}", t(lambda: dgrad(True, None, 65)), gf)
            line(f"conv2 dgrad (4 parity classes, no mask), tiles {name}", t(lambda: dgrad(False)), gf)
            for c in range(4):
                line(f"   class {c} alone ({wts[c].shape[1] // 64} K tiles), tiles {name}", t(lambda: dgrad(True, c)))
        L.s2svc_gemm_set_8ph(prev)
    # forward, for scale
    wpf = K.gather3(wc, (C, 9, C), (C * 9, 1, 9), 0, dt)
    y2 = torch.empty(B, T2, F2, C, dtype=dt, device="cuda")
    bias = torch.zeros(C, device="cuda")
    line("conv2 forward", t(lambda: K.gemm(K.operand(xin, C, mode=K.CONV2D_S2, C=C, T1=T1, F1=F1, T2=T2, F2=F2), K.operand(wpf, 9 * C), M2, C,
                                           9 * C, y2, in_dtype=dt, bias=bias, act="relu")), gf)

    # ---- Conv2d(1 -> C, 3, 2): weight gradient (the last kernel of the backward pass)
    x0 = u(B, T, Fm)
    dy1 = u(B, T1, F1, C)
    dw1, db1 = torch.zeros(C, 1, 3, 3, device="cuda"), torch.zeros(C, device="cuda")
    mb = dy1.numel() * 2 * 1e-6
    w1, b1 = torch.rand(C, 1, 3, 3, device="cuda", generator=g) * 0.6 - 0.3, torch.zeros(C, device="cuda")
    line("conv_in1 forward (122 MB store stream)", t(lambda: K.conv_in1_fwd(x0, w1, b1)), None, mb)
    line("conv_in1 wgrad (dy only)", t(lambda: K.conv_in1_wgrad(x0, dy1, dw1, db1, True, y=None)), None, mb)

    # ---- AAS-VC aligner: Conv1d(1536 -> 1536, k3) over 16 x 256 frames (forward / data-gradient form), 8-wave vs 4-wave kernel
    Bx, Tx, Cx, ks = 16, 256, 1536, 3
    xa, wa = u(Bx, Tx, Cx), u(Cx, ks * Cx) * 0.02
    ya, ba = torch.empty(Bx, Tx, Cx, dtype=dt, device="cuda"), torch.zeros(Cx, device="cuda")
    gfa = 2.0 * Bx * Tx * Cx * ks * Cx * 1e-9
    for on, name in ((1, "8-wave"), (0, "4-wave")):
        prev = L.s2svc_gemm_set_8ph(on)
        line(f"aligner Conv1d 4096 x 1536 x 4608, {name} kernel",
             t(lambda: K.gemm(K.operand(xa, Cx, mode=K.CONV1D, C=Cx, T=Tx, pad=1), K.operand(wa, ks * Cx), Bx * Tx, Cx, ks * Cx, ya, in_dtype=dt,
                              bias=ba, act="relu")), gfa)
        L.s2svc_gemm_set_8ph(prev)
    dya = u(Bx, Tx, Cx)
    dwa = torch.empty(Cx, ks * Cx, dtype=torch.float32, device="cuda")
    dba = torch.zeros(Cx, device="cuda")
    tile, sk = K.plan_gemm(Cx, ks * Cx, Bx * Tx)
    for w8 in (False, True):
        line(f"aligner Conv1d weight gradient 1536 x 4608 x 4096, w8={int(w8)} (plan tile {tile} split {sk})",
             t(lambda: K.gemm(K.operand(dya, Cx, layout=K.RC), K.operand(xa, Cx, layout=K.RC, mode=K.CONV1D, C=Cx, T=Tx, pad=1), Cx, ks * Cx,
                              Bx * Tx, dwa, in_dtype=dt, splitk=sk, tile=tile, a_rowsum=dba, a_rowsum_accumulate=True, wgrad=w8)), gfa)


if __name__ == "__main__":
    main()
