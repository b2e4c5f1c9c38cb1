#!/usr/bin/env python3
"""Micro-benchmark of s2svc_gemm on the GEMM shapes of the two training workloads (VTN vc1, AAS-VC vc2).

    python tools/gemm_bench.py [--iters 50] [--dtype bf16] [--filter substr]

Each line: the operand layouts as the autograd code issues them (fwd = KC x KC, dgrad = KC x RC, wgrad = RC x RC with the
reduction over the B*T rows), average launch time over `--iters` back-to-back launches replayed from one hipGraph (HIP events around the replay),
achieved TFLOP/s and the fraction of the dense bf16 MFMA peak (2.5 PFLOP/s).  Inputs are uniform random in [-1, 1).
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from seq2seq_vc_amd.ops import kernels as K  # noqa: E402

PEAK = {torch.bfloat16: 2500.0, torch.float32: 157.3}

# (name, rows = B*T, in_features, out_features)
LINEARS = [
    ("aas dec ffn/attn 1536x1536 (B16 T256)", 4096, 1536, 1536),
    ("aas dec conv-pw1 1536->3072", 4096, 1536, 3072),
    ("aas enc ffn 384->1536", 4096, 384, 1536),
    ("aas enc ffn 1536->384", 4096, 1536, 384),
    ("aas enc attn 384->384", 4096, 384, 384),
    ("vtn enc attn 384->384 (B32 T63)", 2016, 384, 384),
    ("vtn enc ffn 384->1536", 2016, 384, 1536),
    ("vtn dec ffn 384->1536 (B32 T64)", 2048, 384, 1536),
    ("vtn dec qkv 384->1152", 2048, 384, 1152),
    ("vtn embed 7296->384", 2016, 7296, 384),
    ("square 4096^3 (reference point of the CDNA4 guide)", 4096, 4096, 4096),
]


def bench(fn, iters):
    """Average time of one launch: `iters` launches replayed from one hipGraph (the events then bracket kernels; a Python
    launch loop adds ~10 us of host time per ctypes launch, more than the short GEMMs take)."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for _ in range(iters):
                fn()
        g.replay()
        side.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(side)
        g.replay()
        e1.record(side)
        side.synchronize()
    torch.cuda.current_stream().wait_stream(side)
    return e0.elapsed_time(e1) * 1e3 / iters   # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--filter", default="")
    ap.add_argument("--sweep", action="store_true", help="also time every (tile, split-K) choice of the wgrad GEMMs")
    a = ap.parse_args()
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    dev = "cuda"
    rnd = lambda *s: (torch.rand(*s, device=dev) * 2 - 1).to(dtype)
    print(f"{'case':58s} {'M':>6s} {'N':>6s} {'K':>6s} {'us':>9s} {'TFLOP/s':>9s} {'%peak':>6s}")
    tot = 0.0

    def line(name, M, N, Kd, us):
        nonlocal tot
        tf = 2.0 * M * N * Kd / us / 1e6
        tot += us
        print(f"{name:58s} {M:6d} {N:6d} {Kd:6d} {us:9.1f} {tf:9.1f} {100 * tf / PEAK[dtype]:6.1f}")

    for name, rows, fin, fout in LINEARS:
        if a.filter and a.filter not in name:
            continue
        x, w, dy = rnd(rows, fin), rnd(fout, fin) * 0.05, rnd(rows, fout)
        b = torch.zeros(fout, device=dev)
        y = torch.empty(rows, fout, dtype=dtype, device=dev)
        dx = torch.empty(rows, fin, dtype=dtype, device=dev)
        dw = torch.empty(fout, fin, dtype=torch.float32, device=dev)
        line("fwd   " + name, rows, fout, fin,
             bench(lambda: K.gemm(K.operand(x, fin), K.operand(w, fin), rows, fout, fin, y, in_dtype=dtype, bias=b), a.iters))
        line("dgrad " + name, rows, fin, fout,
             bench(lambda: K.gemm(K.operand(dy, fout), K.operand(w, fin, layout=K.RC), rows, fin, fout, dx, in_dtype=dtype), a.iters))
        tile, sk = K.plan_gemm(fout, fin, rows)
        line(f"wgrad {name} [tile {tile} splitk {sk}]", fout, fin, rows,
             bench(lambda: K.gemm(K.operand(dy, fout, layout=K.RC), K.operand(x, fin, layout=K.RC), fout, fin, rows, dw,
                                  in_dtype=dtype, splitk=sk, tile=tile), a.iters))
        if a.sweep:        # every (tile, split-K) choice for the wgrad GEMM: the data plan_gemm's cost model is fitted to
            cells = []
            for t in (64, 128):
                if t == 128 and (fout < 128 or fin < 128):
                    continue
                for s in (1, 2, 3, 4, 6, 8, 16):
                    if s > 1 and rows // (64 * s) < 2:
                        continue
                    us = bench(lambda: K.gemm(K.operand(dy, fout, layout=K.RC), K.operand(x, fin, layout=K.RC), fout, fin, rows, dw,
                                              in_dtype=dtype, splitk=s, tile=t), max(10, a.iters // 3))
                    cells.append(f"{t}/{s}:{us:.1f}")
            print("      sweep tile/splitk:us  " + "  ".join(cells))
    # the FLOP-heaviest VTN op: implicit-GEMM Conv2d 3x3 stride 2 (subsampling.py:60)
    if not a.filter or a.filter in "conv2d":
        B, T1, F1, C, O = 32, 127, 39, 384, 384
        T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
        M, N, Kd = B * T2 * F2, O, 9 * C
        x, w = rnd(B, T1, F1, C), rnd(O, 9 * C) * 0.02
        y = torch.empty(B, T2, F2, O, dtype=dtype, device=dev)
        b = torch.zeros(O, device=dev)
        line("fwd   vtn conv2d 3x3 s2 implicit", M, N, Kd,
             bench(lambda: K.gemm(K.operand(x, C, mode=K.CONV2D_S2, C=C, T1=T1, F1=F1, T2=T2, F2=F2), K.operand(w, 9 * C), M, N, Kd, y,
                                  in_dtype=dtype, bias=b, act="relu"), a.iters))
        dy = rnd(B, T2, F2, O)
        dcols = torch.empty(M, 9 * C, dtype=dtype, device=dev)
        line("dgrad vtn conv2d dcols = dY.Wp", M, 9 * C, O,
             bench(lambda: K.gemm(K.operand(dy, O), K.operand(w, 9 * C, layout=K.RC), M, 9 * C, O, dcols, in_dtype=dtype), a.iters))
        line("dgrad vtn conv2d col2im (us only)", 0, 0, 0, bench(lambda: K.col2im_s2(dcols, B, T1, F1, C, T2, F2), a.iters))
        if dtype == torch.bfloat16:
            w4 = (rnd(O, C, 3, 3) * 0.02).float()
            wts = K.tconv2d_weights(w4)
            dx = torch.empty(B, T1, F1, C, dtype=dtype, device=dev)

            def tconv():
                for cls, wt in enumerate(wts):
                    pt, pf = cls >> 1, cls & 1
                    Tc, Fc = (T1 - pt + 1) // 2, (F1 - pf + 1) // 2
                    K.gemm(K.operand(dy, O, mode=K.TCONV2D_S2, C=O, T1=Tc, F1=Fc, T2=T2, F2=F2, pad=cls), K.operand(wt, wt.shape[1]),
                           B * Tc * Fc, C, wt.shape[1], dx, in_dtype=dtype, c_map=(T1, F1, Tc, Fc, pt, pf))
            line("dgrad vtn conv2d transposed conv, 4 parity-class GEMMs", M, 9 * C, O, bench(tconv, a.iters))

            def tconv_grouped():
                descs = []
                for cls, wt in enumerate(wts):
                    pt, pf = cls >> 1, cls & 1
                    Tc, Fc = (T1 - pt + 1) // 2, (F1 - pf + 1) // 2
                    K.gemm(K.operand(dy, O, mode=K.TCONV2D_S2, C=O, T1=Tc, F1=Fc, T2=T2, F2=F2, pad=cls), K.operand(wt, wt.shape[1]),
                           B * Tc * Fc, C, wt.shape[1], dx, in_dtype=dtype, c_map=(T1, F1, Tc, Fc, pt, pf), group=descs)
                K.launch_group(descs)
            line("dgrad vtn conv2d transposed conv, the 4 classes as one grid", M, 9 * C, O, bench(tconv_grouped, a.iters))
        dwp = torch.empty(O, 9 * C, dtype=torch.float32, device=dev)
        for tile, sk in ((128, 4), (128, 6), (128, 8), (128, 12), (64, 4), (64, 6)):
            line(f"wgrad vtn conv2d implicit [tile {tile} splitk {sk}] plan={K.plan_gemm(O, 9 * C, M)}", O, 9 * C, M,
                 bench(lambda: K.gemm(K.operand(dy, O, layout=K.RC),
                                      K.operand(x, C, layout=K.RC, mode=K.CONV2D_S2, C=C, T1=T1, F1=F1, T2=T2, F2=F2), O, 9 * C, M, dwp,
                                      in_dtype=dtype, splitk=sk, tile=tile), a.iters))
    # AAS-VC decoder weight gradients (4096 rows, d = 1536): un-grouped on the 8-wave kernel vs the 4-wave kernel, with the fused
    # bias row-sums; and five layers' worth as one grouped launch
    if (not a.filter or a.filter in "aas wgrad 8-wave") and dtype == torch.bfloat16:
        L = K._lib.lib()
        prev = L.s2svc_gemm_set_8ph(-1)
        for rows, fin, fout in ((4096, 1536, 1536), (4096, 1536, 3072), (4096, 3072, 1536), (4096, 1536, 4608)):
            x, dy = rnd(rows, fin), rnd(rows, fout)
            dw = torch.zeros(fout, fin, device=dev)
            db = torch.zeros(fout, device=dev)
            for mode, nm in ((0, "4-wave 128x128"), (1 | (3 << 4), "8-wave 256x128"), (1 | (1 << 4), "8-wave 256x256"), (1, "8-wave policy")):
                L.s2svc_gemm_set_8ph(mode)
                line(f"wgrad aas {fout}x{fin} + bias row-sums, {nm}", fout, fin, rows,
                     bench(lambda: K.gemm(K.operand(dy, fout, layout=K.RC), K.operand(x, fin, layout=K.RC), fout, fin, rows, dw,
                                          in_dtype=dtype, accumulate=True, a_rowsum=db, a_rowsum_accumulate=True), a.iters))
        probs = [(rnd(4096, fin), rnd(4096, fout), torch.zeros(fout, fin, device=dev), torch.zeros(fout, device=dev))
                 for fin, fout in ((1536, 1536), (1536, 3072), (3072, 1536), (1536, 1536), (1536, 1536))]
        gf = sum(2.0 * 4096 * p[0].shape[1] * p[1].shape[1] for p in probs)
        saved = K._GROUP_MAX_TILES
        K._GROUP_MAX_TILES = 1 << 30

        def grouped():
            q = []
            with K.record_grouped(q):
                for x, dy, dw, db in probs:
                    K.gemm(K.operand(dy, dy.shape[1], layout=K.RC), K.operand(x, x.shape[1], layout=K.RC), dy.shape[1], x.shape[1], 4096, dw,
                           in_dtype=dtype, accumulate=True, a_rowsum=db, a_rowsum_accumulate=True)
            K.flush_grouped(q)
        for mode, nm in ((0, "4-wave"), (1 | (3 << 4), "8-wave 256x128"), (1 | (1 << 4), "8-wave 256x256"), (1, "8-wave policy")):
            L.s2svc_gemm_set_8ph(mode)
            us = bench(grouped, a.iters)
            tot += us
            print(f"{'wgrad aas grouped launch of 5 problems, ' + nm:58s} {'':6s} {'':6s} {'':6s} {us:9.1f} {gf / us / 1e6:9.1f} "
                  f"{100 * gf / us / 1e6 / PEAK[dtype]:6.1f}")
        K._GROUP_MAX_TILES = saved
        L.s2svc_gemm_set_8ph(prev)
    # VTN weight gradients as training launches them: the dense problems of two encoder layers (8) / one decoder layer + the stacked
    # K|V projection (7) as ONE grouped launch -- the 4-wave 64 x 64 kernel (until round 3) against the ragged 8-wave kernel (W8)
    if (not a.filter or a.filter in "vtn wgrad grouped w8") and dtype == torch.bfloat16:
        L = K._lib.lib()
        prev = L.s2svc_gemm_set_w8(-1, 0)
        groups = {"2 encoder layers (8 problems, K 2016)": [(2016, 384, 1152), (2016, 384, 384), (2016, 384, 1536), (2016, 1536, 384)] * 2,
                  "decoder layer + stacked K|V (8 problems, K 2048)": [(2048, 384, 1152), (2048, 384, 384), (2048, 384, 384), (2048, 384, 384),
                                                                       (2048, 384, 1536), (2048, 1536, 384), (2016, 384, 4608)],
                  "embed 7296->384 alone": [(2016, 7296, 384)],
                  "aas encoder layer (K 4096)": [(4096, 384, 1152), (4096, 384, 384), (4096, 384, 1536), (4096, 1536, 384), (4096, 384, 1536),
                                                 (4096, 1536, 384), (4096, 384, 768), (4096, 384, 384)]}
        for gname, shapes in groups.items():
            probs = [(rnd(rows, fin), rnd(rows, fout), torch.zeros(fout, fin, device=dev), torch.zeros(fout, device=dev)) for rows, fin, fout in shapes]
            gf = sum(2.0 * x.shape[0] * x.shape[1] * dy.shape[1] for x, dy, _, _ in probs)

            def grouped():
                q = []
                with K.record_grouped(q):
                    for x, dy, dw, db in probs:
                        K.gemm(K.operand(dy, dy.shape[1], layout=K.RC), K.operand(x, x.shape[1], layout=K.RC), dy.shape[1], x.shape[1], x.shape[0],
                               dw, in_dtype=dtype, accumulate=True, a_rowsum=db, a_rowsum_accumulate=True)
                K.flush_grouped(q)
            for on, kt, nm in ((0, 0, "4-wave 64x64"), (1, 64, "W8 unsplit"), (1, 32, "W8 32-tile chunks"), (1, 16, "W8 16-tile chunks"), (1, 8, "W8 8-tile chunks")):
                L.s2svc_gemm_set_w8(on, kt)
                us = bench(grouped, a.iters)
                tot += us
                print(f"{'wgrad ' + gname + ', ' + nm:90s} {us:9.1f} {gf / us / 1e6:9.1f} {100 * gf / us / 1e6 / PEAK[dtype]:6.1f}")
        L.s2svc_gemm_set_w8(prev & 1, prev >> 8)
    # postnet Conv1d k5 256->256 as implicit GEMM (B32 T256)
    if not a.filter or a.filter in "conv1d":
        B, T, Cin, Cout, ks = 32, 256, 256, 256, 5
        x, w = rnd(B, T, Cin), rnd(Cout, ks * Cin) * 0.02
        y = torch.empty(B, T, Cout, dtype=dtype, device=dev)
        line("fwd   vtn postnet conv1d k5 256->256 implicit", B * T, Cout, ks * Cin,
             bench(lambda: K.gemm(K.operand(x, Cin, mode=K.CONV1D, C=Cin, T=T, pad=2), K.operand(w, ks * Cin), B * T, Cout, ks * Cin, y,
                                  in_dtype=dtype), a.iters))
    print(f"{'sum of averages':58s} {'':6s} {'':6s} {'':6s} {tot:9.1f}")


if __name__ == "__main__":
    main()
