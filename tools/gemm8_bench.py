#!/usr/bin/env python3
"""A/B timing of the bf16 GEMM kernel families on the forward / data-gradient shapes of the two workloads:
mode 0 = 128x128 4-wave kernel (gemm_glds.hip), 1 = 8-wave phase-interleaved kernel by policy (gemm_8ph.hip), 2 = the same
without the half-phase skew, 17 / 33 / 49 = mode 1 with the tile geometry forced to 256x256 / 512x128 / 256x128; + 256 (v + 1) sets the
one-round 256 x 96 p geometry (gemm_8ph_kernel_n96): 257 = policy without it, 513 = policy with it (the default), 769 = wherever N % 96 p == 0 (widest p), 1281 / 1537 = p = 2 / 3.  Variants are interleaved in rounds inside ONE process (medians reported); every timing is
`--iters` launches replayed from one hipGraph with HIP events around the replay; operands uniform random in [-1, 1).

    python tools/gemm8_bench.py [--iters 50] [--rounds 5] [--modes 0,1,2]
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

DENSE = [("4096^3", 4096, 4096, 4096), ("8192^3", 8192, 8192, 8192), ("aas 4096x1536x1536", 4096, 1536, 1536),
         ("aas 4096x3072x1536", 4096, 3072, 1536), ("aas 4096x4608x1536 (packed Q|K|V)", 4096, 4608, 1536),
         ("aas 4096x1536x4608 (its dgrad)", 4096, 1536, 4608), ("aas 4096x1536x3072 (dgrad of pw1)", 4096, 1536, 3072),
         ("aas 4096x1536x384", 4096, 1536, 384), ("vtn 2048x1536x384", 2048, 1536, 384), ("vtn 2016x384x7296", 2016, 384, 7296)]


def ksweep(a):
    """Where a short-K GEMM launch spends its time: t(K) = fixed + (K / 64) * per_tile, fitted over K; the fixed part against the
    floor of a dependent do-nothing launch (s2svc_launch_floor) -- what is left is prologue (first operand tiles: one trip to HBM)
    + epilogue."""
    L = K._lib.lib()
    dt = torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(1)
    u = lambda *s: (torch.rand(*s, device="cuda", generator=g) * 2 - 1).to(dt)
    sink = torch.zeros(1024, dtype=torch.int32, device="cuda")
    floor = {w: statistics.median(bench(lambda w=w: K.launch_floor(sink, w, 256 if w != 192 else 512), a.iters) for _ in range(a.rounds))
             for w in (1, 192, 256, 2048)}
    print("launch floor (do-nothing kernel, dependent launches of one graph), us per launch: " +
          ", ".join(f"{w} workgroups {t:.2f}" for w, t in floor.items()))
    prev = L.s2svc_gemm_set_8ph(1 | (3 << 4))           # 256 x 128 tiles forced
    for (M, N) in ((4096, 1536), (4096, 4608)):
        pts = []
        for Kd in (128, 256, 384, 768, 1536, 3072, 4608):
            x, w, y, b = u(M, Kd), u(N, Kd), torch.empty(M, N, dtype=dt, device="cuda"), torch.zeros(N, device="cuda")
            fn = lambda: K.gemm(K.operand(x, Kd), K.operand(w, Kd), M, N, Kd, y, in_dtype=dt, bias=b)
            us = statistics.median(bench(fn, a.iters) for _ in range(a.rounds))
            pts.append((Kd // 64, us))
            print(f"  {M} x {N} x {Kd:5d}: {us:7.2f} us  ({2.0 * M * N * Kd / us / 1e6 / 2500 * 100:5.1f} % of the bf16 peak)", flush=True)
        n = len(pts)
        sx, sy = sum(p[0] for p in pts), sum(p[1] for p in pts)
        sxx, sxy = sum(p[0] * p[0] for p in pts), sum(p[0] * p[1] for p in pts)
        slope = (n * sxy - sx * sy) / (n * sxx - sx * sx)
        fixed = (sy - slope * sx) / n
        tiles = ((M + 255) // 256) * ((N + 127) // 128)
        rounds = (tiles + 255) // 256
        ideal = 2.0 * 256 * 128 * 64 / (2500e12 / 256) * 1e6 * rounds      # us per K tile of one workgroup at the MFMA peak, per round
        print(f"{M} x {N}: {tiles} workgroups ({rounds} round(s) on 256 CUs): fixed {fixed:.2f} us per launch (launch floor {floor[256]:.2f}) "
              f"+ {slope:.3f} us per K tile of 64 (MFMA peak: {ideal:.3f})", flush=True)
    L.s2svc_gemm_set_8ph(prev)
    # the small GEMMs of the VTN chain (4-wave LDS-DMA kernel, tiles of 32 / 64 rows by policy): the same fit
    for (M, N) in ((2016, 384), (2016, 1152), (2016, 1536)):
        pts = []
        for Kd in (64, 128, 384, 768, 1536):
            x, w, y, b = u(M, Kd), u(N, Kd), torch.empty(M, N, dtype=dt, device="cuda"), torch.zeros(N, device="cuda")
            fn = lambda: K.gemm(K.operand(x, Kd), K.operand(w, Kd), M, N, Kd, y, in_dtype=dt, bias=b)
            us = statistics.median(bench(fn, a.iters) for _ in range(a.rounds))
            pts.append((Kd // 64, us))
            print(f"  {M} x {N} x {Kd:5d}: {us:7.2f} us", flush=True)
        n = len(pts)
        sx, sy = sum(p[0] for p in pts), sum(p[1] for p in pts)
        sxx, sxy = sum(p[0] * p[0] for p in pts), sum(p[0] * p[1] for p in pts)
        slope = (n * sxy - sx * sy) / (n * sxx - sx * sx)
        print(f"{M} x {N} (VTN chain, kernel by policy): fixed {(sy - slope * sx) / n:.2f} us per launch (launch floor {floor[256]:.2f}) + "
              f"{slope:.3f} us per K tile of 64", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--modes", default="0,257,513,273,289,305,769")
    ap.add_argument("--ksweep", action="store_true",
                    help="instead: 4096 x 1536 x K for a range of K on the 256 x 128 8-wave kernel -> fixed cost per launch + us per K tile "
                         "(least squares), beside the launch floor of a do-nothing kernel")
    a = ap.parse_args()
    if a.ksweep:
        return ksweep(a)
    modes = [int(m) for m in a.modes.split(",")]
    L = K._lib.lib()
    dt = torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(1)
    u = lambda *s: (torch.rand(*s, device="cuda", generator=g) * 2 - 1).to(dt)
    cases = []
    for name, M, N, Kd in DENSE:
        x, w, y = u(M, Kd), u(N, Kd), torch.empty(M, N, dtype=dt, device="cuda")
        b = torch.zeros(N, device="cuda")
        cases.append((name, 2.0 * M * N * Kd, lambda x=x, w=w, y=y, b=b, M=M, N=N, Kd=Kd: K.gemm(K.operand(x, Kd), K.operand(w, Kd), M, N, Kd, y, in_dtype=dt, bias=b)))
    B, T1, F1, C, O = 32, 127, 39, 384, 384
    T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
    M, N, Kd = B * T2 * F2, O, 9 * C
    x, w, y = u(B, T1, F1, C), u(O, 9 * C), torch.empty(B, T2, F2, O, dtype=dt, device="cuda")
    b = torch.zeros(O, device="cuda")
    cases.append(("vtn conv2d 38304x384x3456", 2.0 * M * N * Kd,
                  lambda: K.gemm(K.operand(x, C, mode=K.CONV2D_S2, C=C, T1=T1, F1=F1, T2=T2, F2=F2), K.operand(w, 9 * C), M, N, Kd, y,
                                 in_dtype=dt, bias=b, act="relu")))
    prev = L.s2svc_gemm_set_8ph(-1)
    print(f"{'shape':40s} " + " ".join(f"{'mode ' + str(m) + ' us':>11s} {'TF':>7s} {'%peak':>6s}" for m in modes))
    for name, flops, fn in cases:
        t = {m: [] for m in modes}
        for _ in range(a.rounds):
            for m in modes:
                L.s2svc_gemm_set_8ph(m)
                t[m].append(bench(fn, a.iters))
        row = f"{name:40s} "
        for m in modes:
            us = statistics.median(t[m])
            tf = flops / us / 1e6
            row += f"{us:11.1f} {tf:7.0f} {100 * tf / 2500:6.1f} "
        print(row, flush=True)
    L.s2svc_gemm_set_8ph(prev)


if __name__ == "__main__":
    main()
