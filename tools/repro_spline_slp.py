#!/usr/bin/env python3
"""Reproducer for the packed-fp32 (SLP) hazard of DESIGN.md section 4 ("Reproducibility"): with clang's SLP vectoriser on, the
spline-gradient kernel of the stochastic duration predictor (csrc/sdp.hip: rq_spline_bwd_kernel) returned, in about 1 of 60 full
AAS-VC steps and only while another stream kept the chip busy, a wrong element in the last partially active 16-lane row of a wave.
The shipped library is built with -fno-slp-vectorize -fno-vectorize; this tool builds VARIANTS of the library on the GPU box
(hipcc is in the image) and looks for the failure in two ways:

  standalone : rq_spline_bwd alone, replayed N times from a hipGraph on one stream while a second stream runs chip-filling GEMMs;
               every launch's dx / dh is compared with the first launch's (bit for bit).  Inputs: every word of every buffer the
               kernel can read is initialised (the partial last rows of an utterance included).
  step       : the full AAS-VC vc2 forward + backward (duration predictor on the auxiliary stream) repeated N times with the same
               seeds and injected flow noise; losses and the flat gradient buffer compared with the first pass.

Variants (--variant): "shipped" (no SLP anywhere), "slp_sdp" (SLP on for sdp.hip only), "slp_all" (SLP on for every source),
"slp_files" (SLP on for the sources listed in --files: bisection).

    python tools/repro_spline_slp.py --variant slp_sdp --mode step --n 300
    python tools/repro_spline_slp.py --variant slp_sdp --mode standalone --n 20000

Prints one JSON line per run: {"variant", "mode", "n", "mismatches", ...}.  Results of round 4: profiles/r04_repro_spline_slp.txt.
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "seq2seq_vc_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "_scratch", "repro_slp")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def build_variant(variant, files=()):
    """-> path of a library built with the variant's flags (objects cached under tools/_scratch/repro_slp/<variant>/)."""
    if variant == "shipped":
        return None
    tag = variant if variant != "slp_files" else "slp_" + "_".join(sorted(f[:-4] for f in files))
    d = os.path.join(OUT, tag)
    os.makedirs(d, exist_ok=True)
    lib = os.path.join(d, "libs2svc_hip.so")
    if os.path.exists(lib):
        return lib
    objs = []
    for f in sorted(os.listdir(CSRC)):
        if not f.endswith(".hip"):
            continue
        slp = variant == "slp_all" or (variant == "slp_sdp" and f == "sdp.hip") or (variant == "slp_files" and f in files)
        if not slp:
            objs.append(os.path.join(CSRC, "build", f[:-4] + ".o"))          # the shipped object
            continue
        obj = os.path.join(d, f[:-4] + ".o")
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-c", os.path.join(CSRC, f), "-o", obj],
                       check=True)
        objs.append(obj)
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, check=True)
    return lib


def run_standalone(n, load):
    import torch
    from seq2seq_vc_amd.ops import kernels as K
    from seq2seq_vc_amd.ops import kernels_sdp as KS
    dev = torch.device("cuda")
    B, T, bins = 16, 64, 10
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(B, T, generator=g) * 2.5).to(dev)                      # some rows outside the +-5 tail bound, most inside
    h = torch.randn(B, T, 3 * bins - 1, generator=g).to(dev)
    lens = torch.randint(33, 65, (B,), generator=g).to(dev, torch.int32)    # partial last rows in most utterances
    lens[0], lens[1] = 64, 49
    g_out = torch.randn(B, T, generator=g).to(dev)
    g_lad = torch.randn(B, generator=g).to(dev)
    ref = KS.rq_spline_bwd(x, h, 1.0 / 384 ** 0.5, 5.0, lens, g_out, g_lad)
    ref = [t.clone() for t in ref]
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device=dev).to(torch.bfloat16)
    w = torch.randn(4096, 4096, device=dev).to(torch.bfloat16)
    y = torch.empty(4096, 4096, dtype=torch.bfloat16, device=dev)
    bad, done = 0, 0
    outs = []
    chunk = 200
    while done < n:
        if load:
            with torch.cuda.stream(side):
                for _ in range(6):
                    K.gemm(K.operand(a, 4096), K.operand(w, 4096), 4096, 4096, 4096, y, in_dtype=torch.bfloat16)
        outs = [KS.rq_spline_bwd(x, h, 1.0 / 384 ** 0.5, 5.0, lens, g_out, g_lad) for _ in range(chunk)]
        torch.cuda.synchronize()
        for o in outs:
            if not all(torch.equal(p, q) for p, q in zip(o, ref)):
                bad += 1
        done += chunk
    return {"launches": done, "mismatches": bad, "concurrent_gemm_stream": bool(load)}


def run_step(n):
    import torch
    import bench
    from seq2seq_vc_amd import losses as L
    from seq2seq_vc_amd import models as M
    from seq2seq_vc_amd.ops import functional as Fn
    from seq2seq_vc_amd.ops import kernels as K
    from seq2seq_vc_amd.optim import FlatAdam
    from tools.bench_aasvc import AASVC_VC2
    dev = torch.device("cuda")
    xs, ilens, ys, _, olens = bench.canonical_batch(16)
    xs_d, ys_d = xs.to(dev), ys.to(dev)
    Fn.set_compute_dtype(torch.bfloat16)
    Fn.enable_side_streams(0, inline_batches=True)
    torch.manual_seed(0)
    model = M.AASVC(**AASVC_VC2).to(dev).train()
    opt = FlatAdam(model, lr=8e-5, grad_norm=1.0, warmup_steps=4000, bf16_shadow=True)
    noise = torch.randn(16, 2, 64, generator=torch.Generator().manual_seed(5))
    name_of = {id(p): k for k, p in model.named_parameters()}

    def fwd_bwd():
        model.duration_predictor.noise = noise
        K.manual_seed(1234)
        K.reset_op_counter()
        opt.zero_grad()
        ret = model(xs_d, ilens, ys_d, olens, xs_d, dp_lengths=ilens)
        l1 = L.L1Loss()(ret["after_outs"], ret["before_outs"], ret["ys"], ret["olens"])
        fs = L.ForwardSumLoss()(ret["log_p_attn"], ret["ilens"], ret["olens_reduced"])
        dur = torch.sum(ret["dur_nll"].float())
        (l1 + 2.0 * (fs + ret["bin_loss"]) + dur).backward()
        Fn.side_join()
        torch.cuda.synchronize()
        return torch.stack([l1.detach().float(), fs.detach().float(), dur.detach().float()]), opt.flat_g.clone()

    l0, g0 = fwd_bwd()
    bad, where = 0, {}
    for _ in range(n):
        l, g = fwd_bwd()
        if not (torch.equal(l, l0) and torch.equal(g, g0)):
            bad += 1
            diff = (g != g0).nonzero().flatten()
            lo = int(diff[0]) if diff.numel() else -1
            for off, p in zip(opt.offsets, opt.params):
                if off <= lo < off + p.numel():
                    k = name_of.get(id(p), "?")
                    where[k] = where.get(k, 0) + 1
    return {"steps": n, "mismatches": bad, "first_differing_parameter_counts": where}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", default="slp_sdp", choices=["shipped", "slp_sdp", "slp_all", "slp_files"])
    ap.add_argument("--files", default="", help="slp_files: comma-separated sources built WITH the SLP / loop vectorisers (bisection)")
    ap.add_argument("--mode", default="step", choices=["step", "standalone"])
    ap.add_argument("--n", type=int, default=300)
    ap.add_argument("--no-load", action="store_true", help="standalone: no concurrent GEMM stream")
    a = ap.parse_args()
    files = tuple(f for f in a.files.split(",") if f)
    lib = build_variant(a.variant, files)
    if lib is not None and os.environ.get("S2SVC_LIB") != lib:
        os.environ["S2SVC_LIB"] = lib                       # _lib.py reads it at import time: re-exec with it set
        os.execv(sys.executable, [sys.executable] + sys.argv)
    res = run_step(a.n) if a.mode == "step" else run_standalone(a.n, not a.no_load)
    print(json.dumps({"variant": a.variant, "files": list(files), "mode": a.mode, **res}), flush=True)


if __name__ == "__main__":
    main()
