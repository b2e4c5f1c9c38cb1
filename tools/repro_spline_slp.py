#!/usr/bin/env python3
"""Reproducer for the packed-fp32 (SLP) hazard of DESIGN.md section 5 ("Hazard"; full account in profiles/AB_LOG.md): with clang's SLP vectoriser on, the
spline-gradient kernel of the stochastic duration predictor (csrc/sdp.hip: rq_spline_bwd_kernel) returned, in about 1 of 60 full
AAS-VC steps and only while another stream kept the chip busy, a wrong element in the last partially active 16-lane row of a wave.
The shipped library is built with -fno-slp-vectorize -fno-vectorize; this tool builds VARIANTS of the library on the GPU box
(hipcc is in the image) and looks for the failure in two ways:

  standalone : rq_spline_bwd alone, replayed N times from a hipGraph on one stream while a second stream runs chip-filling GEMMs;
               every launch's dx / dh is compared with the first launch's (bit for bit).  Inputs: every word of every buffer the
               kernel can read is initialised (the partial last rows of an utterance included).
  dirty      : rq_spline_bwd on the recorded operands of every call of one real backward pass, each launch preceded by
               tools/vgpr_dirty.hip (a kernel that leaves a chosen bit pattern in every VGPR): a result that follows the pattern
               reads a register it never wrote.
  trace      : full steps with every rq_spline_bwd output cloned right behind the launch: where a failing pass's dh differs.
  step       : the full AAS-VC vc2 forward + backward (duration predictor on the auxiliary stream) repeated N times with the same
               seeds and injected flow noise; losses and the flat gradient buffer compared with the first pass.

Variants (--variant): "shipped" (no SLP anywhere), "slp_sdp" (SLP on for sdp.hip only), "slp_all" (SLP on for every source),
"slp_files" (SLP on for the sources listed in --files: bisection), "slp_sdp_nopk" (SLP on for sdp.hip, packed-fp32 instruction
selection off: -target-feature -packed-fp32-ops).

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
        slp = variant == "slp_all" or (variant in ("slp_sdp", "slp_sdp_nopk") and f == "sdp.hip") or (variant == "slp_files" and f in files)
        if not slp:
            objs.append(os.path.join(CSRC, "build", f[:-4] + ".o"))          # the shipped object
            continue
        obj = os.path.join(d, f[:-4] + ".o")
        # slp_sdp_nopk: the SLP vectoriser stays ON but the target may not select packed-fp32 VALU instructions (v_pk_{mul,fma,add}_f32:
        # 756 of them in sdp.hip with the feature, 0 without) -- separates "the vectorised IR is wrong" from "the packed instructions are"
        nopk = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"] if variant == "slp_sdp_nopk" else []
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", *nopk, "-c", os.path.join(CSRC, f), "-o", obj],
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
    bad, where, detail = 0, {}, []
    for it in range(n):
        l, g = fwd_bwd()
        if not (torch.equal(l, l0) and torch.equal(g, g0)):
            bad += 1
            diff = (g != g0).nonzero().flatten()
            lo = int(diff[0]) if diff.numel() else -1
            every = []                                         # EVERY parameter that differs, with the shape of the damage
            for off, p in zip(opt.offsets, opt.params):
                seg = diff[(diff >= off) & (diff < off + p.numel())] - off
                if seg.numel():
                    k = name_of.get(id(p), "?")
                    if off <= lo < off + p.numel():
                        where[k] = where.get(k, 0) + 1
                    a, b = g[off:off + p.numel()], g0[off:off + p.numel()]
                    cols = p.shape[-1] if p.dim() > 1 else 1
                    rows_hit = sorted({int(i) // cols for i in seg.tolist()})
                    cols_hit = sorted({int(i) % cols for i in seg.tolist()})
                    every.append({"param": k, "shape": list(p.shape), "elements": int(seg.numel()),
                                  "rows": rows_hit[:12] + (["..."] if len(rows_hit) > 12 else []),
                                  "cols": cols_hit[:12] + (["..."] if len(cols_hit) > 12 else []),
                                  "max_abs_diff": float((a - b).abs().max()), "max_abs_ref": float(b.abs().max())})
            if len(detail) < 6:
                detail.append({"pass": it, "losses_equal": bool(torch.equal(l, l0)), "differing": every[:12], "n_params_differing": len(every)})
    return {"steps": n, "mismatches": bad, "first_differing_parameter_counts": where, "detail": detail}


def run_trace(n):
    """Full steps as in run_step, with the outputs of every rq_spline_bwd call cloned right behind the launch (same stream): when a
    pass's gradients differ from the first pass's, was dh ALREADY different when the kernel finished, where, and does the kernel
    give the right answer when it is re-run on the same operands afterwards?"""
    import torch
    import bench
    from seq2seq_vc_amd import losses as L
    from seq2seq_vc_amd import models as M
    from seq2seq_vc_amd.ops import functional as Fn
    from seq2seq_vc_amd.ops import kernels as K
    from seq2seq_vc_amd.ops import kernels_sdp as KS
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
    orig = KS.rq_spline_bwd
    trace = []

    def rec(x, h, hscale, bound, lens, g_out, g_lad):
        dx, dh = orig(x, h, hscale, bound, lens, g_out, g_lad)
        trace.append((dx.clone(), dh.clone(), (x, h, hscale, bound, lens, g_out, g_lad), dh))
        return dx, dh
    KS.rq_spline_bwd = rec

    def fwd_bwd():
        del trace[:]
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
        return opt.flat_g.clone(), [(a.clone(), b.clone()) for a, b, _, _ in trace]

    g0, t0 = fwd_bwd()
    bad, events = 0, []
    for it in range(n):
        g, t = fwd_bwd()
        if torch.equal(g, g0):
            continue
        bad += 1
        ev = {"pass": it, "calls": []}
        for ci, ((dx, dh), (dx0, dh0)) in enumerate(zip(t, t0)):
            if torch.equal(dx, dx0) and torch.equal(dh, dh0):
                continue
            d = (dh != dh0).nonzero()
            args, live = trace[ci][2], trace[ci][3]
            again = orig(*args)                                  # the kernel once more on the very same operand tensors
            torch.cuda.synchronize()
            e = {"call": ci, "dh_elements_wrong_in_the_clone": int(d.shape[0]), "dx_elements_wrong": int((dx != dx0).sum()),
                 "live_dh_equals_its_clone": bool(torch.equal(live, dh)), "rerun_equals_first_pass": bool(torch.equal(again[1], dh0)),
                 "operands_equal_first_pass": None}
            if d.shape[0]:
                b_, t_, j_ = [int(v) for v in d[0]]
                xv = float(args[0][b_, t_])
                e.update({"first": [b_, t_, j_], "all_j": sorted({int(v[2]) for v in d.tolist()}), "all_rows": sorted({(int(v[0]), int(v[1])) for v in d.tolist()})[:8],
                          "len_of_utt": int(args[4][b_]), "x": xv, "got": float(dh[b_, t_, j_]), "ref": float(dh0[b_, t_, j_]),
                          "row_got": [float(v) for v in dh[b_, t_]], "row_ref": [float(v) for v in dh0[b_, t_]]})
            ev["calls"].append(e)
        if not ev["calls"]:
            ev["note"] = "gradients differ but every cloned spline output equals the first pass's"
        if len(events) < 5:
            events.append(ev)
    KS.rq_spline_bwd = orig
    return {"steps": n, "mismatches": bad, "events": events}


def run_dirty(n):
    """The spline-gradient kernel on the REAL operands of every call of one AAS-VC backward pass, each launch preceded by a
    kernel that leaves a chosen bit pattern in every VGPR of every SIMD (tools/vgpr_dirty.hip): does the result follow what a
    previous wave left in the registers?"""
    import ctypes
    import torch
    import bench
    from seq2seq_vc_amd import losses as L
    from seq2seq_vc_amd import models as M
    from seq2seq_vc_amd.ops import functional as Fn
    from seq2seq_vc_amd.ops import kernels as K
    from seq2seq_vc_amd.ops import kernels_sdp as KS
    from seq2seq_vc_amd.optim import FlatAdam
    from tools.bench_aasvc import AASVC_VC2
    so = os.path.join(OUT, "vgpr_dirty.so")
    os.makedirs(OUT, exist_ok=True)
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(ROOT, "tools", "vgpr_dirty.hip"), "-o", so], check=True)
    dirty = ctypes.CDLL(so).vgpr_dirty_launch
    dirty.argtypes = [ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p]
    dev = torch.device("cuda")
    xs, ilens, ys, _, olens = bench.canonical_batch(16)
    xs_d, ys_d = xs.to(dev), ys.to(dev)
    Fn.set_compute_dtype(torch.bfloat16)
    Fn.enable_side_streams(0, inline_batches=True)
    torch.manual_seed(0)
    model = M.AASVC(**AASVC_VC2).to(dev).train()
    opt = FlatAdam(model, lr=8e-5, grad_norm=1.0, warmup_steps=4000, bf16_shadow=True)
    model.duration_predictor.noise = torch.randn(16, 2, 64, generator=torch.Generator().manual_seed(5))
    calls = []
    orig = KS.rq_spline_bwd

    def rec(x, h, hscale, bound, lens, g_out, g_lad):
        calls.append((x.clone(), h.clone(), hscale, bound, lens.clone(), g_out.clone(), g_lad.clone()))
        return orig(x, h, hscale, bound, lens, g_out, g_lad)
    KS.rq_spline_bwd = rec
    import seq2seq_vc_amd.ops.functional_sdp as FS
    FS.KS.rq_spline_bwd = rec
    K.manual_seed(1234)
    K.reset_op_counter()
    opt.zero_grad()
    ret = model(xs_d, ilens, ys_d, olens, xs_d, dp_lengths=ilens)
    l1 = L.L1Loss()(ret["after_outs"], ret["before_outs"], ret["ys"], ret["olens"])
    fs = L.ForwardSumLoss()(ret["log_p_attn"], ret["ilens"], ret["olens_reduced"])
    (l1 + 2.0 * (fs + ret["bin_loss"]) + torch.sum(ret["dur_nll"].float())).backward()
    Fn.side_join()
    torch.cuda.synchronize()
    KS.rq_spline_bwd = orig
    FS.KS.rq_spline_bwd = orig
    main, side = torch.cuda.current_stream(), torch.cuda.Stream()
    patterns = [0x00000000, 0x7fc00000, 0x3f800000, 0xbf800000, 0x7f800000, 0x00000001, 0xdeadbeef, 0x41200000]
    out = {"calls": len(calls), "per_call": []}
    for ci, c in enumerate(calls):
        torch.cuda.synchronize()
        ref = [t.clone() for t in orig(*c)]
        torch.cuda.synchronize()
        stat = {"call": ci, "rows": int(c[0].numel()), "same_stream": {}, "other_stream": {}, "no_dirt": 0, "examples": []}
        for _ in range(n):                                     # control: back-to-back launches, nothing in between
            o = orig(*c)
            if not all(torch.equal(a, b) for a, b in zip(o, ref)):
                stat["no_dirt"] += 1
        for where in ("same_stream", "other_stream"):
            for pat in patterns:
                bad = 0
                for _ in range(n):
                    if where == "same_stream":
                        dirty(pat, 2048, ctypes.c_void_p(main.cuda_stream))
                    else:
                        dirty(pat, 8192, ctypes.c_void_p(side.cuda_stream))
                    o = orig(*c)
                    if not all(torch.equal(a, b) for a, b in zip(o, ref)):
                        bad += 1
                        if len(stat["examples"]) < 4:
                            d = (o[1] != ref[1]).nonzero()
                            ex = {"pattern": hex(pat), "where": where, "dh_elements": int(d.shape[0]), "dx_elements": int((o[0] != ref[0]).sum())}
                            if d.shape[0]:
                                b_, t_, j_ = [int(v) for v in d[0]]
                                ex.update({"first": [b_, t_, j_], "len_of_utt": int(c[4][b_]), "got": float(o[1][b_, t_, j_]), "ref": float(ref[1][b_, t_, j_])})
                            stat["examples"].append(ex)
                torch.cuda.synchronize()
                stat[where][hex(pat)] = bad
        out["per_call"].append(stat)
    return {"launches_per_cell": n, **out}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", default="slp_sdp", choices=["shipped", "slp_sdp", "slp_sdp_nopk", "slp_all", "slp_files"])
    ap.add_argument("--files", default="", help="slp_files: comma-separated sources built WITH the SLP / loop vectorisers (bisection)")
    ap.add_argument("--mode", default="step", choices=["step", "standalone", "dirty", "trace"])
    ap.add_argument("--n", type=int, default=300)
    ap.add_argument("--no-load", action="store_true", help="standalone: no concurrent GEMM stream")
    a = ap.parse_args()
    files = tuple(f for f in a.files.split(",") if f)
    lib = build_variant(a.variant, files)
    if lib is not None and os.environ.get("S2SVC_LIB") != lib:
        os.environ["S2SVC_LIB"] = lib                       # _lib.py reads it at import time: re-exec with it set
        os.execv(sys.executable, [sys.executable] + sys.argv)
    res = {"step": run_step, "dirty": run_dirty, "trace": run_trace}[a.mode](a.n) if a.mode != "standalone" else run_standalone(a.n, not a.no_load)
    print(json.dumps({"variant": a.variant, "files": list(files), "mode": a.mode, **res}), flush=True)


if __name__ == "__main__":
    main()
