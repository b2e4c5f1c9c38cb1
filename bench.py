#!/usr/bin/env python3
"""Headline benchmark of the MI355X-native seq2seq-vc hot path.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], SURVEY.md section 8d "C2"): one optimiser step of VTN
(egs/arctic/vc1/conf/vtn.v1.yaml: 6+6 layers, d=384, r=4, postnet 5x256xk5; 30.48 M params) on a
synthetic ARCTIC-shaped batch of 32 utterance pairs PER GPU (T_src = T_tgt padded to 256, 80-dim mel),
bf16 compute with fp32 master weights: forward + Seq2SeqLoss + backward + grad-clip + Adam + WarmupLR.
Metric: mel-frames/sec = sum of valid target frames consumed per step over all ranks / step wall time.

One process per GPU (torch.distributed, backend nccl == RCCL); data parallel = mean all-reduce of the
flat fp32 gradient buffer.  Rank 0 prints ONE JSON line, extended with
  "roofline":     the dominant kernel (MFMA GEMM) timed live with HIP events against the bf16 MFMA peak
  "cpu_baseline": the CPU oracle (fp32 restatement of the reference) timed on this box's host cores
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

VTN_VC1 = dict(idim=80, odim=80, dprenet_layers=2, dprenet_units=256, adim=384, aheads=4, elayers=6, eunits=1536, dlayers=6,
               dunits=1536, postnet_layers=5, postnet_filts=5, postnet_chans=256, use_batch_norm=True,
               encoder_normalize_before=True, decoder_normalize_before=False, encoder_concat_after=False,
               decoder_concat_after=False, decoder_reduction_factor=4)
FWD_BWD_GFLOP = 715.9   # BASELINE.md section 2 (matmul/conv FLOPs of one fwd+bwd at B=32, T=256)
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: ~2.5 PFLOP/s dense bf16
F32_MFMA_PEAK_TFLOPS = 157.3


def canonical_batch(B_total, T=256, idim=80, odim=80, seed=1234):
    """draw order ilens, olens, xs, ys; lens in [128,256], element 0 forced to 256."""
    g = torch.Generator().manual_seed(seed)
    ilens = torch.randint(128, T + 1, (B_total,), generator=g)
    ilens[0] = T
    olens = torch.randint(128, T + 1, (B_total,), generator=g)
    olens[0] = T
    xs = torch.randn(B_total, T, idim, generator=g)
    ys = torch.randn(B_total, T, odim, generator=g)
    ar = torch.arange(T)[None, :]
    xs[(ar >= ilens[:, None])] = 0.0
    ys[(ar >= olens[:, None])] = 0.0
    labels = (ar >= (olens[:, None] - 1)).float()
    return xs, ilens, ys, labels, olens


def cpu_baseline(batch, steps=2):
    """The CPU oracle (oracle/models.py, proven equal to the reference by tests/golden) on the host cores:
    fwd + loss + bwd + clip + Adam at the same shapes, fp32, train-mode dropout on."""
    from oracle import models as OM
    from seq2seq_vc_amd.models import VTN
    xs, ilens, ys, labels, olens = batch
    torch.manual_seed(0)
    ref = VTN(**VTN_VC1)
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    names = [k for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k]
    for k in names:
        sd[k].requires_grad_(True)
    params = [sd[k] for k in names]
    state = [(torch.zeros_like(p), torch.zeros_like(p)) for p in params]
    times = []
    for it in range(steps + 1):
        t0 = time.perf_counter()
        o = OM.vtn_forward(sd, VTN_VC1, xs, ilens, ys, labels, olens, training=True, drop=True)
        l1, bce = OM.seq2seq_loss(o[0], o[1], o[2], o[3], o[4], o[5])
        grads = torch.autograd.grad(l1 + bce, params, allow_unused=True)
        grads = [g if g is not None else torch.zeros_like(p) for g, p in zip(grads, params)]
        with torch.no_grad():
            OM.adam_step(params, grads, state, OM.warmup_lr(8e-5, it + 1), it + 1)
        times.append(time.perf_counter() - t0)
    t = sum(times[1:]) / max(1, len(times) - 1)
    return {"value": float(olens.sum()) / t, "unit": "mel-frames/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{steps} optimiser steps (after 1 warm-up) of the same VTN-vc1 B=32 batch, fp32, {t:.2f} s/step",
            "ms_per_step": t * 1e3}


def dominant_kernel_roofline(dtype, iters=100):
    """Times the FLOP-heaviest single launch of the step -- the implicit-GEMM 3x3 stride-2 Conv2d of the
    encoder front-end (M = 32*63*19, N = 384, K = 9*384; subsampling.py:60) -- with HIP events on the
    stream it is launched on, and rates it against the MFMA peak of its dtype."""
    from seq2seq_vc_amd.ops import kernels as K
    B, T1, F1, C, O = 32, 127, 39, 384, 384
    T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
    M, N, Kd = B * T2 * F2, O, 9 * C
    x = torch.randn(B, T1, F1, C, device="cuda").to(dtype)
    w = (torch.randn(O, 9 * C, device="cuda") * 0.02).to(dtype)
    b = torch.zeros(O, device="cuda")
    y = torch.empty(B, T2, F2, O, dtype=dtype, device="cuda")

    def launch():
        K.gemm(K.operand(x, C, mode=K.CONV2D_S2, C=C, T1=T1, F1=F1, T2=T2, F2=F2), K.operand(w, 9 * C), M, N, Kd, y,
               in_dtype=dtype, bias=b, act="relu")
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    # the `iters` launches are replayed from one hipGraph: HIP events around a Python launch loop would time the host's
    # ~10 us per ctypes launch between 140 us kernels, not the kernel (rocprofv3's per-dispatch average is the cross-check)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    timed = "hipGraph replay"
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                for _ in range(iters):
                    launch()
            g.replay()
            side.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side)
            g.replay()
            e1.record(side)
            side.synchronize()
    except Exception as e:  # noqa: BLE001 -- report it and time a plain launch loop instead
        print(f"[bench] roofline graph capture failed ({type(e).__name__}: {e}); timing a Python launch loop", file=sys.stderr)
        timed = "python launch loop"
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side)
            for _ in range(iters):
                launch()
            e1.record(side)
            side.synchronize()
    torch.cuda.current_stream().wait_stream(side)
    ms = e0.elapsed_time(e1) / iters
    flops = 2.0 * M * N * Kd
    peak = BF16_MFMA_PEAK_TFLOPS if dtype == torch.bfloat16 else F32_MFMA_PEAK_TFLOPS
    ach = flops / (ms * 1e-3) / 1e12
    kernel = ("gemm_glds_kernel<128,128,KC_CONV2D,KC_DENSE> (bf16, LDS-DMA staged)" if dtype == torch.bfloat16
              else "gemm_fast_kernel<float,128,128,32> (exact-fp32 MFMA)")
    # HBM-side bytes per launch come from rocprofv3 PMC passes of exactly this loop (they cannot be read from inside the
    # process): profiles/roofline_pmc.json holds (2*FETCH_SIZE + WRITE_SIZE)*1024 as MI355X_MICROARCH.md prescribes.
    traffic, alg_bytes = None, float((x.numel() + w.numel() + y.numel()) * x.element_size())
    pmc = os.path.join(ROOT, "profiles", "roofline_pmc.json")
    if dtype == torch.bfloat16 and os.path.exists(pmc):
        with open(pmc) as f:
            traffic = json.load(f).get("hbm_bytes_per_launch")
    return {"bound": "mfma", "kernel": kernel + " conv2d-3x3-s2 implicit GEMM M=%d N=%d K=%d" % (M, N, Kd), "achieved": ach,
            "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic, "algorithmic_bytes": alg_bytes,
            "avg_launch_us": ms * 1e3, "flops_per_launch": flops, "timed": f"{iters} launches, {timed}, HIP events"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32, help="utterance pairs per GPU")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-graph", action="store_true", help="launch kernels eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--roofline-only", action="store_true", help="only run the dominant-kernel loop (for rocprofv3)")
    ap.add_argument("--split-backward", action="store_true",
                    help="N = 1: run the two-graph step of the data-parallel path (autograd cut at the encoder output) without collectives")
    ap.add_argument("--inline-batches", action="store_true", help="with --side-streams 0: queue the gradient work and run it in batches on its own stream")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the N > 1 code path (RCCL process group, split backward, overlapped all-reduce) at world size 1")
    ap.add_argument("--side-streams", type=int, default=4, help="HIP side streams for parameter-gradient kernels (0 = off)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    dp = world > 1 or args.force_dist          # the data-parallel code path (also reachable at world size 1 for testing)
    if dp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        dist.init_process_group("nccl", device_id=dev)

    from seq2seq_vc_amd import losses as L
    from seq2seq_vc_amd.distributed import allreduce_end, allreduce_mean_, allreduce_sum_begin
    from seq2seq_vc_amd.models import VTN
    from seq2seq_vc_amd.ops import functional as Fn
    from seq2seq_vc_amd.ops import kernels as K
    from seq2seq_vc_amd.optim import FlatAdam

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    if args.roofline_only:
        print(json.dumps(dominant_kernel_roofline(dtype, iters=200)))
        return
    Fn.set_compute_dtype(dtype)
    Fn.enable_side_streams(args.side_streams, inline_batches=args.inline_batches)
    K.manual_seed(1234 + rank)

    B = args.batch
    xs, ilens, ys, labels, olens = canonical_batch(B * world)
    sl = slice(rank * B, (rank + 1) * B)
    xs, ilens, ys, labels, olens = xs[sl], ilens[sl], ys[sl], labels[sl], olens[sl]
    if int(ilens.max()) < 256:  # keep the padded shape canonical on every rank
        ilens[0] = 256
    if int(olens.max()) < 256:
        olens[0] = 256
    frames_local = float(olens.sum())
    cpu_batch = (xs.clone(), ilens.clone(), ys.clone(), labels.clone(), olens.clone())
    xs_d, ys_d, labels_d = xs.to(dev), ys.to(dev), labels.to(dev)

    torch.manual_seed(0)  # identical initial weights on every rank (stands in for the DDP broadcast)
    model = VTN(**VTN_VC1).to(dev)
    model.train()
    crit = L.Seq2SeqLoss(bce_pos_weight=10.0)
    opt = FlatAdam(model, lr=8e-5, grad_norm=1.0, warmup_steps=4000, bf16_shadow=(dtype == torch.bfloat16))
    loss_buf = torch.zeros(2, device=dev)

    def fwd_bwd():
        K.reset_op_counter()
        K.advance_seed(dev)
        opt.zero_grad()
        after, before, logits, ys_, labels_, olens_, _ = model(xs_d, ilens, ys_d, labels_d, olens)
        l1, bce = crit(after, before, logits, ys_, labels_, olens_)
        (l1 + bce).backward()
        loss_buf[0].copy_(l1.detach())       # (before the join: the copies run while the side streams finish)
        loss_buf[1].copy_(bce.detach())
        Fn.side_join()

    # -- data-parallel overlap: the autograd graph is cut at the encoder output.  Graph 1 = forward + loss + the decoder-
    # side backward; its gradients (one contiguous range of the flat buffer) start their all-reduce while graph 2, the
    # encoder's backward, runs.  The loss carries the 1/world of the mean, so the collectives are plain sums.
    enc_range = opt.param_range(model.encoder)
    split = (dp or args.split_backward) and enc_range is not None and enc_range[0] == 0
    gscale = 1.0 / world
    cut = {}

    def fwd_bwd_decoder():
        K.reset_op_counter()
        K.advance_seed(dev)
        opt.zero_grad()
        cut.clear()
        after, before, logits, ys_, labels_, olens_, _ = model(xs_d, ilens, ys_d, labels_d, olens, _memory_cut=cut)
        l1, bce = crit(after, before, logits, ys_, labels_, olens_)
        ((l1 + bce) * gscale if dp else (l1 + bce)).backward()
        loss_buf[0].copy_(l1.detach())
        loss_buf[1].copy_(bce.detach())
        Fn.side_join()

    def bwd_encoder():
        cut["encoder_out"].backward(cut["decoder_in"].grad)
        Fn.side_join()

    def reduce_begin(part):      # part 0: everything behind the encoder's parameters, part 1: the encoder's
        lo, hi = (enc_range[1], opt.numel) if part == 0 else enc_range
        return allreduce_sum_begin(opt.flat_g[lo:hi], dist, world, force=args.force_dist)

    def step_eager():
        if split:
            fwd_bwd_decoder()
            h = reduce_begin(0)
            bwd_encoder()
            h += reduce_begin(1)
            allreduce_end(h)
        else:
            fwd_bwd()
            if dp:
                allreduce_mean_(opt.flat_g, dist, world, force=args.force_dist)
        opt.step()

    # warm-up (eager, on a side stream so that a later capture sees a quiet default stream)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(max(2, args.warmup if args.no_graph else 2)):
            step_eager()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()

    use_graph = not args.no_graph
    g_fb = g_fb2 = g_opt = None
    if use_graph:
        try:
            g_fb = torch.cuda.CUDAGraph()
            # thread_local: RCCL's watchdog thread polls events while this thread captures (N > 1)
            with torch.cuda.graph(g_fb, capture_error_mode="thread_local"):
                fwd_bwd_decoder() if split else fwd_bwd()
            if split:
                g_fb2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_fb2, pool=g_fb.pool(), capture_error_mode="thread_local"):
                    bwd_encoder()
            g_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_opt, capture_error_mode="thread_local"):
                opt.step()
        except Exception as e:  # noqa: BLE001 -- report and fall back to eager launches, loudly
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            use_graph = False
            torch.cuda.synchronize()

    def step():
        if not use_graph:
            step_eager()
        elif split:
            g_fb.replay()
            h = reduce_begin(0)          # decoder / postnet gradients travel ...
            g_fb2.replay()               # ... while the encoder's backward pass runs
            h += reduce_begin(1)
            allreduce_end(h)
            g_opt.replay()
        else:
            g_fb.replay()
            if dp:
                allreduce_mean_(opt.flat_g, dist, world, force=args.force_dist)
            g_opt.replay()

    for _ in range(args.warmup):
        step()

    def barrier():
        if dp:
            dist.barrier()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if dp:
        tt = torch.tensor([dt], device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        ft = torch.tensor([frames_local], device=dev)
        dist.all_reduce(ft)
        frames = float(ft.item())
    else:
        frames = frames_local
    ms = dt / args.steps * 1e3
    losses = loss_buf.tolist()
    stats = opt.last_stats()
    if not all(map(lambda v: v == v and abs(v) < 1e6, losses)):
        raise SystemExit(f"bench: non-finite loss {losses}")

    if rank == 0:
        out = {
            "metric": "mel-frames/sec (train)", "value": frames / (dt / args.steps), "unit": "mel-frames/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "VTN egs/arctic/vc1 (vtn.v1.yaml) training step: fwd+Seq2SeqLoss+bwd+clip+Adam+WarmupLR",
                       "batch_per_gpu": B, "global_batch": B * world, "T_src": 256, "T_tgt": 256, "mel_dim": 80,
                       "params_M": 30.48, "parallelism": f"dp{world}", "hip_graph": bool(use_graph), "split_backward": bool(split),
                       "valid_target_frames_per_step": frames},
            "final_losses": {"l1": losses[0], "bce": losses[1], "grad_norm": stats["grad_norm"], "opt_steps": stats["step"]},
            "step_mfma": {"gflop_per_step_per_gpu": FWD_BWD_GFLOP,
                          "achieved_tflops_per_gpu": FWD_BWD_GFLOP / ms,
                          "frac_of_bf16_peak": FWD_BWD_GFLOP / ms / BF16_MFMA_PEAK_TFLOPS},
        }
        out["roofline"] = dominant_kernel_roofline(dtype)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cpu_batch)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        # RCCL writes its version banner to the C-level stdout; flush that buffer first so the JSON line stays the last line
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(out), flush=True)
    if dp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
