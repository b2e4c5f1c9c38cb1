#!/usr/bin/env python3
"""Timing of the STFT -> log-mel front-end: wav batch -> normalised (B, Tmax, 80) batch.

    python tools/bench_frontend.py [--batch 32] [--seconds 3.0] [--iters 20] [--cpu]

Reports the FFT-in-LDS kernel alone (HIP events), us per utterance for the batched path (1 launch per batch; impl="gemm": 3) and for the per-utterance path (5 launches each), the
achieved fraction of the two roofs that apply -- fp32 MFMA for the DFT-as-GEMM formulation (2 * frames * 1026 * 1024 flop;
peak 157.3 TFLOP/s) and HBM for the algorithmic traffic of the whole front-end (4 B per sample in + 320 B per frame out;
8 TB/s) -- and, with --cpu, the numpy restatement on the host (one thread per numpy's defaults).  Prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--cpu", action="store_true")
    a = ap.parse_args()
    from seq2seq_vc_amd.frontend import logmelfilterbank, logmelfilterbank_batch
    sr, kw = 16000, dict(fft_size=1024, hop_size=256, num_mels=80, fmin=80, fmax=7600)
    rng = np.random.default_rng(0)
    lens = [int(sr * a.seconds * f) for f in rng.uniform(0.6, 1.0, a.batch)]
    lens[0] = int(sr * a.seconds)
    nmax = max(lens)
    x = torch.zeros(a.batch, nmax)
    for b, n in enumerate(lens):
        x[b, :n] = torch.from_numpy((rng.standard_normal(n) * 0.1).astype(np.float32))
    xd = x.cuda()
    singles = [xd[b, :n].contiguous() for b, n in enumerate(lens)]
    mean, scale = np.zeros(80, np.float32), np.ones(80, np.float32)

    def batched():
        return logmelfilterbank_batch(xd, sr, lengths=lens, mean=mean, scale=scale, **kw)

    def looped():
        return [logmelfilterbank(s, sr, mean=mean, scale=scale, **kw) for s in singles]

    def gemm_batched():
        return logmelfilterbank_batch(xd, sr, lengths=lens, mean=mean, scale=scale, impl="gemm", **kw)

    # the FFT kernel alone, device-resident arguments, HIP events around `kiters` back-to-back launches
    from seq2seq_vc_amd.frontend import stft_logmel_fft_device
    nlen_d = torch.tensor(lens, dtype=torch.int32, device="cuda")
    fr = [1 + n // 256 for n in lens]
    frames_d = torch.tensor(fr, dtype=torch.int32, device="cuda")
    mean_t, isc_t = torch.zeros(80, device="cuda"), torch.ones(80, device="cuda")
    obuf = torch.empty((a.batch, max(fr), 80), dtype=torch.float32, device="cuda")

    def kernel_only(radix8=None):
        stft_logmel_fft_device(xd, nlen_d, frames_d, max(fr), sr, 1024, 256, None, 80, 80, 7600, 1e-10, 1.0 / np.log(10.0), mean_t, isc_t, out=obuf,
                               radix8=radix8)

    def graph_time(fn, kiters=100):
        """seconds per launch: `kiters` launches replayed from one hipGraph (a Python launch loop costs ~10 us of host time per
        launch, as much as the kernel takes at small batches), HIP events around the replay"""
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                for _ in range(kiters):
                    fn()
            g.replay()
            side.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side)
            g.replay()
            e1.record(side)
            side.synchronize()
        torch.cuda.current_stream().wait_stream(side)
        return e0.elapsed_time(e1) * 1e-3 / kiters
    kiters = 100
    t_kernel = graph_time(kernel_only, kiters)
    t_kernel_r4 = graph_time(lambda: kernel_only(False), kiters)

    out = {}
    for name, fn in (("batched", batched), ("per_utterance", looped), ("batched_gemm", gemm_batched)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            fn()
        torch.cuda.synchronize()
        out[name] = (time.perf_counter() - t0) / a.iters
    frames = sum(1 + n // 256 for n in lens)
    frames_padded = a.batch * (1 + nmax // 256)
    flops = 2.0 * frames_padded * 1026 * 1024
    alg_bytes = 4.0 * sum(lens) + 320.0 * frames
    tb, tg = out["batched"], out["batched_gemm"]
    res = {"metric": "STFT->log-mel front-end", "batch": a.batch, "audio_seconds_per_utt_max": a.seconds, "frames": frames,
           "fft_kernel": {"us_per_launch": t_kernel * 1e6, "us_per_utterance": t_kernel / a.batch * 1e6, "ns_per_frame": t_kernel / frames * 1e9,
                          "timed": f"{kiters} launches of s2svc_stft_logmel_fft8 (radix-8 passes) replayed from one hipGraph, HIP events, device-resident arguments",
                          "radix4_kernel_us_per_launch": t_kernel_r4 * 1e6, "radix4_ns_per_frame": t_kernel_r4 / frames * 1e9},
           "us_per_utterance_batched": tb / a.batch * 1e6, "us_per_utterance_batched_gemm": tg / a.batch * 1e6,
           "us_per_utterance_looped": out["per_utterance"] / a.batch * 1e6,
           "ns_per_frame_batched": tb / frames * 1e9,
           "launches_per_batch": {"batched": 1, "batched_gemm": 3, "per_utterance": 5 * a.batch},
           "note": "batched / batched_gemm / per_utterance are host wall times of the Python entry points (length upload + launch overhead "
                   "included); fft_kernel is the kernel alone",
           "roofline": {"hbm_algorithmic": {"bound": "hbm", "bytes": alg_bytes, "achieved_GBs": alg_bytes / t_kernel / 1e9, "peak": 8000.0,
                                            "frac": alg_bytes / t_kernel / 1e9 / 8000.0,
                                            "note": "FFT-in-LDS kernel; 4 B/sample in + 320 B/frame out"},
                        "gemm_formulation_fp32_mfma": {"achieved_TFLOPs": flops / tg / 1e12, "peak": 157.3, "frac": flops / tg / 1e12 / 157.3,
                                                       "note": "round 2's DFT as GEMM over the padded batch (host wall time)"}},
           "realtime_factor": t_kernel / (sum(lens) / sr)}
    if a.cpu:
        from oracle import logmel as OL
        t0 = time.perf_counter()
        for b, n in enumerate(lens):
            OL.logmelfilterbank(x[b, :n].numpy(), sr, **kw)
        tc = time.perf_counter() - t0
        res["cpu_baseline"] = {"us_per_utterance": tc / a.batch * 1e6, "kind": "port", "cores": 1,
                               "sample": f"{a.batch} utterances, numpy rfft restatement, one process"}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
