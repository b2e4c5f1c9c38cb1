#!/usr/bin/env python3
"""Autoregressive decode benchmark (configuration C5).

    python tools/bench_decode.py [--batch 16] [--iters 5] [--dtype bf16] [--cpu-baseline]

VTN vc1 weights (seeded init), `--batch` sources of 256 frames, threshold 2.0 (never fires) so every utterance
runs to maxlen = int(63 * 6.0 / 4) = 94 steps = 376 frames (vtn.py:334,378; vtn.v1.yaml:62-64), prenet dropout
0.5 as configured.  RTF = wall time / (generated frames * hop / sr), hop 256, sr 16 kHz, batch aggregate.
Encoder + source K/V projection + 94 graph replays + postnet are all inside the timed region.
Prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from bench import VTN_VC1  # noqa: E402

ARGS = {"threshold": 2.0, "minlenratio": 0.0, "maxlenratio": 6.0}
HOP, SR = 256, 16000


def cpu_baseline(x):
    """The CPU oracle's generation loop (== the reference's schedule: recompute the prefix every step) for ONE utterance."""
    from oracle import models as OM
    from seq2seq_vc_amd.models import VTN
    torch.manual_seed(0)
    sd = {k: v.clone() for k, v in VTN(**VTN_VC1).state_dict().items()}
    t0 = time.perf_counter()
    with torch.no_grad():
        outs, _, _ = OM.vtn_inference(sd, VTN_VC1, x, drop=True, **ARGS)
    t = time.perf_counter() - t0
    return {"value": t / (outs.shape[0] * HOP / SR), "unit": "RTF", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"one 256-frame utterance -> {outs.shape[0]} frames in {t:.2f} s (fp32, reference schedule)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--poll", type=int, default=32)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--cpu-baseline", action="store_true")
    a = ap.parse_args()
    from seq2seq_vc_amd import decode as D
    from seq2seq_vc_amd.models import VTN
    from seq2seq_vc_amd.ops import functional as Fn
    from seq2seq_vc_amd.ops import kernels as K
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    Fn.set_compute_dtype(dtype)
    torch.manual_seed(0)
    K.manual_seed(1234)
    model = VTN(**VTN_VC1).to("cuda").eval()
    g = torch.Generator().manual_seed(1234)
    xs = torch.randn(a.batch, 256, 80, generator=g)
    xs_d = xs.to("cuda")
    ilens = torch.full((a.batch,), 256)

    def run():
        with torch.no_grad():
            lens = D.Mo.Lens.of(ilens, xs_d.device)
            hs, hlens = model.encoder(Fn.to_compute(xs_d), lens, exact_lens=True)
            return D.decode(model, hs, list(hlens.host), ARGS, poll=a.poll, use_graph=not a.no_graph)

    res = run()            # builds the session + captures the step graph
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        res = run()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / a.iters
    frames = sum(r[0].shape[0] for r in res)
    steps = frames // (a.batch * VTN_VC1["decoder_reduction_factor"])
    out = {"metric": "RTF (AR decode, batch aggregate)", "value": t / (frames * HOP / SR), "unit": "wall s / audio s",
           "higher_is_better": False, "n_gpus": 1, "dtype": a.dtype, "data": "synthetic",
           "ms_per_batch": t * 1e3, "us_per_step": t * 1e6 / steps, "frames_per_sec": frames / t,
           "config": {"workload": "VTN egs/arctic/vc1 AR decode (C5): encoder + 94 steps x r=4 + postnet", "batch": a.batch,
                      "T_src": 256, "steps": steps, "frames_per_utt": frames // a.batch, "hip_graph": not a.no_graph,
                      "poll": a.poll},
           "finite": bool(all(torch.isfinite(r[0]).all() for r in res))}
    if a.cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(xs[0])
    print(json.dumps(out))


if __name__ == "__main__":
    main()
