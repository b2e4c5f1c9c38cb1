#!/usr/bin/env python3
"""Autoregressive decode benchmark (configuration C5) -- the stand-alone form of the "decode" sub-object
of bench.py.

    python tools/bench_decode.py [--batch 16] [--iters 5] [--dtype bf16] [--cpu-baseline]

VTN vc1 weights (seeded init), `--batch` sources of 256 frames, threshold 2.0 (never fires) so every utterance runs to
maxlen = int(63 * 6.0 / 4) = 94 steps = 376 frames (vtn.py:334,378; vtn.v1.yaml:62-64), prenet dropout 0.5 as configured.
RTF = wall time / (generated frames * hop / sr), hop 256, sr 16 kHz, batch aggregate.  Prints ONE JSON line.
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from bench import bench_decode  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--poll", type=int, default=32)
    ap.add_argument("--cpu-baseline", action="store_true")
    a = ap.parse_args()
    from seq2seq_vc_amd.ops import functional as Fn
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    Fn.set_compute_dtype(dtype)
    print(json.dumps(bench_decode(torch.device("cuda", 0), dtype, batch=a.batch, iters=a.iters, poll=a.poll, cpu=a.cpu_baseline)))


if __name__ == "__main__":
    main()
