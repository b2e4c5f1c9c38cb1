#!/usr/bin/env python3
"""AAS-VC training-step benchmark (configuration C3; one rank's share) -- the stand-alone form of the
"aasvc" sub-object of bench.py (for rocprofv3 runs and A/B timing).

    python tools/bench_aasvc.py [--batch 16] [--steps 20] [--warmup 3] [--dtype bf16] [--cpu-baseline]

Model: egs/arctic/vc2/conf/aas_vc.melmelmel.v1.yaml (Conformer 4+4, d=384 -> 1536, stochastic duration predictor), batch 16
utterance pairs, T_src = T_tgt padded to 256, bf16 compute with fp32 master weights; one step = forward (encoder, alignment
module + MAS, duration predictor, Gaussian upsampling, decoder, postnet) + L1 + lambda_align*(forward-sum + bin) + duration NLL
+ backward + clip + Adam + WarmupLR (trainers/aas_vc.py:56-164), replayed from hipGraphs.  Prints ONE JSON line.
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from bench import AASVC_VC2, bench_aasvc_single  # noqa: E402,F401  (AASVC_VC2 is imported from here by the tests)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--cpu-baseline", action="store_true")
    a = ap.parse_args()
    from seq2seq_vc_amd.ops import functional as Fn
    from seq2seq_vc_amd.ops import kernels as K
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    Fn.set_compute_dtype(dtype)
    K.manual_seed(1234)
    print(json.dumps(bench_aasvc_single(torch.device("cuda", 0), dtype, steps=a.steps, warmup=a.warmup, cpu=a.cpu_baseline,
                                        batch=a.batch)))


if __name__ == "__main__":
    main()
