#!/usr/bin/env python3
"""Step time of the PRODUCT trainers (seq2seq_vc_amd.trainers) on synthetic batches of varying lengths, eager against
config["hip_graph"] (captured steps, trainers/graphed.py).  Unlike bench.py the batches arrive as host tensors from a
"collater" every step (the H2D copy is inside the timed region) and every batch has its own lengths.

    python tools/bench_trainer.py [--workload vtn|aasvc] [--steps 60] [--dtype bf16]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from seq2seq_vc_amd import losses as L  # noqa: E402
from seq2seq_vc_amd import models as M  # noqa: E402
from seq2seq_vc_amd import trainers as T  # noqa: E402
from seq2seq_vc_amd.ops import functional as Fn  # noqa: E402
from seq2seq_vc_amd.ops import kernels as K  # noqa: E402
from seq2seq_vc_amd.optim import FlatAdam  # noqa: E402


def make_batches(workload, n, B, seed=7):
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        ilens = torch.randint(128, 257, (B,), generator=g)
        olens = torch.randint(128, 257, (B,), generator=g)
        ilens[0] = olens[0] = 256                       # every batch pads to 256 frames (one graph); the other lengths vary
        xs, ys = torch.randn(B, 256, 80, generator=g), torch.randn(B, 256, 80, generator=g)
        for b in range(B):
            xs[b, ilens[b]:] = 0
            ys[b, olens[b]:] = 0
        bt = {"xs": xs, "ilens": ilens, "ys": ys, "olens": olens}
        if workload == "vtn":
            labels = torch.zeros(B, 256)
            for b in range(B):
                labels[b, olens[b] - 1:] = 1.0
            bt["labels"] = labels
        else:
            bt["dp_inputs"], bt["dplens"] = xs, ilens
        out.append(bt)
    return out


def run(workload, mode, data, dtype, dev, accum=1):
    Fn.set_compute_dtype(dtype)
    K.manual_seed(1234)
    torch.manual_seed(0)
    conf = {"train_max_steps": len(data), "log_interval_steps": 10 ** 9, "save_interval_steps": 10 ** 9, "grad_norm": 1.0, "outdir": "."}
    conf["hip_graph"] = bool(mode)          # (captured steps are the trainers' default; False = the eager step)
    if accum > 1:                           # micro-steps: the recipes' batch 2 x accumulation 8 (egs/hificaptain_jp/vc2/README.md:11)
        conf["gradient_accumulate_steps"] = accum
        conf["train_max_steps"] = len(data) // accum
    if workload == "vtn":
        model = M.VTN(**bench.VTN_VC1).to(dev).train()
        opt = FlatAdam(model, lr=8e-5, grad_norm=1.0, warmup_steps=4000, bf16_shadow=(dtype == torch.bfloat16))
        tr = T.ARVCTrainer(0, 0, {"train": data}, None, model, None, {"Seq2SeqLoss": L.Seq2SeqLoss(10.0)}, opt, None, conf, device=dev)
    else:
        model = M.AASVC(**bench.AASVC_VC2).to(dev).train()
        opt = FlatAdam(model, lr=8e-5, grad_norm=1.0, warmup_steps=4000, bf16_shadow=(dtype == torch.bfloat16))
        conf.update({"criterions": ["L1Loss", "ForwardSumLoss", "StochasticDurationPredictorLoss"], "lambda_align": 2.0,
                     "dp_train_start_steps": 0})
        tr = T.AASVCTrainer(0, 0, {"train": data}, None, model, None, {"L1Loss": L.L1Loss(), "ForwardSumLoss": L.ForwardSumLoss()},
                            opt, None, conf, device=dev)
    warm = 6 if accum == 1 else 3 * accum       # every role of a micro-step is seen (eager), captured and replayed once
    it = iter(data)
    for _ in range(warm):
        tr._step(next(it))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    for batch in it:
        tr._step(batch)
        n += 1
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    frames = sum(float(b["olens"].sum()) for b in data[warm:]) / n
    out = {"mode": "hip_graph" if mode else "eager", "ms_per_step": dt * 1e3, "mel_frames_per_s": frames / dt, "steps": n}
    if accum > 1:
        out = {"mode": out["mode"], "ms_per_micro_step": dt * 1e3, "ms_per_optimizer_step": dt * 1e3 * accum, "mel_frames_per_s": frames / dt,
               "micro_steps": n, "gradient_accumulate_steps": accum}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="vtn")
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--accum", type=int, default=1, help="gradient_accumulate_steps (AAS-VC trainer: captured micro-steps)")
    ap.add_argument("--batch", type=int, default=None)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    B = a.batch or (32 if a.workload == "vtn" else 16)
    n = a.steps + 6 if a.accum == 1 else (a.steps // a.accum + 3) * a.accum
    data = make_batches(a.workload, n, B)
    res = [run(a.workload, mode, data, dtype, dev, accum=a.accum) for mode in (False, True)]
    print(json.dumps({"workload": a.workload, "batch": B, "dtype": a.dtype, "data": "synthetic, lengths vary per batch, host batches (H2D inside)",
                      "results": res}))


if __name__ == "__main__":
    main()
