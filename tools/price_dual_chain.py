#!/usr/bin/env python3
"""Pricing experiment: does the VTN step gain from running its forward / data-gradient chain as TWO independent
chains over utterance ranges [0,16) and [16,32) on two streams of the one captured graph?  Timing only: the halves run the whole
model on their slice (so BatchNorm statistics are per half -- not a parity configuration), the comparison is
    one B = 32 pass   vs   two concurrent B = 16 passes
with and without the parameter-gradient work (the chain alone), both without the decoder-head branch (one auxiliary stream exists).
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def time_graph(fn, reps=60):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    ms = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1) / reps)
    return sorted(ms)[1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chains", type=int, nargs="+", default=[1, 2, 4])
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    from seq2seq_vc_amd.models import vtn as vtn_mod
    from seq2seq_vc_amd.ops import functional as Fn
    from seq2seq_vc_amd.ops import kernels as K
    dtype = torch.bfloat16
    Fn.set_compute_dtype(dtype)
    K.manual_seed(1234)
    wl = bench.Workload("vtn", dev, dtype, 32, 1, 0)
    model, opt = wl.model, wl.opt
    real_side_run = Fn._side_run
    results = {}

    def halves(k):
        n = 32 // k
        out = []
        for i in range(k):
            sl = slice(i * n, (i + 1) * n)
            il, ol = wl.ilens[sl].clone(), wl.olens[sl].clone()
            il[0], ol[0] = 256, 256
            out.append((wl.xs[sl].contiguous(), il, wl.ys[sl].contiguous(), wl.labels[sl].contiguous(), ol))
        return out

    def make(k, streams):
        parts = halves(k)

        def one(p):
            after, before, logits, ys_, labels_, olens_, _ = model(*p)
            l1, bce = wl.crit(after, before, logits, ys_, labels_, olens_)
            return l1 + bce

        def fn():
            K.reset_op_counter()
            K.advance_seed(dev)
            opt.begin_step()
            main = torch.cuda.current_stream()
            if k == 1:
                total = one(parts[0])
            else:
                losses = []
                for p, st in zip(parts, streams):
                    st.wait_stream(main)
                    with torch.cuda.stream(st):
                        losses.append(one(p))
                for st in streams[:k]:
                    main.wait_stream(st)
                total = losses[0]
                for l in losses[1:]:
                    total = total + l
            opt.join_prologue()
            total.backward()
            for st in streams[:k] if k > 1 else []:
                main.wait_stream(st)
            Fn.side_join()
        return fn

    for head in (True, False):
        vtn_mod._HEAD_START = head
        for wgrad in (True, False):
            Fn._side_run = real_side_run if wgrad else (lambda fn, keep=(), solo=False: None)
            for k in args.chains:
                if head and k > 1:
                    continue            # one auxiliary stream: the decoder-head branch is a single-chain feature
                Fn.enable_side_streams(4 if wgrad else 0)
                streams = [Fn.distinct_stream() for _ in range(k)] if k > 1 else []
                try:
                    ms = time_graph(make(k, streams))
                except Exception as e:  # noqa: BLE001
                    ms = f"{type(e).__name__}: {e}"
                key = f"chains={k} head_start={int(head)} wgrad={int(wgrad)}"
                results[key] = ms
                print(key, ms, flush=True)
    Fn._side_run = real_side_run
    print(json.dumps(results))


if __name__ == "__main__":
    main()
