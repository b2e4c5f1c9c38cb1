"""Which lines of the package still launch ATen kernels inside a training step?

DESIGN.md section 1: the package holds no ATen compute call -- torch is the allocator, the streams, the autograd tape.  This tool
checks that claim where it matters, inside the step that gets captured: it runs bench.py's workload eagerly under torch.profiler with
Python stacks and lists every ATen operator that launched a device kernel (name of the kernel, count per step, the innermost frame of
this repository on its stack).  `--workload vtn|aasvc|tts`, `--fail` exits 1 if anything but the duration predictor's noise draw is left.

    python tools/aten_in_step.py --workload aasvc
"""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402

ALLOWED = ("distribution_elementwise", "normal_", "randn")          # the stochastic duration predictor's torch.randn (an RNG draw)


def repo_frame(stack):
    for fr in stack:
        if ROOT in fr and "/tools/aten_in_step.py" not in fr and "torch/" not in fr:
            return fr.replace(ROOT + "/", "")
    return stack[0] if stack else "?"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="aasvc", choices=["vtn", "aasvc", "tts"])
    ap.add_argument("--fail", action="store_true")
    args = ap.parse_args()
    from seq2seq_vc_amd.ops import functional as Fn
    from seq2seq_vc_amd.ops import kernels as K
    dev = torch.device("cuda", 0)
    dtype = torch.bfloat16
    Fn.set_compute_dtype(dtype)
    if args.workload == "aasvc":
        Fn.enable_side_streams(0, inline_batches=True)
    else:
        Fn.enable_side_streams(4)
    K.manual_seed(1234)
    wl = bench.Workload(args.workload, dev, dtype, {"vtn": 32, "aasvc": 16, "tts": 8}[args.workload], 1, 0)
    step, _ = bench.build_step(wl, None, 1, False, False, "fp32", False, warmup_eager=3)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    # where they come from: torch.profiler's Python stacks are empty on this stack, so the same step runs once more under a
    # TorchDispatchMode that sees every ATen call (forward: on the caller's Python stack; backward: inside the Function.backward frames)
    import traceback
    from torch.utils._python_dispatch import TorchDispatchMode
    VIEWS = ("view", "as_strided", "detach", "alias", "aten.t.", "transpose", "slice", "select", "unsqueeze", "squeeze", "expand",
             "reshape", "empty", "permute", "_unsafe_view", "unbind", "split", "chunk", "narrow", "lift_fresh", "is_", "size", "stride",
             "storage_offset", "numel", "dim", "record_stream", "_to_copy_meta", "sym_", "prim.", "_local_scalar_dense", "item")
    calls = collections.Counter()

    class Spy(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = str(func)
            if not any(v in name for v in VIEWS):
                flat = [a for a in list(args) + list((kwargs or {}).values()) if isinstance(a, torch.Tensor)]
                if any(t.is_cuda for t in flat) or "zeros" in name or "full" in name or "arange" in name or "randn" in name:
                    fr = [f for f in traceback.extract_stack() if ROOT in f.filename and "aten_in_step" not in f.filename]
                    where = f"{fr[-1].filename.replace(ROOT + '/', '')}:{fr[-1].lineno} {fr[-1].name}" if fr else "(autograd engine)"
                    shp = ",".join("x".join(map(str, t.shape)) or "()" for t in flat[:3])
                    calls[(name, where, shp)] += 1
            return func(*args, **(kwargs or {}))

    with Spy():
        step()
        torch.cuda.synchronize()
    print(f"[aten_in_step] ATen calls on device tensors during one eager step, by call site ({sum(calls.values())} calls):")
    for (name, where, shp), c in sorted(calls.items(), key=lambda t: (t[0][1], t[0][0])):
        print(f"  {c:3d} x {name:34s} {shp:40s} {where}")
    seen = collections.Counter()
    total_kernels = 0
    for ev in prof.events():
        if ev.device_type != torch.autograd.DeviceType.CPU:
            continue
        kerns = [k for k in (ev.kernels or [])]
        if not kerns:
            continue
        total_kernels += len(kerns)
        for k in kerns:
            if "at::native" in k.name or "at6native" in k.name or "rocclr" in k.name:
                seen[(ev.name, k.name.split("<")[0][:70], repo_frame(ev.stack or []))] += 1
    n_aten = sum(seen.values())
    left = [(key, c) for key, c in seen.items() if not any(a in key[0] or a in key[1] for a in ALLOWED)]
    print(f"[aten_in_step] workload {args.workload}: {n_aten} ATen kernel launches in one eager step "
          f"({total_kernels} launches attributed to operators); {sum(c for _, c in left)} besides the RNG draw")
    for (op, kern, frame), c in sorted(seen.items(), key=lambda t: (-t[1], t[0])):
        print(f"  {c:3d} x {op:32s} {kern:72s} {frame}")
    if args.fail and left:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
