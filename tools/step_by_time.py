"""profiles/step_by_time.json: for each training workload, the kernel FAMILY the step spends most of its GPU time in, its share of
the step's kernel time and its MFMA-pipe busy fraction -- read by bench.py into `roofline.by_time` (the FLOP-heaviest launch the
headline roofline times is a few percent of the VTN step; this names where the time goes).

    python tools/step_by_time.py <round prefix, e.g. r04>

Inputs (committed summaries, made by tools/collect_profiles.sh on a GPU box):
    profiles/<r>_{vtn,aasvc}_train_bf16_timeline.txt       ONE step's launch sequence of a rocprofv3 --kernel-trace run of bench.py
                                                           (tools/rocpd_timeline.py; falls back to the per-kernel table
                                                           <r>_..._kernel_stats.txt, which also holds the dominant-kernel loop)
    profiles/<r>_step_mfma_busy.txt                        SQ_VALU_MFMA_BUSY_CYCLES pass over whole steps (tools/step_mfma_busy.py)
"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout
        return out.split("\n")[: len(names)]
    except (OSError, subprocess.CalledProcessError):
        return names


def family(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z_0-9:]+)", name)
    return m.group(1) if m else name


def kernel_stats(path):
    rows, total = [], None
    for line in open(path):
        if line.startswith("# kernels:"):
            total = float(re.search(r"total GPU kernel time ([0-9.]+) ms", line).group(1)) * 1e3
        p = line.split(None, 10)
        if len(p) == 11 and re.match(r"^[0-9.]+$", p[0]):
            rows.append((p[10].strip(), int(p[1]), float(p[2])))
    names = demangle([r[0] for r in rows])
    fam = {}
    for (_, calls, us), n in zip(rows, names):
        f = fam.setdefault(family(n), [0, 0.0])
        f[0] += calls
        f[1] += us
    return fam, total


def timeline_stats(path):
    """family -> [launches, us] of ONE step, from the launch sequence tools/rocpd_timeline.py wrote (the kernel-stats table of the same
    profiling run also holds the dominant-kernel loop and the sub-benchmarks of bench.py: the timeline is the step and nothing else)"""
    rows = []
    for line in open(path):
        m = re.match(r"\s*[0-9.]+ \S+\s+dur\s+([0-9.]+) gap\s+-?[0-9.]+ idle\s+[0-9.]+\s+(\S.*)", line)
        if m:
            rows.append((m.group(2).strip(), float(m.group(1))))
    names = demangle([n if n.startswith("_Z") else "_ZN12_GLOBAL__N_1" + n for n, _ in rows])
    fam, total = {}, 0.0
    for (_, us), n in zip(rows, names):
        f = fam.setdefault(family(n), [0, 0.0])
        f[0] += 1
        f[1] += us
        total += us
    return fam, total


def mfma_busy(path, section):
    """launch-time-weighted MFMA busy per family from the '# <section>' block of <r>_step_mfma_busy.txt"""
    fam, on = {}, False
    for line in open(path):
        if line.startswith("# "):
            on = line.strip() == f"# {section}"
            continue
        m = re.match(r"\s*(\d+) x\s+([0-9.]+) us\s+mfma_busy ([0-9.]+)\s+wait/wave ([0-9.]+)\s+(.*)", line)
        if on and m:
            n, us, busy, wait, name = int(m.group(1)), float(m.group(2)), float(m.group(3)), float(m.group(4)), m.group(5)
            f = fam.setdefault(family(name), [0.0, 0.0, 0.0])
            f[0] += n * us
            f[1] += n * us * busy
            f[2] += n * us * wait
    return {k: (v[1] / v[0], v[2] / v[0]) for k, v in fam.items() if v[0] > 0}


def main():
    r = sys.argv[1]
    out = {}
    for wl in ("vtn", "aasvc"):
        ks = os.path.join(ROOT, "profiles", f"{r}_{wl}_train_bf16_kernel_stats.txt")
        if not os.path.exists(ks):
            continue
        tl = os.path.join(ROOT, "profiles", f"{r}_{wl}_train_bf16_timeline.txt")
        src = ks
        if os.path.exists(tl):
            fam, total = timeline_stats(tl)
            src = tl
        else:
            fam, total = kernel_stats(ks)
        busy_path = os.path.join(ROOT, "profiles", f"{r}_step_mfma_busy.txt")
        busy = mfma_busy(busy_path, wl) if os.path.exists(busy_path) else {}
        top = sorted(fam.items(), key=lambda kv: -kv[1][1])
        name, (calls, us) = top[0]
        b = busy.get(name)
        out[wl] = {"family": name, "share_of_kernel_time": us / total, "launches_per_step" if src != ks else "launches_in_profile": calls,
                   "mfma_busy": b[0] if b else None, "wait_per_wave": b[1] if b else None,
                   "source": f"profiles/{os.path.basename(src)} + profiles/{r}_step_mfma_busy.txt",
                   "top5": [{"family": n, "share_of_kernel_time": u / total, "mfma_busy": (busy.get(n) or (None,))[0]} for n, (c, u) in top[:5]]}
    path = os.path.join(ROOT, "profiles", "step_by_time.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
