"""MFMA-pipe busy fraction per GEMM-shaped kernel over whole training steps, from a rocprofv3 counter pass (rocpd database):

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace -d <dir> -- python bench.py --no-graph ...
    python tools/step_mfma_busy.py <section name> <results.db>

One line per kernel: launches x average duration, mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x avg us x 2400 cycles/us),
wait/wave = SQ_WAIT_ANY / SQ_WAVE_CYCLES (the format tools/step_by_time.py reads)."""
import sqlite3
import subprocess
import sys

KEEP = ("gemm", "attn", "mfma", "conv2d")


def main():
    section, db = sys.argv[1], sys.argv[2]
    c = sqlite3.connect(db)
    q = "select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection group by kernel_name, counter_name"
    per = {}
    for name, ctr, n, avg, dur in c.execute(q):
        d = per.setdefault(name, {"n": n, "us": dur / 1e3})
        d[ctr] = avg
    names = list(per)
    try:
        dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")[: len(names)]
    except (OSError, subprocess.CalledProcessError):
        dem = names
    print(f"# {section}")
    rows = []
    for name, pretty in zip(names, dem):
        d = per[name]
        if not any(k in pretty for k in KEEP) or "SQ_VALU_MFMA_BUSY_CYCLES" not in d:
            continue
        busy = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * d["us"] * 2400.0)
        wait = d.get("SQ_WAIT_ANY", 0.0) / max(d.get("SQ_WAVE_CYCLES", 1.0), 1.0)
        rows.append((d["n"] * d["us"], d["n"], d["us"], busy, wait, pretty.replace("void ", "").replace("(anonymous namespace)::", "")))
    for _, n, us, busy, wait, pretty in sorted(rows, reverse=True):
        print(f"{n:5d} x {us:7.2f} us  mfma_busy {busy:.3f}  wait/wave {wait:.2f}  {pretty[:120]}")


if __name__ == "__main__":
    main()
