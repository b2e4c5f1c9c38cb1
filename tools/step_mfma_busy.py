"""MFMA-pipe busy fraction per GEMM-shaped kernel over whole training steps, from a rocprofv3 counter pass (rocpd database):

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace -d <dir> -- python bench.py --no-graph ...
    python tools/step_mfma_busy.py <section name> <results.db> [by-grid]

One line per kernel: launches x average duration, mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x avg us x 2400 cycles/us),
wait/wave = SQ_WAIT_ANY / SQ_WAVE_CYCLES (the format tools/step_by_time.py reads)."""
import sqlite3
import subprocess
import sys

KEEP = ("gemm", "attn", "mfma", "conv2d")


def main():
    section, db = sys.argv[1], sys.argv[2]
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    # with "by-grid" as third argument: one line per (kernel, grid) -- the launch shapes of a kernel family apart
    grid = [g for g in ("grid_size_x", "grid_size_y", "grid_size_z") if g in cols] if len(sys.argv) > 3 and sys.argv[3] == "by-grid" else []
    wg = [g for g in ("workgroup_size_x", "workgroup_size_y", "workgroup_size_z") if g in cols] if grid else []
    key = ", ".join(["kernel_name"] + grid + wg)
    q = f"select {key}, counter_name, count(*), avg(value), avg(duration) from counters_collection group by {key}, counter_name"
    per = {}
    for row in c.execute(q):
        name, ctr, n, avg, dur = row[0], row[-4], row[-3], row[-2], row[-1]
        shape = row[1:-4]
        tag = ""
        if grid:
            g, w = shape[:len(grid)], shape[len(grid):]
            tag = " grid " + "x".join(str(int(a) // max(int(b), 1)) for a, b in zip(g, w if len(w) == len(g) else [1] * len(g)))
        d = per.setdefault((name, tag), {"n": n, "us": dur / 1e3})
        d[ctr] = avg
    names = [k[0] for k in per]
    tags = [k[1] for k in per]
    try:
        dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")[: len(names)]
    except (OSError, subprocess.CalledProcessError):
        dem = names
    print(f"# {section}")
    rows = []
    for name, tag, pretty in zip(names, tags, dem):
        d = per[(name, tag)]
        pretty = pretty + tag
        if not any(k in pretty for k in KEEP) or "SQ_VALU_MFMA_BUSY_CYCLES" not in d:
            continue
        busy = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * d["us"] * 2400.0)
        wait = d.get("SQ_WAIT_ANY", 0.0) / max(d.get("SQ_WAVE_CYCLES", 1.0), 1.0)
        pretty = pretty.replace("void ", "").replace("(anonymous namespace)::", "")
        if tag:
            pretty = pretty.split("(")[0] + tag
        rows.append((d["n"] * d["us"], d["n"], d["us"], busy, wait, pretty))
    for _, n, us, busy, wait, pretty in sorted(rows, reverse=True):
        print(f"{n:5d} x {us:7.2f} us  mfma_busy {busy:.3f}  wait/wave {wait:.2f}  {pretty[:120]}")


if __name__ == "__main__":
    main()
