"""Summarise a rocprofv3 (rocpd SQLite) kernel trace into a per-kernel table, the same numbers
`rocprofv3 --kernel-trace --stats` reports: calls, total / average / min / max duration, share of GPU time.

    python tools/rocpd_stats.py gpurun_out/prof1/vtn_results.db [steps] > profiles/r01_xxx.txt
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else None
    c = sqlite3.connect(db)
    t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [x for x in t if "kernel_dispatch" in x][0]
    ks = [x for x in t if "kernel_symbol" in x][0]
    q = (f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), "
         f"max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.sgpr_count), max(d.group_segment_size) "
         f"from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc")
    rows = list(c.execute(q))
    tot = sum(r[2] for r in rows)
    n = sum(r[1] for r in rows)
    print(f"# source: {db}")
    print(f"# kernels: {len(rows)} distinct, {n} dispatches, total GPU kernel time {tot / 1e6:.3f} ms" +
          (f" = {tot / 1e6 / steps:.3f} ms per step over {steps} traced steps" if steps else ""))
    print(f"{'%time':>7} {'calls':>7} {'total_us':>12} {'avg_us':>10} {'min_us':>9} {'max_us':>10} {'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'lds':>6}  kernel")
    for r in rows:
        name = r[0].replace(".kd", "")
        print(f"{r[2] / tot * 100:7.2f} {r[1]:7d} {r[2] / 1e3:12.1f} {r[3] / 1e3:10.2f} {r[4] / 1e3:9.2f} {r[5] / 1e3:10.2f} "
              f"{r[6] or 0:5d} {r[7] or 0:5d} {r[8] or 0:5d} {r[9] or 0:6d}  {name[:160]}")


if __name__ == "__main__":
    main()
