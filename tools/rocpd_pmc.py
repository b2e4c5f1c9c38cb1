"""Average per-dispatch PMC counter values per kernel from a rocprofv3 (rocpd SQLite) counter-collection run.

    python tools/rocpd_pmc.py <results.db> [kernel-substring]

Run counters in their own passes (rocprofv3 --pmc FETCH_SIZE --kernel-trace ... ; --pmc WRITE_SIZE --kernel-trace ...),
never together with the trace domains gpurun refuses.  HBM traffic per launch on gfx950 (MI355X_MICROARCH.md, HBM
section): bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- FETCH_SIZE (KB) under-reports wide streaming reads by
exactly 2x on this rocprofv3.
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    c = sqlite3.connect(db)
    q = ("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
         "group by kernel_name, counter_name order by sum(duration) desc")
    print(f"# source: {db}")
    print(f"{'counter':16s} {'dispatches':>10s} {'avg value/dispatch':>20s} {'avg us':>10s}  kernel")
    for name, ctr, n, avg, dur in c.execute(q):
        if sub in name:
            print(f"{ctr:16s} {n:10d} {avg:20.1f} {dur / 1e3:10.2f}  {name[:110]}")


if __name__ == "__main__":
    main()
