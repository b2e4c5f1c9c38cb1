"""Timeline of ONE replayed training step from a rocprofv3 (rocpd SQLite) kernel trace: which kernels ran on which
queue/stream, in start order, with the idle gap before each one on the main stream, plus per-stream busy time.

    python tools/rocpd_timeline.py /tmp/prof/vtn_results.db [step_index_from_end=1] [marker=adam_update] > gpurun_out/timeline.txt

`marker` is a substring of the kernel that ends a step (adam_update for the training steps, decode_advance for one position
of the autoregressive decode loop).
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    c = sqlite3.connect(db)
    t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [x for x in t if "kernel_dispatch" in x][0]
    ks = [x for x in t if "kernel_symbol" in x][0]
    cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
    print("# dispatch columns:", cols)
    sid = "stream_id" if "stream_id" in cols else "queue_id"
    rows = list(c.execute(f"select d.start, d.end, d.{sid}, d.queue_id, s.kernel_name from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
    marker = sys.argv[3] if len(sys.argv) > 3 else "adam_update"
    marks = [i for i, r in enumerate(rows) if marker in r[4]]
    if len(marks) < back + 1:
        print("not enough steps")
        return
    lo, hi = marks[-back - 1] + 1, marks[-back] + 1
    step = rows[lo:hi]
    t0 = step[0][0]
    print(f"# step: {len(step)} dispatches, wall {(step[-1][1] - t0) / 1e3:.1f} us, kernel-time sum {sum(r[1] - r[0] for r in step) / 1e3:.1f} us")
    streams = {}
    for r in step:
        streams.setdefault(r[2], []).append(r)
    main_s = max(streams, key=lambda k: len(streams[k]))
    for k, v in sorted(streams.items(), key=lambda kv: -len(kv[1])):
        print(f"# stream {k}: {len(v)} kernels, busy {sum(r[1] - r[0] for r in v) / 1e3:.1f} us")
    last_end = {}
    any_end = t0
    for r in step:
        gap = (r[0] - last_end.get(r[2], r[0])) / 1e3
        idle = max(0.0, (r[0] - any_end) / 1e3)        # whole-GPU idle before this kernel
        tag = "M" if r[2] == main_s else "s"
        name = r[4].replace(".kd", "")
        name = name.replace("_ZN12_GLOBAL__N_1", "")
        print(f"{(r[0] - t0) / 1e3:9.1f} {tag}{r[2]:<3d} dur {(r[1] - r[0]) / 1e3:7.1f} gap {gap:6.1f} idle {idle:5.1f}  {name[:90]}")
        last_end[r[2]] = r[1]
        any_end = max(any_end, r[1])


main()
