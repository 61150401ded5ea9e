#!/usr/bin/env python
"""Idle gaps of the GPU from a rocprofv3 --kernel-trace rocpd database: merges the busy intervals of ALL queues, lists the
largest gaps with the kernels before / after, and sums idle time per step-sized window.
    python tools/rocpd_gaps.py gpurun_out/prof/x_results.db [--min-us 20] [--top 25] [--last-ms 0]"""
import re
import sqlite3
import sys


def main():
    path = sys.argv[1]
    arg = lambda k, d: float(sys.argv[sys.argv.index(k) + 1]) if k in sys.argv else d
    min_us, top = arg("--min-us", 20.0), int(arg("--top", 25))
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = db.execute(f"select start, end, {name_col} from kernels order by start").fetchall()
    last_ms = arg("--last-ms", 0.0)              # only the final stretch of the trace (the timed steps of bench.py)
    if last_ms > 0:
        t_end = max(r[1] for r in rows)
        rows = [r for r in rows if r[0] >= t_end - last_ms * 1e6]
    short = lambda n: re.sub(r"\(.*$", "", n).replace("void ", "")[:70]
    t0 = rows[0][0]
    busy_end, last_name = rows[0][1], rows[0][2]
    gaps = []
    total_busy = 0
    cur_start = rows[0][0]
    for s, e, n in rows[1:]:
        if s > busy_end:
            total_busy += busy_end - cur_start
            gaps.append(((s - busy_end) / 1e3, (busy_end - t0) / 1e6, short(last_name), short(n)))
            cur_start = s
        if e > busy_end:
            busy_end, last_name = e, n
    total_busy += busy_end - cur_start
    span = busy_end - t0
    print(f"kernels {len(rows)}, span {span / 1e6:.2f} ms, busy (any queue) {total_busy / 1e6:.2f} ms, idle {(span - total_busy) / 1e6:.2f} ms "
          f"({100.0 * (span - total_busy) / span:.1f} %), gaps >= {min_us} us: {sum(1 for g in gaps if g[0] >= min_us)} "
          f"= {sum(g[0] for g in gaps if g[0] >= min_us) / 1e3:.2f} ms; gaps < {min_us} us: {sum(g[0] for g in gaps if g[0] < min_us) / 1e3:.2f} ms")
    print("largest gaps (us | at ms | after kernel -> before kernel):")
    for g in sorted(gaps, reverse=True)[:top]:
        print(f"  {g[0]:9.1f} | {g[1]:9.2f} | {g[2]}  ->  {g[3]}")
    # idle by kernel pair
    agg = {}
    for g in gaps:
        k = (g[2], g[3])
        a = agg.setdefault(k, [0.0, 0])
        a[0] += g[0]
        a[1] += 1
    print("idle by (after -> before) pair:")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
        print(f"  {a[0] / 1e3:8.3f} ms in {a[1]:5d} gaps | {k[0]}  ->  {k[1]}")


if __name__ == "__main__":
    main()
