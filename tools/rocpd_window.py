#!/usr/bin/env python
"""Kernel sequence of a rocprofv3 --kernel-trace rocpd database around one step boundary of bench.py: everything from the N-th
last launch of `--after` (default: the head's gather_rows_kernel, the last kernel of a UNet forward) for `--ms` milliseconds, with
start offsets, durations and the queue each ran on.    python tools/rocpd_window.py x_results.db [--nth 3] [--ms 4]"""
import re
import sqlite3
import sys


def main():
    path = sys.argv[1]
    arg = lambda k, d: type(d)(sys.argv[sys.argv.index(k) + 1]) if k in sys.argv else d
    after, nth, ms = arg("--after", "gather_rows_kernel"), arg("--nth", 3), arg("--ms", 4.0)
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    sel = f"select start, end, {name_col}" + (f", {qcol}" if qcol else ", 0") + " from kernels order by start"
    rows = db.execute(sel).fetchall()
    hits = [i for i, r in enumerate(rows) if after in r[2]]
    i0 = hits[-nth]
    t0 = rows[i0][0]
    short = lambda n: re.sub(r"\(.*$", "", n).replace("void ", "").replace("at::native::", "")[:64]
    prev_end = rows[i0][0]
    for s, e, n, q in rows[i0:]:
        if (s - t0) / 1e6 > ms:
            break
        print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  q{q}  {short(n)}")


if __name__ == "__main__":
    main()
