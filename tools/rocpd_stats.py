#!/usr/bin/env python
"""Per-kernel statistics (calls, total/avg/min/max duration) from a rocprofv3 rocpd SQLite database
(`rocprofv3 --kernel-trace --stats` writes <name>_results.db on ROCm 7.2) -> markdown/CSV table.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [--csv] [--top N]
"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "")
    return name if len(name) <= 110 else name[:107] + "..."


def main():
    path = sys.argv[1]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 40
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = db.execute(f"select {name_col}, count(*), sum(end-start), min(end-start), max(end-start) "
                      f"from kernels group by {name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    sep = "," if "--csv" in sys.argv else " | "
    print(sep.join(["kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct"]))
    for n, c, t, mn, mx in rows[:top]:
        print(sep.join([short(n), str(c), f"{t / 1e6:.3f}", f"{t / c / 1e3:.1f}", f"{mn / 1e3:.1f}", f"{mx / 1e3:.1f}",
                        f"{100.0 * t / total:.1f}"]))
    print(sep.join(["TOTAL", str(sum(r[1] for r in rows)), f"{total / 1e6:.3f}", "", "", "", "100.0"]))


if __name__ == "__main__":
    main()
