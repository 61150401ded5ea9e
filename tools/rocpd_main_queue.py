#!/usr/bin/env python
"""Busy / idle accounting of the MAIN queue (the one the dominant conv kernel runs on) from a rocprofv3 --kernel-trace rocpd
database over bench.py: span of the last `--last-ms`, sum of its kernels' durations, the gaps between consecutive kernels binned
by length, and the kernels by total time.    python tools/rocpd_main_queue.py x_results.db [--last-ms 380]"""
import collections
import re
import sqlite3
import sys


def main():
    path = sys.argv[1]
    arg = lambda k, d: type(d)(sys.argv[sys.argv.index(k) + 1]) if k in sys.argv else d
    last_ms = arg("--last-ms", 380.0)
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    qcol = "queue_id" if "queue_id" in cols else "stream_id"
    rows = db.execute(f"select start, end, {name_col}, {qcol} from kernels order by start").fetchall()
    t_end = max(r[1] for r in rows)
    rows = [r for r in rows if r[0] >= t_end - last_ms * 1e6]
    per_q = collections.Counter()
    for s, e, n, q in rows:
        if "spconv_fwd" in n:
            per_q[q] += e - s
    mainq = per_q.most_common(1)[0][0]
    mq = [r for r in rows if r[3] == mainq]
    span = mq[-1][1] - mq[0][0]
    busy = sum(e - s for s, e, _, _ in mq)
    bins = collections.OrderedDict((k, [0, 0.0]) for k in ("<2us", "2-5us", "5-10us", "10-30us", "30-100us", ">100us"))
    for (s0, e0, _, _), (s1, e1, _, _) in zip(mq, mq[1:]):
        g = (s1 - e0) / 1e3
        if g <= 0:
            continue
        k = "<2us" if g < 2 else "2-5us" if g < 5 else "5-10us" if g < 10 else "10-30us" if g < 30 else "30-100us" if g < 100 else ">100us"
        bins[k][0] += 1
        bins[k][1] += g
    print(f"main queue q{mainq}: {len(mq)} kernels over {span / 1e6:.2f} ms: busy {busy / 1e6:.2f} ms ({100 * busy / span:.1f} %), idle {(span - busy) / 1e6:.2f} ms")
    for k, (n, us) in bins.items():
        print(f"  gaps {k:>8}: {n:6d} = {us / 1e3:7.3f} ms")
    short = lambda n: re.sub(r"\(.*$", "", n).replace("void ", "").replace("lidiff::", "")[:60]
    agg = collections.defaultdict(lambda: [0, 0])
    for s, e, n, q in mq:
        a = agg[short(n)]
        a[0] += e - s
        a[1] += 1
    print("main-queue kernels by time:")
    for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:22]:
        print(f"  {t / 1e6:8.3f} ms {c:6d} x {t / c / 1e3:8.1f} us  {n}")
    other = collections.defaultdict(float)
    for s, e, n, q in rows:
        if q != mainq:
            other[q] += e - s
    print("other queues busy (ms):", {f"q{q}": round(v / 1e6, 2) for q, v in other.items()})
    # the long gaps of the main queue: between which kernels, and what the other queues ran meanwhile
    gaps = sorted(((s1 - e0, e0, s1, n0, n1) for (s0, e0, n0, _), (s1, e1, n1, _) in zip(mq, mq[1:]) if s1 - e0 > 100e3), reverse=True)
    pair = collections.defaultdict(lambda: [0, 0.0])
    for g, e0, s1, n0, n1 in gaps:
        a = pair[(short(n0)[:40], short(n1)[:40])]
        a[0] += 1
        a[1] += g
    print("main-queue gaps > 100 us by (kernel before -> kernel after):")
    for (n0, n1), (c, t) in sorted(pair.items(), key=lambda kv: -kv[1][1])[:8]:
        print(f"  {t / 1e6:8.3f} ms in {c:4d} gaps | {n0}  ->  {n1}")
    if gaps:
        g, e0, s1, n0, n1 = gaps[len(gaps) // 2]
        print(f"inside a median long gap ({g / 1e3:.0f} us, after {short(n0)[:40]}): kernels of the other queues")
        for s, e, n, q in rows:
            if q != mainq and e > e0 and s < s1:
                print(f"    {(s - e0) / 1e3:8.1f} us  +{(e - s) / 1e3:7.1f}  q{q}  {short(n)[:70]}")


if __name__ == "__main__":
    main()
