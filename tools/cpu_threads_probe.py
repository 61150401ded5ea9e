#!/usr/bin/env python
"""bench.py's cpu_baseline leg (one C1 step on the C++ / OpenMP restatement of ME's CPU path) at several thread counts:
    python tools/cpu_threads_probe.py 32 64
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

if __name__ == "__main__":
    scan = np.load(os.path.join(ROOT, "tests", "golden", "scan_000123_fps18000.npy")).astype(np.float32)
    for n in [int(a) for a in sys.argv[1:]] or [0]:
        r = bench.cpu_baseline(scan, threads=n)
        print(f"threads {r['cores']}: {1.0 / r['value']:.1f} s per step = {r['value']:.4f} steps/s", flush=True)
