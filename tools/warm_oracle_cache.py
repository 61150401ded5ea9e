#!/usr/bin/env python
"""Runs the slow whole-network ORACLE legs of tests/test_gpu_baseline.py on this machine's CPU and leaves their results in
tests/.oracle_cache/ (git-ignored; travels to the GPU box with the gpurun snapshot), so a GPU-box test run spends its
minutes on the device side.  The cache is keyed on the oracle sources and the inputs (tests/conftest.py: oracle_cached)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import test_gpu_baseline as tb  # noqa: E402

if __name__ == "__main__":
    fps = np.load(os.path.join(ROOT, "tests", "golden", "scan_000123_fps18000.npy"))
    t0 = time.time()
    tb.t50_oracle()
    print(f"t50 oracle trajectory: {time.time() - t0:.0f} s", flush=True)
    t0 = time.time()
    tb.c1_oracle(fps)
    print(f"C1 oracle step: {time.time() - t0:.0f} s", flush=True)
