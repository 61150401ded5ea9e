#!/usr/bin/env python
"""Status of the heavy oracle fixtures (tests/golden/c1_t*.npz, t50_small.npz; tests/heavy_oracle.py) and, for any that is
absent or stale (other inputs / edited oracle sources), a run of the oracle on this machine's CPU into tests/.oracle_cache/
(git-ignored; travels to the GPU box with the gpurun snapshot) so that a GPU-box test run spends its minutes on the device
side.  Regenerate the tracked fixtures themselves with `python tests/golden/make_golden.py --heavy`."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import heavy_oracle as heavy  # noqa: E402

if __name__ == "__main__":
    fps = np.load(os.path.join(ROOT, "tests", "golden", "scan_000123_fps18000.npy"))
    t0 = time.time()
    print("t50_small:", heavy.t50_oracle()[-1], f"{time.time() - t0:.0f} s", flush=True)
    for t in heavy.C1_TIMESTEPS:
        t0 = time.time()
        print(f"c1_t{t}:", heavy.c1_oracle(fps, t)[1], f"{time.time() - t0:.0f} s", flush=True)
