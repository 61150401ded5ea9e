"""Timing of the evaluation kernels at the bench size (180k x 180k nearest-neighbour search, IoU counts)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from lidiff_amd import evaluation as ev, ops  # noqa: E402

dev = "cuda"
rng = np.random.default_rng(0)
for dt in (torch.float32, torch.float64):
    for n, m in ((180000, 180000), (18000, 180000), (180000, 18000)):
        a = torch.from_numpy(rng.uniform(-50, 50, (n, 3))).to(dev, dt)
        b = torch.from_numpy(rng.uniform(-50, 50, (m, 3))).to(dev, dt)
        ops.nn_dist(a, b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            ops.nn_dist(a, b)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        print(f"nn_dist {str(dt):14s} {n:7d} x {m:7d}: {ms:8.3f} ms  {n * m / ms / 1e6:8.1f} G pair/s", flush=True)
a = rng.uniform(-50, 50, (180000, 3))
b = a + rng.normal(0, 0.1, a.shape)
for cls in (ev.ChamferDistance, ev.CompletionIoU):
    mtr = cls()
    mtr.update(a, b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mtr.update(a, b)
    torch.cuda.synchronize()
    print(f"{cls.__name__}.update 180k vs 180k: {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
