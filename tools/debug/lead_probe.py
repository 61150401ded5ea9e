"""Is the host ahead of the device?  At fixed points of every step the host asks whether the MAIN stream has drained
(stream.query(): True = the device is waiting for the host) and notes its own clock."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from lidiff_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
pipe = bench.build_pipeline(dev)
steps = 12
x_init, xs, tvals = bench.make_inputs(pipe, bench.load_scan(), steps, seed=1000, device=dev)
main = torch.cuda.current_stream(dev)
log = []
orig_get = ops.SizeFeed.get


def get(self, seq=None, timeout_s=30.0):
    t0 = time.perf_counter()
    idle0 = main.query()
    out = orig_get(self, seq, timeout_s)
    log.append(("feed.get", 1e3 * (time.perf_counter() - t0), idle0, main.query()))
    return out


ops.SizeFeed.get = get
orig_cp = pipe.classfree_pair


def cp(*a, **k):
    log.append(("step enter", 1e3 * time.perf_counter(), main.query(), None))
    out = orig_cp(*a, **k)
    log.append(("network queued", 1e3 * time.perf_counter(), main.query(), None))
    return out


pipe.classfree_pair = cp
with torch.no_grad():
    bench.run_steps(pipe, x_init, xs[:3], tvals[:3], 0, 3)
    torch.cuda.synchronize()
    log.clear()
    bench.run_steps(pipe, x_init, xs, tvals, 0, steps)
    torch.cuda.synchronize()
t0 = None
for name, t, a, b in log:
    if name == "feed.get":
        print(f"    feed.get waited {t:7.3f} ms   main idle before: {a}  after: {b}")
    else:
        t0 = t0 or t
        print(f"{name:<16} at {t - t0:9.3f} ms   main stream idle: {a}")
