"""Where the HOST spends a denoising step (python tools/host_profile.py [--steps 10] [--sort tottime]): cProfile over bench.py's
run_steps after a warm-up, plus the time the host spends WAITING for the device (SizeFeed.get polls, the final synchronise) -- a
step whose host time equals its device time is launch-bound, whatever the kernels do."""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from lidiff_amd import ops  # noqa: E402

steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 10
sort = sys.argv[sys.argv.index("--sort") + 1] if "--sort" in sys.argv else "tottime"
dev = torch.device("cuda:0")
pipe = bench.build_pipeline(dev)
x_init, xs, tvals = bench.make_inputs(pipe, bench.load_scan(), steps, seed=1000, device=dev)
waits = [0.0]
orig_get = ops.SizeFeed.get


def timed_get(self, seq=None, timeout_s=30.0):
    t0 = time.perf_counter()
    try:
        return orig_get(self, seq, timeout_s)
    finally:
        waits[0] += time.perf_counter() - t0


ops.SizeFeed.get = timed_get
with torch.no_grad():
    for _ in range(2):
        bench.run_steps(pipe, x_init, xs, tvals, 0, min(3, steps))
    torch.cuda.synchronize()
    waits[0] = 0.0
    t0 = time.perf_counter()
    bench.run_steps(pipe, x_init, xs, tvals, 0, steps)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"plain: host returns after {1e3 * t_host / steps:.2f} ms/step, device done after {1e3 * t_all / steps:.2f} ms/step; "
          f"of the host time {1e3 * waits[0] / steps:.2f} ms/step waiting in SizeFeed.get")
    waits[0] = 0.0
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    bench.run_steps(pipe, x_init, xs, tvals, 0, steps)
    pr.disable()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"under cProfile: host {1e3 * t_host / steps:.2f} ms/step, waiting {1e3 * waits[0] / steps:.2f} ms/step")
s = io.StringIO()
pstats.Stats(pr, stream=s).strip_dirs().sort_stats(sort).print_stats(45)
print(s.getvalue()[:9000])
