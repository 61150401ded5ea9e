"""Does it matter that the network runs on the DEFAULT (null) stream?  20 steps of bench.run_steps on the default stream and
inside an explicit torch.cuda.Stream."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda:0")
pipe = bench.build_pipeline(dev)
steps = 20
x_init, xs, tvals = bench.make_inputs(pipe, bench.load_scan(), steps, seed=1000, device=dev)


def run():
    with torch.no_grad():
        for w in range(3):
            bench.run_steps(pipe, x_init, xs[:5], tvals[:5], w, 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bench.run_steps(pipe, x_init, xs, tvals, 0, steps)
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / steps


print("default stream   : %.2f ms/step" % run())
print("default stream   : %.2f ms/step" % run())
s = torch.cuda.Stream(dev)
pipe._side = None
with torch.cuda.stream(s):
    print("explicit stream  : %.2f ms/step" % run())
    print("explicit stream  : %.2f ms/step" % run())
