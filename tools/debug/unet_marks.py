"""GPU-clock marks inside the step (no profiler): boundary -> stem -> stage1 ... on the main stream, mean over the steady steps."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from lidiff_amd import minkunet as mn  # noqa: E402

dev = torch.device("cuda:0")
pipe = bench.build_pipeline(dev)
steps = 14
x_init, xs, tvals = bench.make_inputs(pipe, bench.load_scan(), steps, seed=1000, device=dev)
marks = []
MAIN = torch.cuda.current_stream(dev)


def mark(label):
    e = torch.cuda.Event(enable_timing=True)
    if torch.cuda.current_stream(dev) != MAIN:
        return
    e.record(MAIN)
    marks.append((label, e))


orig_sb = pipe.step_boundary


def sb(*a, **k):
    mark("boundary: enter")
    out = orig_sb(*a, **k)
    mark("boundary: done")
    return out


pipe.step_boundary = sb
orig_ec = pipe.encode_conditions


def ec(*a, **k):
    mark("conditions: enter")
    out = orig_ec(*a, **k)
    mark("conditions: joined")
    return out


pipe.encode_conditions = ec
orig_rp = pipe.reset_partial_pcd


def rp(*a, **k):
    mark("reset_partial: enter (main)")
    out = orig_rp(*a, **k)
    mark("reset_partial: done (main)")
    return out


pipe.reset_partial_pcd = rp
from lidiff_amd import ops  # noqa: E402
orig_p2f, orig_mf = ops.points_to_field, pipe._make_field


def p2f(*a, **k):
    mark("p2f: enter")
    out = orig_p2f(*a, **k)
    mark("p2f: done")
    return out


def mf(*a, **k):
    mark("make_field: enter")
    out = orig_mf(*a, **k)
    mark("make_field: done")
    return out


if os.environ.get("FINE", "1") == "1":
    ops.points_to_field = p2f
    pipe._make_field = mf
with torch.no_grad():
    bench.run_steps(pipe, x_init, xs[:3], tvals[:3], 0, 3)
    torch.cuda.synchronize()
    mn.MinkUNetDiff.MARK = staticmethod(mark)
    marks.clear()
    bench.run_steps(pipe, x_init, xs, tvals, 0, steps)
    torch.cuda.synchronize()
seg = {}
order = []
for (la, ea), (lb, eb) in zip(marks, marks[1:]):
    k = f"{la} -> {lb}"
    if k not in seg:
        order.append(k)
    seg.setdefault(k, []).append(ea.elapsed_time(eb))
for k in order:
    v = seg[k][3:] or seg[k]
    print(f"{k:<60} {np.mean(v):8.3f} ms  (min {np.min(v):.3f} max {np.max(v):.3f})  x{len(v)}")
