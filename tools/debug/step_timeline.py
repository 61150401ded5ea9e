#!/usr/bin/env python
"""Where a denoising step's wall time goes on the MAIN stream (HIP events inside DiffCompletion.classfree_forward):
condition encoders | wait for the side stream (x_t's maps and matches) | MinkUNetDiff | rest of the step.
    python tools/debug/step_timeline.py [--steps 10]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    pipe = bench.build_pipeline(dev)
    x_init, xs, tvals = bench.make_inputs(pipe, bench.load_scan(), a.steps, seed=1000, device=dev)
    with torch.no_grad():
        for w in range(2):
            bench.run_steps(pipe, x_init, xs[:2], tvals[:2], w, 1)
        torch.cuda.synchronize()
        pipe.timeline = []
        pipe.host_stamps = []
        t0 = time.perf_counter()
        bench.run_steps(pipe, x_init, xs, tvals, 0, a.steps)
        torch.cuda.synchronize()
        wall = 1e3 * (time.perf_counter() - t0) / a.steps
    tl = pipe.timeline
    seg = np.array([[m[i].elapsed_time(m[i + 1]) for i in range(3)] for m in tl])
    between = np.array([tl[i][3].elapsed_time(tl[i + 1][0]) for i in range(len(tl) - 1)])
    print(f"wall {wall:.2f} ms per step; main stream: encoders {seg[:, 0].mean():.2f}  wait for x_t maps {seg[:, 1].mean():.2f}  "
          f"UNet {seg[:, 2].mean():.2f}  between steps (scheduler, next field) {between.mean():.2f} ms")
    for j, r in enumerate(seg):
        print(f"  step {j} (t={tvals[j]}): encoders {r[0]:.2f}  wait {r[1]:.2f}  UNet {r[2]:.2f}")
    # host thread: where its time goes between the stamps (ms, averaged over the steps after the first two)
    st = pipe.host_stamps
    seg = {}
    order = []
    for (la, ta), (lb, tb) in zip(st, st[1:]):
        k = f"{la} -> {lb}"
        if k not in seg:
            order.append(k)
        seg.setdefault(k, []).append((tb - ta) * 1e3)
    print("host thread (ms per occurrence, mean over all occurrences):")
    for k in order:
        v = seg[k]
        print(f"  {k:<70} {np.mean(v[2:] if len(v) > 4 else v):7.3f}  x{len(v)}")
