#!/usr/bin/env python
"""Where a denoising step's wall time goes on the MAIN stream (HIP events inside DiffCompletion.classfree_forward):
condition encoders | wait for the side stream (x_t's maps and matches) | MinkUNetDiff | rest of the step.
    python tools/step_timeline.py [--steps 10]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
        from lidiff_amd import ops
        ops.PYRAMID_TRACE = []
        ms0 = torch.cuda.memory_stats()
        t0 = time.perf_counter()
        bench.run_steps(pipe, x_init, xs, tvals, 0, a.steps)
        torch.cuda.synchronize()
        wall = 1e3 * (time.perf_counter() - t0) / a.steps
    ms1 = torch.cuda.memory_stats()
    print("allocator over the timed steps: hipMalloc segments +%d, retries %d, reserved %.2f GB, allocated peak %.2f GB" % (
        ms1["segment.all.allocated"] - ms0["segment.all.allocated"], ms1["num_alloc_retries"] - ms0["num_alloc_retries"],
        ms1["reserved_bytes.all.current"] / 1e9, ms1["allocated_bytes.all.peak"] / 1e9))
    tr = ops.PYRAMID_TRACE
    ops.PYRAMID_TRACE = None
    segs, order = {}, []
    tr_ev = tr
    tr = [(a, b) for a, b, _ in tr]
    for (la, ta), (lb, tb) in zip(tr, tr[1:]):
        if lb == "enter":
            continue
        k = f"{la} -> {lb}"
        if k not in segs:
            order.append(k)
        segs.setdefault(k, []).append((tb - ta) * 1e6)
    print("build_pyramid host time (us: mean / max over all calls):")
    for k in order:
        print(f"  {k:<44} {np.mean(segs[k]):8.1f} {np.max(segs[k]):8.1f}  x{len(segs[k])}")
    tl = pipe.timeline
    # GPU-time view of every pyramid of 180k points (x_t's) against the main-stream marks: offsets from the end of the previous UNet
    big = [i for i, (l, _, e) in enumerate(tr_ev) if l == "enter"]
    rows = []
    for i in big:
        evs = {l: e for l, _, e in tr_ev[i:i + 10] if e is not None}
        hosts = {l: t for l, t, _ in tr_ev[i:i + 10]}
        if len(evs) < 3:
            continue
        chain = evs["enter"].elapsed_time(evs["tails queued"])
        if chain < 0.25:                 # the small condition fields
            continue
        for j in range(1, len(tl)):
            m = tl[j]
            if tl[j - 1][3].elapsed_time(evs["enter"]) >= 0 and evs["enter"].elapsed_time(m[2]) >= 0:
                rows.append((tl[j - 1][3].elapsed_time(m[0]), tl[j - 1][3].elapsed_time(evs["enter"]), chain,
                             evs["tails queued"].elapsed_time(evs["done"]), evs["done"].elapsed_time(m[2]),
                             1e3 * (hosts["sizes read"] - hosts["tails queued"]), m[2].elapsed_time(m[3])))
    if rows:
        r = np.array(rows)
        for q in r:
            print("   ", " ".join(f"{v:8.3f}" for v in q))
        print("x_t pyramid on the GPU clock (ms, mean over %d steps): UNet end -> step mark %.3f, -> chain start %.3f | chain (device "
              "counts) %.3f | end of chain -> read done + narrow %.3f | -> mark 'x_t voxelised' on main %.3f  (host blocked in the "
              "read %.3f) | UNet %.2f" % ((len(r),) + tuple(r.mean(0))))
    seg = np.array([[m[i].elapsed_time(m[i + 1]) for i in range(3)] for m in tl])
    between = np.array([tl[i][3].elapsed_time(tl[i + 1][0]) for i in range(len(tl) - 1)])
    print(f"wall {wall:.2f} ms per step; main stream: encoders {seg[:, 0].mean():.2f}  wait for x_t maps {seg[:, 1].mean():.2f}  "
          f"UNet {seg[:, 2].mean():.2f}  between steps (scheduler, next field) {between.mean():.2f} ms")
    if len(seg) > 4:        # steady state: without the first three steps (cold encoders, first lazy pyramids)
        s3, b3 = seg[3:], between[2:]
        print(f"steady state (steps 3..): encoders {s3[:, 0].mean():.2f}  wait for x_t maps {s3[:, 1].mean():.2f}  UNet {s3[:, 2].mean():.2f}  "
              f"between steps {b3.mean():.2f} ms  =>  UNet end -> next UNet start (mark to mark) "
              f"{(b3.mean() + s3[:, 0].mean() + s3[:, 1].mean()):.2f} ms")
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
