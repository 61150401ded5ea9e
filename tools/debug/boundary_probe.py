#!/usr/bin/env python
"""Main-stream events around one step boundary of bench.run_steps (GPU clock, relative to the end of the previous UNet) next to
the host times at which they were queued: where the main stream waits between two steps.    python tools/debug/boundary_probe.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    pipe = bench.build_pipeline(dev)
    steps = 8
    x_init, xs, tvals = bench.make_inputs(pipe, bench.load_scan(), steps, seed=1000, device=dev)
    rows = []

    def ev(label):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        cur.append((label, time.perf_counter(), e))

    user_stream = torch.cuda.Stream() if os.environ.get("PROBE_USER_STREAM") else torch.cuda.current_stream()
    with torch.no_grad(), torch.cuda.stream(user_stream):
        bench.run_steps(pipe, x_init, xs[:2], tvals[:2], 0, 2)
        torch.cuda.synchronize()
        pipe.timeline = []
        x_t = pipe.points_to_tensor(xs[0])
        x_cond = pipe.points_to_tensor(x_init)
        x_uncond = pipe.points_to_tensor(torch.zeros_like(x_init))
        for j in range(steps):
            cur = []
            ev("loop top")
            t = torch.full((1,), tvals[j], dtype=torch.int64, device=dev)
            ev("t made")
            noise_t = pipe.classfree_forward(x_t, x_cond, x_uncond, t, None, t_host=tvals[j])
            ev("classfree returned")
            input_noise = x_t.F.reshape(1, -1, 3) - x_init
            if j == 0:
                pipe.new_scheduler()
            _ = x_init + pipe.dpm_scheduler.step(noise_t, tvals[j], input_noise)["prev_sample"]
            ev("scheduler queued")
            import lidiff_amd.MinkowskiEngine as ME
            pts = xs[min(j + 1, len(xs) - 1)]
            x_feats = ME.utils.batched_coordinates(list(pts[:]), dtype=torch.float32, device=dev)
            ev("  batched_coordinates")
            x_coord = torch.round(x_feats / pipe.hparams["data"]["resolution"])
            ev("  round")
            x_t = ME.TensorField(features=x_feats[:, 1:], coordinates=x_coord,
                                 quantization_mode=ME.SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE,
                                 minkowski_algorithm=ME.MinkowskiAlgorithm.SPEED_OPTIMIZED, device=dev)
            ev("  TensorField")
            x_t.coordinate_manager.pyramid = pipe.single_read
            x_t.ready = torch.cuda.Event()
            x_t.ready.record(torch.cuda.current_stream(dev))
            ev("next points queued")
            x_cond, x_uncond = pipe.reset_partial_pcd(x_cond, x_uncond, next_t=tvals[j + 1] if j + 1 < steps else None)
            ev("conditions reset")
            rows.append(cur)
        torch.cuda.synchronize()
    tl = pipe.timeline
    for j in range(2, steps - 1):
        unet_end = tl[j][3]
        t_host0 = rows[j][2][1]
        print(f"step {j}: UNet end = 0")
        for label, th, e in rows[j][2:] + rows[j + 1][:2]:
            print(f"   {label:<22} GPU {unet_end.elapsed_time(e) * 1e3:9.1f} us   host queued at {1e3 * (th - t_host0):8.3f} ms after 'classfree returned'")
        m = tl[j + 1]
        print(f"   marks of step {j + 1}: enter {unet_end.elapsed_time(m[0]) * 1e3:.1f}  conditions {unet_end.elapsed_time(m[1]) * 1e3:.1f}  "
              f"x_t voxelised {unet_end.elapsed_time(m[2]) * 1e3:.1f} us")


if __name__ == "__main__":
    main()
