#!/usr/bin/env python
"""Host wall time of the phases of the chain behind x_t's points (voxelise -> strided maps -> kernel maps -> tail maps -> up
orders -> part->full matches), each followed by a device synchronise, on an otherwise idle GPU: what the step waits for once the
condition encoders are through.    python tools/debug/chain_probe.py [--sigma 1.0]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import lidiff_amd.MinkowskiEngine as ME  # noqa: E402
from lidiff_amd import ops  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--sigma", type=float, default=1.0)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    scan = np.load(os.path.join(ROOT, "tests", "golden", "scan_000123_fps18000.npy")).astype(np.float32)
    rng = np.random.default_rng(0)
    base = torch.from_numpy(np.tile(scan, (10, 1))).to(dev)
    cond = ME.TensorField(features=base, coordinates=torch.cat([torch.zeros(180000, 1, device=dev), torch.round(base / 0.05)], 1), device=dev)
    cond.sparse()
    cm = cond.coordinate_manager
    cm.prebuild_strides()
    part_c = cm.maps[16].coords
    tot = {}
    for it in range(6):
        pts = base + a.sigma * torch.from_numpy(rng.standard_normal((180000, 3)).astype(np.float32)).to(dev)
        coord = torch.cat([torch.zeros(180000, 1, device=dev), torch.round(pts / 0.05)], 1)
        torch.cuda.synchronize()
        t = [time.perf_counter()]
        mark = lambda: (torch.cuda.synchronize(), t.append(time.perf_counter()))
        f = ME.TensorField(features=pts, coordinates=coord, device=dev)
        f.sparse(); mark()
        mgr = f.coordinate_manager
        mgr.prebuild_strides(); mark()
        ts = 1
        while True:
            mgr.kernel_map(ts, ts, 3)
            if ts == 16:
                break
            mgr.kernel_map(ts, 2 * ts, 2); mgr.kernel_map(2 * ts, ts, 2, True)
            ts *= 2
        mark()
        for ts in list(mgr.maps):
            if mgr.maps[ts].coords.shape[0] >= 1024 and mgr.is_sparse_map(ts, ts, 3):
                mgr.tail_map(ts)
        mark()
        for ts in (1, 2, 4, 8):
            mgr.up_order(2 * ts, ts)
        mark()
        for ts in (1, 2, 4, 8, 16):
            ops.nn_match(mgr.maps[ts].coords, part_c)
        mark()
        if it >= 1:
            for k, (x, y) in zip(("voxelise+mean", "4 strided maps", "13 kernel maps", "tail maps", "up orders", "5 matches"), zip(t, t[1:])):
                tot[k] = tot.get(k, 0.0) + (y - x) * 1e3 / 5
    print(f"sigma {a.sigma}: " + "  ".join(f"{k} {v:.2f} ms" for k, v in tot.items()) + f"  | total {sum(tot.values()):.2f} ms")
