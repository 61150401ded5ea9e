#!/usr/bin/env python
"""Times the coordinate-side kernels (voxel hash, strided maps, kernel maps, voxel mean, slice, match) on the
bench workload (180k-point scan, sigma given) with HIP events and prints achieved GB/s against the
ALGORITHMIC bytes of SURVEY.md 8(d) (HBM roofline: 8 TB/s spec, ~6.3 TB/s achievable).

    python tools/coord_probe.py [--sigma 1.0] [--iters 20]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sigma", type=float, default=1.0)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    from lidiff_amd import ops
    dev = torch.device("cuda:0")
    scan = np.load(os.path.join(ROOT, "tests", "golden", "scan_000123_fps18000.npy"))
    rng = np.random.default_rng(0)
    pts = (np.tile(scan, (10, 1)) + a.sigma * rng.standard_normal((180000, 3))).astype(np.float32)
    feats = torch.from_numpy(pts).to(dev)
    cf = torch.cat([torch.zeros(180000, 1, device=dev), torch.round(feats / 0.05)], 1)
    st = torch.zeros(1, dtype=torch.int32, device=dev)

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(a.iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        return 1e3 * s.elapsed_time(e) / a.iters          # us

    rows = []
    N = 180000
    us = timed(lambda: ops.coords_floor(cf))
    rows.append(("coords_floor", us, 32 * N))
    ci = ops.coords_floor(cf)
    us = timed(lambda: ops.vox_unique(ci, st))
    uniq, inv, first, table = ops.vox_unique(ci, st)
    M = uniq.shape[0]
    rows.append((f"vox_unique (N={N}, M={M}; incl. table memset + 1 host sync)", us, 16 * N + 8 * N + 28 * M))
    us = timed(lambda: ops.vox_mean(feats, inv, M))
    rows.append(("vox_mean", us, 12 * N + 8 * N + 12 * M))
    cur, cur_t, ts = uniq, table, 1
    for lvl in range(1, 5):
        us = timed(lambda: ops.map_stride(cur, ts * 2, st))
        coarse, parent, ctable = ops.map_stride(cur, ts * 2, st)
        rows.append((f"map_stride level {lvl} ({cur.shape[0]} -> {coarse.shape[0]})", us, 16 * cur.shape[0] + 16 * coarse.shape[0] + 4 * cur.shape[0]))
        us = timed(lambda: ops.kernel_map(cur, cur_t, 3, ts))
        nbr = ops.kernel_map(cur, cur_t, 3, ts)
        P = int((nbr >= 0).sum())
        rows.append((f"kernel_map ks3 level {lvl - 1} (M={cur.shape[0]}, P={P})", us, 16 * 2 * cur.shape[0] + 4 * 27 * cur.shape[0]))
        us = timed(lambda: ops.kernel_map(coarse, cur_t, 2, ts))
        rows.append((f"kernel_map ks2/s2 level {lvl - 1}->{lvl}", us, 16 * (cur.shape[0] + coarse.shape[0]) + 4 * 8 * coarse.shape[0]))
        us = timed(lambda: ops.kernel_map_up(cur, parent, ts))
        rows.append((f"kernel_map_up level {lvl}->{lvl - 1}", us, 20 * cur.shape[0] + 4 * 8 * cur.shape[0]))
        cur, cur_t, ts = coarse, ctable, ts * 2
    f96 = torch.randn(M, 96, device=dev)
    us = timed(lambda: ops.gather_rows(f96, inv))
    rows.append(("gather_rows (slice) [M,96] -> [N,96]", us, 4 * (M * 96 + N * 96) + 8 * N))
    # the 'part' side as in the network: stride-16 map of the clean 18k-point scan (x_cond's encoder output)
    pc = torch.cat([torch.zeros(18000, 1, device=dev), torch.round(torch.from_numpy(scan.astype(np.float32)).to(dev) / 0.05)], 1)
    p0, _, _, _ = ops.vox_unique(ops.coords_floor(pc), st)
    pcur, ptab = p0, None
    for s_ in (2, 4, 8, 16):
        pcur, _, ptab = ops.map_stride(pcur, s_, st)
    part = pcur
    us = timed(lambda: ops.nn_match(uniq, part))
    rows.append((f"nn_match exhaustive, full {M} x part {part.shape[0]} ({8e-9 * M * part.shape[0]:.1f} GFLOP)", us, 16 * (M + part.shape[0]) + 8 * M))
    print(f"sigma={a.sigma}")
    print("kernel | us | algorithmic MB | GB/s | % of 8 TB/s")
    for name, us, b in rows:
        gbs = b / (us * 1e-6) / 1e9
        print(f"{name} | {us:.1f} | {b / 1e6:.2f} | {gbs:.0f} | {100 * gbs / 8000:.1f}")


if __name__ == "__main__":
    main()
