#!/usr/bin/env python
"""What a row count BOUND costs a tile-kernel launch: the same convolution with its map handed over at the exact size and at
the point-count bound (tiles behind the device-side count leave at once).   python tools/debug/bound_cost.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lidiff_amd import ops  # noqa: E402
import lidiff_amd.MinkowskiEngine as ME  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    scan = np.load(os.path.join(ROOT, "tests", "golden", "scan_000123_fps18000.npy"))
    pts = np.tile(scan, (10, 1)) + np.random.default_rng(0).standard_normal((180000, 3)).astype(np.float32)
    feats = torch.from_numpy(pts.astype(np.float32)).to(dev)
    coord = torch.cat([torch.zeros(180000, 1, device=dev), torch.round(feats / 0.05)], 1)
    field = ME.TensorField(features=feats, coordinates=coord, device=dev)
    field.sparse()
    mgr = field.coordinate_manager
    ts = 1
    for _ in range(4):
        ts = mgr.stride(ts, 2)
    n = 180000
    for level, cin, cout in ((4, 256, 256), (3, 256, 256), (3, 128, 128), (2, 128, 128), (2, 64, 64)):
        ts = 1 << level
        nbr = mgr.kernel_map(ts, ts, 3)
        m = nbr.shape[1]
        x = torch.randn(2 * m, cin, device=dev)
        w = torch.randn(27, cin, cout, device=dev) * 0.05
        nbr_b = torch.full((27, n), -1, dtype=torch.int32, device=dev)
        nbr_b[:, :m] = nbr
        xb = torch.zeros(2 * n, cin, device=dev)
        xb[:m], xb[n:n + m] = x[:m], x[m:]
        rows = torch.tensor([m], dtype=torch.int32, device=dev)
        hint = mgr.is_sparse_map(ts, ts, 3, c_out=cout)
        forms = {"exact": lambda: ops.spconv_fwd(x, w, nbr, m, replicas=2, sparse_map=hint),
                 "bound": lambda: ops.spconv_fwd(xb, w, nbr_b, n, replicas=2, sparse_map=hint, d_rows=rows, rows_hint=m, in_rows_hint=m)}
        res = {}
        for name, f in forms.items():
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                f()
            e.record()
            torch.cuda.synchronize()
            res[name] = 1e3 * s.elapsed_time(e) / 20
        print(f"level {level} {cin}->{cout} rows {m} of bound {n}: exact {res['exact']:.1f} us, bound {res['bound']:.1f} us "
              f"(+{res['bound'] - res['exact']:.1f} us)", flush=True)


if __name__ == "__main__":
    main()
