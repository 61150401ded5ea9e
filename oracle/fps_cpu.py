"""Farthest-point sampling oracle: open3d ``PointCloud.farthest_point_down_sample`` as
DiffCompletion.preprocess_scan calls it (/root/reference/lidiff/tools/diff_completion_pipeline.py:97-99).

TEST INFRASTRUCTURE ONLY (see oracle/me_cpu.py).  PARITY UNPINNED: open3d==0.17.0 (requirements.txt:8) is not
installable here; this restates its published algorithm (geometry/PointCloud.cpp FarthestPointDownSample): start
from point 0; keep, for every point, the smallest squared float64 distance to the selected set; next = the FIRST
point of maximal distance; return the selected indices in selection order.
"""
from __future__ import annotations

import numpy as np


def farthest_point_sample(points: np.ndarray, n_samples: int) -> np.ndarray:
    pts = np.asarray(points, dtype=np.float64)
    n = pts.shape[0]
    if n_samples >= n:
        return np.arange(n, dtype=np.int64)
    sel = np.empty(n_samples, np.int64)
    dist = np.full(n, np.inf)
    far = 0
    for i in range(n_samples):
        sel[i] = far
        d = pts - pts[far]
        dist = np.minimum(dist, d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2])
        far = int(np.argmax(dist))                  # first maximum
    return sel


def range_filter(points: np.ndarray, max_range: float = 50.0, min_range: float = 3.5) -> np.ndarray:
    """pipeline:93-94: keep 3.5 m < |p| < max_range (distance over ALL columns handed in, App. D.8)."""
    pts = np.asarray(points)
    dist = np.sqrt(np.sum(pts ** 2, -1))
    return pts[(dist < max_range) & (dist > min_range)][:, :3]
