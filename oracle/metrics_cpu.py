"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the evaluation metrics and the Chamfer loss.

Follows lidiff/utils/metrics.py (RMSE :62-78, CompletionIoU :80-121, ChamferDistance :123-141,
PrecisionRecall :143-176) and pytorch3d.loss.chamfer_distance as called at models_refine.py:72.
open3d (``PointCloud.compute_point_cloud_distance``) and pytorch3d are absent from this image: their
published behaviour -- exact nearest-neighbour Euclidean distance in float64; squared K=1 distances,
mean over points, both directions added, mean over the batch -- is restated with a scipy KD-tree.
Parity at those two boundaries is therefore unpinned; the numpy parts (histogramdd occupancy, the
threshold loops) are the reference's own arithmetic.
"""
from __future__ import annotations

import numpy as np
from scipy.spatial import cKDTree


def point_cloud_distance(src: np.ndarray, dst: np.ndarray) -> np.ndarray:
    """open3d ``src.compute_point_cloud_distance(dst)``: per src point the distance to its nearest dst point."""
    d, _ = cKDTree(np.asarray(dst, dtype=np.float64)).query(np.asarray(src, dtype=np.float64), k=1)
    return d


def rmse_update(gt: np.ndarray, pt: np.ndarray) -> float:
    """metrics.py:68-70."""
    return float(np.mean(point_cloud_distance(pt, gt)))


def chamfer_update(gt: np.ndarray, pt: np.ndarray) -> float:
    """metrics.py:128-131."""
    return float((np.mean(point_cloud_distance(gt, pt)) + np.mean(point_cloud_distance(pt, gt))) / 2)


def completion_iou_counts(gt: np.ndarray, pred: np.ndarray, voxel_sizes=(0.5, 0.2, 0.1), max_range=50.0):
    """metrics.py:86-106: (tp, fn, fp) per voxel size from boolean occupancy histograms over [-50, 50]^3.
    The dense histogram of the reference (bins^3 cells) is replaced by the set of occupied cells -- the same
    searchsorted binning np.histogramdd performs, including its closed last bin."""
    out = []
    for vsize in voxel_sizes:
        bins = int(2 * max_range / vsize)
        edges = np.linspace(-max_range, max_range, bins + 1)

        def occupied(p):
            p = np.asarray(p, dtype=np.float64)[:, :3]
            ix = np.stack([np.searchsorted(edges, p[:, d], side="right") for d in range(3)], axis=1)
            on_edge = p == edges[-1]
            ix[on_edge] -= 1
            keep = np.all((ix >= 1) & (ix <= bins), axis=1)
            ix = ix[keep] - 1
            return set(map(tuple, ix.tolist()))
        g, q = occupied(gt), occupied(pred)
        tp = len(g & q)
        out.append((tp, len(g) - tp, len(q) - tp))
    return np.asarray(out, dtype=np.uint64)


def precision_recall_update(gt: np.ndarray, pt: np.ndarray, thresholds):
    """metrics.py:150-170: percentages per threshold."""
    d_pt = point_cloud_distance(pt, gt)
    d_gt = point_cloud_distance(gt, pt)
    res = []
    for t in thresholds:
        p = 100 / len(d_pt) * len(np.where(d_pt < t)[0])
        r = 100 / len(d_gt) * len(np.where(d_gt < t)[0])
        f = 0 if p == 0 or r == 0 else 2 * p * r / (p + r)
        res.append((p, r, f))
    return res


def chamfer_loss(pred: np.ndarray, target: np.ndarray) -> float:
    """pytorch3d chamfer_distance defaults on [B,N,3], [B,M,3] (models_refine.py:72), float64."""
    tot = 0.0
    for p, q in zip(pred, target):
        tot += np.mean(point_cloud_distance(p, q) ** 2) + np.mean(point_cloud_distance(q, p) ** 2)
    return tot / len(pred)
