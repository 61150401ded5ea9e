"""Evaluation metrics of lidiff/utils/metrics.py on the GPU.

Same classes, ``update(gt, pred)`` / ``compute()`` / ``reset()`` protocol and accumulated quantities as the
reference; clouds are passed as ``[N,3]`` tensors / arrays (or anything with a ``.points`` attribute, as open3d
point clouds have) instead of open3d geometries.  The nearest-neighbour distances of open3d
``compute_point_cloud_distance`` come from the exhaustive HIP search ``lidiff_nn_dist`` in float64; the
occupancy histograms of ``CompletionIoU`` use numpy's own bin edges and the voxel-hash kernel for the set sizes.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops


def _cloud(x, device) -> torch.Tensor:
    if hasattr(x, "points"):
        x = np.asarray(x.points)
    t = torch.as_tensor(x)
    return t.to(device=device, dtype=torch.float64)[:, :3].contiguous()


def point_cloud_distance(src, dst, device="cuda") -> torch.Tensor:
    """open3d ``src.compute_point_cloud_distance(dst)`` (metrics.py:68,128-129,150-153): float64 [N]."""
    d2, _ = ops.nn_dist(_cloud(src, device), _cloud(dst, device))
    return d2.sqrt()


class RMSE:
    """metrics.py:62-78 (mean distance prediction -> ground truth per scan)."""

    def __init__(self, device="cuda"):
        self.device = device
        self.dists = []

    def update(self, gt_pcd, pt_pcd):
        self.dists.append(float(point_cloud_distance(pt_pcd, gt_pcd, self.device).mean()))

    def reset(self):
        self.dists = []

    def compute(self):
        dist = np.array(self.dists)
        return dist.mean(), dist.std()


class ChamferDistance:
    """metrics.py:123-141 (symmetric mean nearest-neighbour distance, halved)."""

    def __init__(self, device="cuda"):
        self.device = device
        self.dists = []

    def update(self, gt_pcd, pt_pcd):
        pt_2_gt = point_cloud_distance(pt_pcd, gt_pcd, self.device).mean()
        gt_2_pt = point_cloud_distance(gt_pcd, pt_pcd, self.device).mean()
        self.dists.append(float((gt_2_pt + pt_2_gt) / 2))

    def reset(self):
        self.dists = []

    def compute(self):
        cdist = np.array(self.dists)
        return cdist.mean(), cdist.std()


class CompletionIoU:
    """metrics.py:80-121: occupancy IoU over [-50, 50]^3 at several voxel sizes.  The reference fills dense
    ``bins^3`` histograms; here the occupied cells are kept as sets (voxel-hash dedup) and
    tp = |G| + |P| - |G u P|."""

    def __init__(self, voxel_sizes=(0.5, 0.2, 0.1), device="cuda"):
        self.voxel_sizes = list(voxel_sizes)
        self.device = device
        self.conf_matrix = np.zeros((len(self.voxel_sizes), 3)).astype(np.uint64)

    def _cells(self, pts: torch.Tensor, edges: torch.Tensor, bins: int) -> torch.Tensor:
        # np.histogramdd binning: searchsorted(side='right') on numpy's own edges, last bin closed on the right
        ix = torch.searchsorted(edges, pts.t().contiguous(), right=True).t()
        ix = ix - (pts == edges[-1]).to(ix.dtype)
        keep = ((ix >= 1) & (ix <= bins)).all(dim=1)
        ix = (ix[keep] - 1).to(torch.int32)
        return torch.cat([torch.zeros((ix.shape[0], 1), dtype=torch.int32, device=ix.device), ix], dim=1).contiguous()

    def _count(self, cells: torch.Tensor) -> int:
        if cells.shape[0] == 0:
            return 0
        status = torch.zeros(1, dtype=torch.int32, device=cells.device)
        uniq = ops.vox_unique(cells, status)[0]
        if int(status.item()) != 0:
            raise RuntimeError("voxel index outside the hash-key range")
        return int(uniq.shape[0])

    def update(self, gt, pred):
        max_range = 50.
        g, q = _cloud(gt, self.device), _cloud(pred, self.device)
        for i, vsize in enumerate(self.voxel_sizes):
            bins = int(2 * max_range / vsize)
            edges = torch.from_numpy(np.linspace(-max_range, max_range, bins + 1)).to(self.device)
            cg, cq = self._cells(g, edges, bins), self._cells(q, edges, bins)
            n_g, n_q, n_u = self._count(cg), self._count(cq), self._count(torch.cat([cg, cq], dim=0))
            tp = n_g + n_q - n_u
            self.conf_matrix[i][0] += np.uint64(tp)
            self.conf_matrix[i][1] += np.uint64(n_g - tp)      # fn
            self.conf_matrix[i][2] += np.uint64(n_q - tp)      # fp

    def compute(self):
        res_vsizes = {}
        for i, vsize in enumerate(self.voxel_sizes):
            tp, fn, fp = (float(v) for v in self.conf_matrix[i])
            res_vsizes[vsize] = tp / (tp + fn + fp + 1e-15)
        return res_vsizes

    def reset(self):
        self.conf_matrix = np.zeros((len(self.voxel_sizes), 3)).astype(np.uint64)


class PrecisionRecall:
    """metrics.py:143-223: precision / recall / F-score (percent) over a threshold sweep."""

    def __init__(self, min_t, max_t, num, device="cuda"):
        self.device = device
        self.thresholds = np.linspace(min_t, max_t, num)
        self.reset()

    def update(self, gt_pcd, pt_pcd):
        dist_pt_2_gt = point_cloud_distance(pt_pcd, gt_pcd, self.device)      # precision: predicted -> ground truth
        dist_gt_2_pt = point_cloud_distance(gt_pcd, pt_pcd, self.device)      # recall: ground truth -> predicted
        th = torch.from_numpy(self.thresholds).to(self.device)
        n_p = (dist_pt_2_gt[None, :] < th[:, None]).sum(dim=1).tolist()
        n_r = (dist_gt_2_pt[None, :] < th[:, None]).sum(dim=1).tolist()
        for t, cp, cr in zip(self.thresholds, n_p, n_r):
            p = 100 / len(dist_pt_2_gt) * cp
            r = 100 / len(dist_gt_2_pt) * cr
            f = 0 if p == 0 or r == 0 else 2 * p * r / (p + r)
            self.pr_dict[t].append(p)
            self.re_dict[t].append(r)
            self.f1_dict[t].append(f)

    def reset(self):
        self.pr_dict, self.re_dict, self.f1_dict = ({t: [] for t in self.thresholds} for _ in range(3))

    def find_nearest_threshold(self, value):
        return self.thresholds[np.abs(self.thresholds - value).argmin()]

    def _means(self, t):
        return tuple(sum(d[t]) / len(d[t]) for d in (self.pr_dict, self.re_dict, self.f1_dict))

    def compute_at_threshold(self, threshold):
        t = self.find_nearest_threshold(threshold)
        return (*self._means(t), t)

    def compute_at_all_thresholds(self):
        pr, re, f1 = zip(*(self._means(t) for t in self.thresholds))
        return list(pr), list(re), list(f1)

    def compute_auc(self):
        """Simpson areas under the three curves, normalised by a perfect predictor's (metrics.py:199-217)."""
        from scipy.integrate import simpson
        dx = self.thresholds[1] - self.thresholds[0]
        unit = simpson(np.ones_like(self.thresholds), dx=dx)
        return tuple(simpson(np.asarray(c), dx=dx) / unit for c in self.compute_at_all_thresholds())
