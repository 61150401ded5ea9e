"""CPU tests of the metrics oracle: hand-computed known answers, and the set-based occupancy counts against
the dense np.histogramdd formulation the reference uses (utils/metrics.py:86-106)."""
import numpy as np

from oracle import metrics_cpu as om


def test_point_cloud_distance_known_answers():
    src = np.array([[0.0, 0, 0], [1, 1, 1], [10, 0, 0]])
    dst = np.array([[0.0, 0, 1], [9, 0, 0]])
    np.testing.assert_allclose(om.point_cloud_distance(src, dst), [1.0, np.sqrt(2.0), 1.0])
    assert om.rmse_update(dst, src) == (1.0 + np.sqrt(2.0) + 1.0) / 3
    # gt -> pt: (0,0,1) -> (0,0,0) = 1 ; (9,0,0) -> (10,0,0) = 1
    assert om.chamfer_update(dst, src) == ((1.0 + np.sqrt(2.0) + 1.0) / 3 + 1.0) / 2
    assert om.chamfer_loss(src[None], dst[None]) == (1 + 2 + 1) / 3 + (1 + 1) / 2


def test_precision_recall_known_answer():
    gt = np.array([[0.0, 0, 0], [1, 0, 0]])
    pt = np.array([[0.1, 0, 0], [5, 0, 0], [1, 0.2, 0], [0, 0, 0.05]])
    (p, r, f), = om.precision_recall_update(gt, pt, [0.15])
    assert p == 50.0 and r == 50.0 and f == 50.0


def test_completion_iou_counts_equal_dense_histograms():
    rng = np.random.default_rng(0)
    gt = rng.uniform(-52, 52, (4000, 3))
    gt[:3] = [[50.0, 50.0, 50.0], [-50.0, 0, 0], [10.0, 20.0, 50.0]]      # closed outer edge, open lower edge
    gt[3:300] = np.round(gt[3:300])                                        # on inner edges of both grids
    pred = np.concatenate([gt[::3] + rng.normal(0, 0.4, (1334, 3)), rng.uniform(-52, 52, (300, 3))])
    sizes = (2.0, 1.0)
    got = om.completion_iou_counts(gt, pred, voxel_sizes=sizes)
    for i, v in enumerate(sizes):
        bins = int(100 / v)
        rg = ([-50., 50.],) * 3
        hg = np.histogramdd(gt, bins=bins, range=rg)[0].astype(bool)
        hp = np.histogramdd(pred, bins=bins, range=rg)[0].astype(bool)
        assert tuple(got[i]) == ((hg & hp).sum(), (hg & ~hp).sum(), (~hg & hp).sum())
