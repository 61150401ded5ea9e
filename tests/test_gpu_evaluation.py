"""GPU parity tests of the evaluation / Chamfer-loss row: the exhaustive HIP
nearest-neighbour search through the C ABI against the KD-tree oracle, and the metric classes against the
restated utils/metrics.py arithmetic.  Indices bit-exact (ties included), float64 distances to rtol 1e-12,
float32 distances to rtol 1e-5 (the device forms (a-b)^2 in float32, the oracle in float64).
"""
import numpy as np
import pytest
import torch

from oracle import metrics_cpu as om

pytestmark = pytest.mark.gpu


def clouds(n, m, seed, scale=20.0):
    rng = np.random.default_rng(seed)
    return rng.normal(0, scale, (n, 3)), rng.normal(0, scale, (m, 3))


def brute(a, b):
    d = ((a[:, None, :] - b[None, :, :]) ** 2).sum(-1)
    return d.min(1), d.argmin(1)


@pytest.mark.parametrize("n,m", [(1, 1), (7, 3), (513, 1025), (3000, 2049), (40, 70000)])
def test_nn_dist_float64_matches_brute_force(device, n, m):
    from lidiff_amd import ops
    a, b = clouds(n, m, n + m)
    d2, idx = ops.nn_dist(torch.from_numpy(a).to(device), torch.from_numpy(b).to(device))
    want_d2, want_idx = brute(a, b)
    assert np.array_equal(idx.cpu().numpy(), want_idx)
    np.testing.assert_allclose(d2.cpu().numpy(), want_d2, rtol=1e-12, atol=0)


def test_nn_dist_ties_take_the_lowest_row(device):
    """Integer lattice clouds: many exact ties, also across the LDS tiles and the b splits."""
    from lidiff_amd import ops
    rng = np.random.default_rng(3)
    a = rng.integers(-4, 5, (700, 3)).astype(np.float32)
    b = np.tile(rng.integers(-4, 5, (1500, 3)).astype(np.float32), (3, 1))      # every row three times
    d2, idx = ops.nn_dist(torch.from_numpy(a).to(device), torch.from_numpy(b).to(device))
    want_d2, want_idx = brute(a.astype(np.float64), b.astype(np.float64))
    assert np.array_equal(idx.cpu().numpy(), want_idx)                          # np.argmin = first minimum
    assert np.array_equal(d2.cpu().numpy(), want_d2.astype(np.float32))        # small integers: exact in fp32


def test_nn_dist_float32_and_errors(device):
    from lidiff_amd import ops
    a, b = clouds(5000, 4000, 11)
    ta, tb = torch.from_numpy(a).float().to(device), torch.from_numpy(b).float().to(device)
    d2, idx = ops.nn_dist(ta, tb)
    want = om.point_cloud_distance(ta.cpu().numpy(), tb.cpu().numpy()) ** 2
    np.testing.assert_allclose(d2.cpu().numpy(), want, rtol=1e-5, atol=1e-6)
    picked = ((ta - tb[idx]) ** 2).sum(1).cpu().numpy()
    np.testing.assert_allclose(picked, want, rtol=1e-5, atol=1e-6)
    e2, eidx = ops.nn_dist(ta[:0], tb)
    assert e2.shape == (0,) and eidx.shape == (0,)
    with pytest.raises(RuntimeError):
        ops.nn_dist(ta, tb[:0])
    with pytest.raises(TypeError):
        ops.nn_dist(ta, tb.double())
    with pytest.raises(RuntimeError):
        ops.nn_dist(ta.cpu(), tb.cpu())


def test_nn_dist_full_size_properties(device, fps_scan):
    """BASELINE size (180k x 180k): too large for the oracle; check the size-independent properties -- a cloud
    matches itself at distance 0 with the identity index; for a shifted copy no distance exceeds the shift."""
    from lidiff_amd import ops
    pts = torch.from_numpy(np.tile(fps_scan, (10, 1))).to(device).double()
    pts = pts + torch.arange(10, device=device).repeat_interleave(fps_scan.shape[0])[:, None] * 1e-3   # distinct rows
    d2, idx = ops.nn_dist(pts, pts)
    assert float(d2.max()) == 0.0
    assert torch.equal(idx, torch.arange(pts.shape[0], device=device))
    shift = torch.tensor([0.03, -0.02, 0.01], dtype=pts.dtype, device=device)
    d2s, idxs = ops.nn_dist(pts + shift, pts)
    assert float(d2s.max()) <= float((shift ** 2).sum()) * (1 + 1e-9)        # float64: (p + s) - p = s to 1e-13
    picked = ((pts + shift - pts[idxs]) ** 2).sum(1)
    assert torch.allclose(picked, d2s, rtol=1e-12, atol=0)


def test_metric_classes_vs_oracle(device):
    from lidiff_amd import evaluation as ev
    rng = np.random.default_rng(5)
    rmse, cd, iou, pr = ev.RMSE(), ev.ChamferDistance(), ev.CompletionIoU(), ev.PrecisionRecall(0.05, 1.0, 20)
    want_rmse, want_cd, want_pr = [], [], []
    want_conf = np.zeros((3, 3), dtype=np.uint64)
    for scan in range(3):
        gt = rng.uniform(-55, 55, (6000, 3))                   # some points outside the +-50 m histogram range
        gt[:4] = [[50.0, 0, 0], [-50.0, 1, 1], [0, 50.0, 49.99], [50.0, 50.0, 50.0]]     # on the outer edges
        gt[4:200] = np.round(gt[4:200] * 2) / 2                # exactly on inner bin edges of the 0.5 m grid
        pred = np.concatenate([gt[::2] + rng.normal(0, 0.15, (3000, 3)), rng.uniform(-55, 55, (500, 3))])
        for m in (rmse, cd, iou, pr):
            m.update(torch.from_numpy(gt), pred)               # tensors and arrays are both accepted
        want_rmse.append(om.rmse_update(gt, pred))
        want_cd.append(om.chamfer_update(gt, pred))
        want_conf += om.completion_iou_counts(gt, pred)
        want_pr.append(om.precision_recall_update(gt, pred, pr.thresholds))
    np.testing.assert_allclose(rmse.compute(), (np.mean(want_rmse), np.std(want_rmse)), rtol=1e-12)
    np.testing.assert_allclose(cd.compute(), (np.mean(want_cd), np.std(want_cd)), rtol=1e-12)
    assert np.array_equal(iou.conf_matrix, want_conf)          # integer counts: bit-exact
    res = iou.compute()
    for i, v in enumerate(iou.voxel_sizes):
        tp, fn, fp = (float(x) for x in want_conf[i])
        assert res[v] == tp / (tp + fn + fp + 1e-15)
    got = pr.compute_at_all_thresholds()
    for k in range(3):
        np.testing.assert_allclose(got[k], np.mean([[row[k] for row in scan] for scan in want_pr], axis=0), rtol=1e-12)
    p, r, f, t = pr.compute_at_threshold(0.31)
    assert t == pr.thresholds[np.abs(pr.thresholds - 0.31).argmin()]
    assert all(0 <= a <= 100 for a in pr.compute_auc())          # percentages, as in the reference
    iou.reset(), rmse.reset()
    assert iou.conf_matrix.sum() == 0 and rmse.dists == []


def test_chamfer_loss_value_and_gradient(device):
    """models_refine.py:72: value against the float64 oracle, gradient against a dense torch formulation."""
    from lidiff_amd.diffusion import chamfer_distance
    rng = np.random.default_rng(9)
    pred = torch.from_numpy(rng.normal(0, 3, (2, 900, 3))).float().to(device).requires_grad_(True)
    target = torch.from_numpy(rng.normal(0, 3, (2, 400, 3))).float().to(device)
    loss = chamfer_distance(pred, target)
    want = om.chamfer_loss(pred.detach().cpu().numpy().astype(np.float64), target.cpu().numpy().astype(np.float64))
    assert abs(float(loss) - want) <= 1e-5 * want
    loss.backward()
    ref_in = pred.detach().clone().requires_grad_(True)
    d = torch.cdist(ref_in, target) ** 2                       # [B,N,M]
    ref = (d.min(dim=2).values.mean(dim=1) + d.min(dim=1).values.mean(dim=1)).mean()
    ref.backward()
    assert torch.allclose(pred.grad, ref_in.grad, rtol=1e-3, atol=1e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_nn_dist_grid_is_bit_identical_to_the_exhaustive_search(device, fps_scan, dtype):
    """lidiff_nn_dist_grid (uniform grid over the searched cloud, shells of cells nearest first) against
    lidiff_nn_dist: the same d2 BITS and the same indices -- on a scan-like pair at the refinement loss's scale (216 000
    predicted against 120 000 target points), uniform clouds, queries far from every target point (the exhaustive leftover
    pass), exact ties (integer lattices: the lowest index must win across cells and shells), cells far too small (cell indices
    leave the key range: everything falls back) and far too large (one cell), one target point, NaN queries."""
    from lidiff_amd import ops
    rng = np.random.default_rng(3)
    base = np.tile(fps_scan[:12000], (10, 1))
    target = base + rng.normal(0, 0.03, base.shape)
    pred = np.repeat(base[::10][:, None, :] + 0.0, 18, axis=1).reshape(-1, 3) + rng.normal(0, 0.2, (18 * 12000, 3))
    lattice_a = rng.integers(-6, 7, (5000, 3)).astype(np.float64)
    lattice_b = rng.integers(-6, 7, (4000, 3)).astype(np.float64)
    far = np.concatenate([rng.normal(0, 5, (3000, 3)), rng.normal(0, 5, (300, 3)) + 400.0])
    nan_q = rng.normal(0, 5, (100, 3))
    nan_q[7, 1] = np.nan
    cases = [("scan", pred, target, 0.5), ("scan small cells", pred[:20000], target, 0.1), ("uniform", rng.uniform(-30, 30, (50000, 3)),
             rng.uniform(-30, 30, (40000, 3)), 0.5), ("far queries", far, rng.normal(0, 5, (20000, 3)), 0.5),
             ("ties", lattice_a, lattice_b, 1.0), ("ties, cell 2.5", lattice_a, lattice_b, 2.5),
             ("key range", rng.normal(0, 5, (2000, 3)), rng.normal(0, 5, (3000, 3)), 1e-5),
             ("one cell", rng.normal(0, 5, (2000, 3)), rng.normal(0, 5, (3000, 3)), 1e4),
             ("one target", rng.normal(0, 5, (500, 3)), rng.normal(0, 5, (1, 3)), 0.5), ("nan", nan_q, rng.normal(0, 5, (900, 3)), 0.5)]
    for name, a, b, cell in cases:
        ta, tb = torch.from_numpy(a).to(dtype).to(device), torch.from_numpy(b).to(dtype).to(device)
        want_d, want_i = ops.nn_dist(ta, tb, grid=False)
        got_d, got_i = ops.nn_dist(ta, tb, grid=True, cell=cell)
        assert torch.equal(got_i, want_i), (name, int((got_i != want_i).sum()))
        assert torch.equal(got_d.view(torch.int32 if dtype == torch.float32 else torch.int64),
                           want_d.view(torch.int32 if dtype == torch.float32 else torch.int64)), name
    # the loss itself at >= 100k points: same value bit for bit whichever search runs, gradient included
    from lidiff_amd.diffusion import chamfer_distance
    p = torch.from_numpy(pred[None, :108000]).float().to(device).requires_grad_(True)
    t = torch.from_numpy(target[None]).float().to(device)
    outs = []
    for min_pairs in (1, 1 << 62):
        ops.NN_GRID_MIN_PAIRS, keep = min_pairs, ops.NN_GRID_MIN_PAIRS
        try:
            p.grad = None
            loss = chamfer_distance(p, t)
            loss.backward()
            outs.append((loss.detach().clone(), p.grad.clone()))
        finally:
            ops.NN_GRID_MIN_PAIRS = keep
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
