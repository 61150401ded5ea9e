"""The reference's OWN Python executed (tests/ref_exec.py): /root/reference/lidiff/models/minkunet.py and
tools/diff_completion_pipeline.py are imported unmodified and run

  (b) over the CPU oracle's ME / KeOps stand-ins (oracle/me_shim.py) -- here in the build container -- and compared
      with the functional restatement oracle/minkunet_cpu.py that every GPU parity test uses as truth.  This pins
      the HOST semantics the oracle restates (module tree, state-dict keys, op order, the cat((t, p)) quirk of
      minkunet.py:461, batch handling, the pipeline loop) to the reference's code instead of to a reading of it;
  (a) over the product's HIP kernels (lidiff_amd.compat.install()) -- on the GPU -- and compared with the fused
      product networks: the drop-in claim of INTEGRATION.md section 2a, executed.

Both need /root/reference (absent on the GPU box: (a) is skipped there, and the fixtures generated from (b) by
tests/golden/make_golden.py travel instead).  The ME-internal arithmetic stays restated: PARITY UNPINNED there.
"""
import os

import numpy as np
import pytest
import torch

import ref_exec
from conftest import GOLDEN, build_seeded_models, diffusion_state_dict, small_scene

needs_reference = pytest.mark.skipif(not ref_exec.have_reference(), reason="/root/reference is not mounted")


def _ref_models(backend, device="cpu"):
    """The reference's three classes carrying the seeded weights of conftest.build_seeded_models."""
    enc, unet, refine = build_seeded_models(42)
    ref, _ = ref_exec.reference_minkunet(backend, device)
    r_enc = ref.MinkGlobalEnc(in_channels=3, out_channels=96)
    r_unet = ref.MinkUNetDiff(in_channels=3, out_channels=96)
    r_ref = ref.MinkUNet(in_channels=3, out_channels=18)
    r_enc.load_state_dict(enc.state_dict(), strict=True)
    r_unet.load_state_dict(unet.state_dict(), strict=True)
    r_ref.load_state_dict(refine.state_dict(), strict=True)
    return (enc, unet, refine), tuple(m.eval().to(device) for m in (r_enc, r_unet, r_ref)), ref


@needs_reference
def test_state_dict_keys_and_shapes_match_reference_classes():
    """Checkpoint compatibility: same parameter / buffer names and shapes, in the same order."""
    (enc, unet, refine), (r_enc, r_unet, r_ref), _ = _ref_models("oracle")
    for ours, theirs in ((enc, r_enc), (unet, r_unet), (refine, r_ref)):
        a, b = ours.state_dict(), theirs.state_dict()
        assert list(a.keys()) == list(b.keys())
        assert all(a[k].shape == b[k].shape and a[k].dtype == b[k].dtype for k in a)
    assert sum(p.numel() for p in r_unet.parameters()) + sum(p.numel() for p in r_enc.parameters()) == 32672467
    assert sum(p.numel() for p in r_ref.parameters()) == 21722926


def _oracle_fields(points, batch_col_divided=True):
    from oracle import minkunet_cpu as net
    pts = torch.from_numpy(points)[None] if points.ndim == 2 else torch.from_numpy(points)
    return net.points_to_field(pts, divide_batch_col=batch_col_divided)


def _shim_field(me_mod, cpu_field):
    return me_mod.TensorField(features=cpu_field.F, coordinates=cpu_field.coords_f,
                              quantization_mode=me_mod.SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE,
                              minkowski_algorithm=me_mod.MinkowskiAlgorithm.SPEED_OPTIMIZED)


@needs_reference
@pytest.mark.parametrize("batch", [1, 2])
def test_reference_networks_over_oracle_shim_equal_functional_oracle(batch):
    """minkunet.py:83-141, 144-497, 500-619 executed vs oracle/minkunet_cpu.py (same CPU kernels underneath, so the
    tolerance only covers torch's nn.Linear / BatchNorm1d against their functional forms)."""
    from oracle import minkunet_cpu as net
    (enc, unet, refine), (r_enc, r_unet, r_ref), ref = _ref_models("oracle")
    _, mods = ref_exec.reference_minkunet("oracle")
    me_mod = mods["MinkowskiEngine"]
    sd = diffusion_state_dict(enc, unet)
    scan, noisy = small_scene(seed=5, n=1200)
    if batch == 2:
        scan2, noisy2 = small_scene(seed=6, n=1200)
        scan, noisy = np.stack([scan, scan2]), np.stack([noisy, noisy2])
    t = torch.tensor([500, 120][:batch])
    # DiffusionPoints.points_to_tensor (models.py:162-178) leaves the batch column alone -- needed for batch 2
    of, oc = _oracle_fields(noisy, batch == 1), _oracle_fields(scan, batch == 1)
    with torch.no_grad():
        want = net.denoise_forward(sd, of, of.sparse(), oc, t)
        xf, cf = _shim_field(me_mod, _oracle_fields(noisy, batch == 1)), _shim_field(me_mod, _oracle_fields(scan, batch == 1))
        got = r_unet(xf, xf.sparse(), r_enc(cf), t.float() if False else t).reshape(t.shape[0], -1, 3)
        assert torch.allclose(got, want, rtol=1e-5, atol=1e-5), (got - want).abs().max()
        # refinement network (models_refine.py:53-76 divides the batch column: ids 0, 20, ...)
        rf = _oracle_fields(noisy, True)
        want_r = net.unet_refine_forward(refine.state_dict(), rf)
        got_r = r_ref(_shim_field(me_mod, _oracle_fields(noisy, True)))
        assert torch.allclose(got_r, want_r, rtol=1e-5, atol=1e-5)


@needs_reference
def test_golden_unet_comes_from_the_reference_code():
    """tests/golden/unet_small.npz (consumed by the GPU tests test_golden_unet_cfg / test_golden_refine_unet) is
    what the reference's classfree_forward computes over the oracle shim (make_golden.py generates it that way)."""
    from golden.make_golden import reference_unet_outputs
    g = np.load(os.path.join(GOLDEN, "unet_small.npz"))
    eps, off = reference_unet_outputs()
    assert np.allclose(eps, g["eps"], rtol=1e-5, atol=1e-5) and np.allclose(off, g["refine"], rtol=1e-5, atol=1e-5)


def _write_checkpoints(tmp_path, enc, unet, refine):
    from lidiff_amd.pipeline import DEFAULT_HPARAMS
    diff = os.path.join(tmp_path, "diff_net.ckpt")
    ref = os.path.join(tmp_path, "refine_net.ckpt")
    torch.save({"state_dict": diffusion_state_dict(enc, unet), "hyper_parameters": DEFAULT_HPARAMS}, diff)
    torch.save({"state_dict": {"model_refine." + k: v for k, v in refine.state_dict().items()}}, ref)
    return diff, ref


@needs_reference
def test_reference_pipeline_loop_over_oracle_shim(tmp_path, monkeypatch):
    """tools/diff_completion_pipeline.py executed: DiffCompletion.__init__ (checkpoint loading, scheduler),
    points_to_tensor, completion_loop (3 DPM-Solver++ steps), postprocess_scan, refine_forward -- against the
    oracle's functional loop fed the same Gaussian draws."""
    from oracle import minkunet_cpu as net
    from oracle.dpm_solver import DpmSolverSdeOracle
    enc, unet, refine = build_seeded_models(42)
    diff_p, ref_p = _write_checkpoints(str(tmp_path), enc, unet, refine)
    monkeypatch.chdir(tmp_path)
    with ref_exec.cuda_calls_as("cpu"):
        pipe_mod, _ = ref_exec.reference_pipeline("oracle")
        pipe = pipe_mod.DiffCompletion(diff_p, ref_p, 3, 6.0)
        assert pipe.dpm_scheduler.timesteps.tolist() == [999, 666, 333]
        scan_np, noisy_np = small_scene(seed=9, n=800)
        scan = torch.from_numpy(scan_np).double()[None]              # float64 as preprocess_scan makes it (App. D.1)
        x_feats = torch.from_numpy(noisy_np).double()[None]
        x_full, x_cond = pipe.points_to_tensor(x_feats), pipe.points_to_tensor(scan)
        x_uncond = pipe.points_to_tensor(torch.zeros_like(scan))
        torch.manual_seed(123)
        got = pipe.completion_loop(scan, x_full, x_cond, x_uncond)
        # the oracle's loop with the same draws (torch's global CPU generator, float64, one draw per step)
        sd = diffusion_state_dict(enc, unet)
        o = DpmSolverSdeOracle()
        ts = o.set_timesteps(3)
        torch.manual_seed(123)
        x = x_feats.clone()
        with torch.no_grad():
            for t in ts:
                xf = net.points_to_field(x.float())
                eps = net.classfree_forward(sd, xf, net.points_to_field(scan.float()),
                                            net.points_to_field(torch.zeros(1, scan.shape[1], 3)), torch.tensor([int(t)]), w=6.0)
                inp = xf.F.reshape(1, -1, 3).double() - scan
                z = torch.randn(inp.shape, dtype=torch.float64)
                x = scan + torch.from_numpy(o.step(eps.double().numpy(), int(t), inp.numpy(), z.numpy()))
        want = net.points_to_field(x.float()).F.numpy()
        assert got.shape == want.shape and np.allclose(got, want, rtol=1e-5, atol=1e-5), np.abs(got - want).max()
        # post-filter + refinement (pipeline:107-138)
        post = pipe.postprocess_scan(got, scan)
        ours = __import__("lidiff_amd.pipeline", fromlist=["DiffCompletion"]).DiffCompletion.postprocess_scan(
            type("H", (), {"hparams": {"data": {"max_range": 50.0}}})(), got, scan)
        assert np.array_equal(post, ours)
        off = pipe.refine_forward(pipe.points_to_tensor(torch.from_numpy(post[None])))
        want_off = net.unet_refine_forward(refine.state_dict(), net.points_to_field(torch.from_numpy(post[None])))
        assert torch.allclose(off, want_off, rtol=1e-5, atol=1e-5)


@needs_reference
@pytest.mark.parametrize("seed", [11, 12, 13])
def test_reference_training_step_over_oracle_shim_equals_functional_oracle(seed):
    """models.py:180-217 executed: the reference's own DiffusionPoints.training_step (q_sample, points_to_tensor, the
    condition drop, forward through ITS minkunet.py in training mode, the three loss terms) over the oracle's ME stand-in,
    with torch autograd, against oracle.minkunet_cpu.training_loss -- the functional restatement every GPU training-parity
    test uses as truth -- on the same weights, batch and random draws (the draws are replayed from the same seed in the
    order the reference makes them: randn noise, randint t, rand for the drop).  Seeds cover both branches of the drop."""
    from lidiff_amd.pipeline import DEFAULT_HPARAMS
    from oracle import minkunet_cpu as net
    enc, unet, _ = build_seeded_models(42)
    ref, _ = ref_exec.reference_models("oracle")
    with ref_exec.cuda_calls_as("cpu"):
        mod = ref.DiffusionPoints(DEFAULT_HPARAMS)
        mod.partial_enc.load_state_dict(enc.state_dict(), strict=True)
        mod.model.load_state_dict(unet.state_dict(), strict=True)
        mod.train()
        scan, _ = small_scene(seed=seed, n=400)
        full = torch.from_numpy(np.stack([scan, scan[::-1].copy() + np.float32(0.37)]))
        part = full[:, :60].contiguous()
        batch = {"pcd_full": full, "pcd_part": part, "mean": torch.zeros(2, 3), "std": torch.ones(2, 3)}
        torch.manual_seed(seed)
        loss_ref = mod.training_step(batch, 0)
        params = dict(mod.named_parameters())
        names = [k for k in params if k.startswith(("partial_enc.", "model."))]
        grads_ref = torch.autograd.grad(loss_ref, [params[k] for k in names], allow_unused=True)
    # the same draws, in the reference's order (models.py:183,186,195)
    torch.manual_seed(seed)
    noise = torch.randn(full.shape)
    t = torch.randint(0, DEFAULT_HPARAMS["diff"]["t_steps"], size=(2,))
    drop = not bool(torch.rand(1) > DEFAULT_HPARAMS["train"]["uncond_prob"])
    sd = {k: v.detach().clone() for k, v in diffusion_state_dict(enc, unet).items()}
    for k, v in sd.items():
        v.requires_grad_(v.is_floating_point() and "running" not in k)
    loss_o, _ = net.training_loss(sd, full, part, noise, t, drop_condition=drop)
    grads_o = torch.autograd.grad(loss_o, [sd[k] for k in names], allow_unused=True)
    assert abs(float(loss_ref) - float(loss_o)) <= 1e-5 * abs(float(loss_o)), (float(loss_ref), float(loss_o), drop)
    compared = 0
    for k, a, b in zip(names, grads_ref, grads_o):
        if a is None or b is None:
            assert (a is None or float(a.abs().max()) == 0.0) and (b is None or float(b.abs().max()) == 0.0), k
            continue
        if drop and k.startswith("partial_enc."):
            continue                     # inf - inf on both sides (zero-variance BatchNorm of the zeroed condition)
        if float(b.norm()) <= 1e-7:
            continue
        compared += 1
        cos = float((a * b).sum() / (a.norm() * b.norm()))
        assert cos >= 0.999, (k, cos, drop)
    assert compared >= 100, compared


@needs_reference
def test_reference_refine_training_step_over_oracle_shim_equals_functional_oracle():
    """models_refine.py:53-76 executed: the reference's RefineDiffusion.training_step (voxelisation with the batch column
    divided too, its MinkUNet in training mode, 6 offsets per point, Chamfer loss) over the oracle's ME stand-in and a
    brute-force stand-in of pytorch3d's chamfer_distance, against oracle.minkunet_cpu.refine_training_loss (KD-tree
    nearest neighbours) -- the truth of the GPU refine-step parity test."""
    from lidiff_amd.diffusion import REFINE_HPARAMS
    from oracle import minkunet_cpu as net
    _, _, refine = build_seeded_models(42)
    ref, _ = ref_exec.reference_models_refine("oracle")
    scan, noisy = small_scene(seed=5, n=300)
    pcd_noise = torch.from_numpy(np.stack([noisy, noisy[::-1].copy()]))
    rng = np.random.default_rng(3)
    pcd_full = torch.from_numpy(np.stack([scan, scan[::-1].copy()]).astype(np.float32)
                                + 0.02 * rng.standard_normal((2,) + scan.shape).astype(np.float32))
    with ref_exec.cuda_calls_as("cpu"):
        mod = ref.RefineDiffusion(REFINE_HPARAMS)
        mod.model_refine.load_state_dict(refine.state_dict(), strict=True)
        mod.train()
        loss_ref = mod.training_step({"pcd_noise": pcd_noise, "pcd_full": pcd_full}, 0)
        params = dict(mod.model_refine.named_parameters())
        grads_ref = torch.autograd.grad(loss_ref, list(params.values()), allow_unused=True)
    sd = {k: v.detach().clone() for k, v in refine.state_dict().items()}
    for k, v in sd.items():
        v.requires_grad_(v.is_floating_point() and "running" not in k)
    loss_o = net.refine_training_loss(sd, pcd_noise, pcd_full, up_factor=REFINE_HPARAMS["train"]["up_factor"])
    grads_o = torch.autograd.grad(loss_o, [sd[k] for k in params], allow_unused=True)
    assert abs(float(loss_ref.detach()) - float(loss_o.detach())) <= 1e-5 * abs(float(loss_o.detach()))
    compared = 0
    for k, a, b in zip(params, grads_ref, grads_o):
        if a is None or b is None or float(b.norm()) <= 1e-9:
            continue
        compared += 1
        cos = float((a * b).sum() / (a.norm() * b.norm()))
        assert cos >= 0.999, (k, cos)
    assert compared >= 100, compared


@needs_reference
def test_reference_metrics_equal_the_metrics_oracle():
    """utils/metrics.py executed (RMSE, ChamferDistance, PrecisionRecall incl. its AUC, CompletionIoU) over an open3d
    stand-in whose compute_point_cloud_distance is a brute-force search: the same numbers as oracle/metrics_cpu.py (KD-tree
    search, set-based occupancy), which is the truth of tests/test_gpu_evaluation.py."""
    from oracle import metrics_cpu as om
    ref, o3d = ref_exec.reference_metrics()
    rng = np.random.default_rng(8)

    def cloud(a):
        c = o3d.geometry.PointCloud()
        c.points = o3d.utility.Vector3dVector(a)
        return c
    pairs = []
    for _ in range(3):
        gt = rng.uniform(-48, 48, (1500, 3)) * np.array([1.0, 1.0, 0.1])
        pt = np.concatenate([gt[::2] + rng.normal(0, 0.05, (750, 3)), rng.uniform(-48, 48, (100, 3))])
        pairs.append((gt, pt))
    rmse, cd, iou = ref.RMSE(), ref.ChamferDistance(), ref.CompletionIoU(voxel_sizes=[2.0, 1.0])
    pr = ref.PrecisionRecall(0.05, 0.1, 20)
    for gt, pt in pairs:
        rmse.update(cloud(gt), cloud(pt))
        cd.update(cloud(gt), cloud(pt))
        pr.update(cloud(gt), cloud(pt))
        iou.update(cloud(gt), cloud(pt))
    np.testing.assert_allclose(rmse.dists, [om.rmse_update(g, p) for g, p in pairs], rtol=1e-12)
    np.testing.assert_allclose(cd.dists, [om.chamfer_update(g, p) for g, p in pairs], rtol=1e-12)
    want = [om.precision_recall_update(g, p, pr.thresholds) for g, p in pairs]
    for j, t in enumerate(pr.thresholds):
        np.testing.assert_allclose(pr.pr_dict[t], [w[j][0] for w in want], rtol=1e-12)
        np.testing.assert_allclose(pr.re_dict[t], [w[j][1] for w in want], rtol=1e-12)
        np.testing.assert_allclose(pr.f1_dict[t], [w[j][2] for w in want], rtol=1e-12)
    counts = sum(om.completion_iou_counts(g, p, voxel_sizes=(2.0, 1.0)).astype(np.int64) for g, p in pairs)
    assert np.array_equal(iou.conf_matrix.astype(np.int64), counts)
    assert all(np.isfinite(v) for v in pr.compute_auc())


@needs_reference
def test_reference_preprocess_scan_equals_product_and_fixture(tmp_path, monkeypatch, fps_scan):
    """preprocess_scan (pipeline:92-105) executed on the bundled scan with the FPS oracle behind the open3d stand-in
    reproduces the committed 18 000-point fixture (x10), i.e. the input every bench / parity run starts from."""
    from lidiff_amd.pipeline import read_ply_points
    enc, unet, refine = build_seeded_models(42)
    diff_p, ref_p = _write_checkpoints(str(tmp_path), enc, unet, refine)
    monkeypatch.chdir(tmp_path)
    pts = read_ply_points(os.path.join(ref_exec.REF_PKG, "Datasets", "test", "000123.ply"))
    assert pts.shape == (125773, 3)
    with ref_exec.cuda_calls_as("cpu"):
        pipe_mod, _ = ref_exec.reference_pipeline("oracle")
        pipe = pipe_mod.DiffCompletion(diff_p, ref_p, 50, 6.0)
        pipe.hparams["data"]["num_points"] = 20000                     # 2 000 samples: seconds, same code path
        scan = pipe.preprocess_scan(pts)
    assert scan.shape == (1, 20000, 3) and scan.dtype == torch.float64
    assert np.array_equal(scan[0, :2000].numpy().astype(np.float32), fps_scan[:2000])
    assert torch.equal(scan[0, :2000], scan[0, 18000:])


# ----------------------------------------------------------------------------------------
# (a) the reference file on the HIP kernels
# ----------------------------------------------------------------------------------------
@needs_reference
@pytest.mark.gpu
def test_reference_networks_over_hip_shim_equal_fused_product(device):
    from test_gpu_network import to_field
    (enc, unet, refine), (r_enc, r_unet, r_ref), _ = _ref_models("hip", device)
    enc, unet, refine = enc.to(device), unet.to(device), refine.to(device)
    scan, noisy = small_scene(seed=5, n=4000)
    t = torch.tensor([500], device=device)
    with torch.no_grad():
        xf, cf = to_field(noisy, device), to_field(scan, device)
        want = unet(xf, xf.sparse(), enc(cf), t)
        xr, cr = to_field(noisy, device), to_field(scan, device)
        got = r_unet(xr, xr.sparse(), r_enc(cr), t)
        assert torch.allclose(got, want, rtol=1e-5, atol=3e-6), (got - want).abs().max()
        assert torch.allclose(r_ref(to_field(noisy, device)), refine(to_field(noisy, device)), rtol=1e-5, atol=3e-6)
