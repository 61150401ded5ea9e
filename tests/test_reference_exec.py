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
        assert torch.allclose(got, want, rtol=1e-3, atol=2e-3), (got - want).abs().max()
        assert torch.allclose(r_ref(to_field(noisy, device)), refine(to_field(noisy, device)), rtol=1e-3, atol=2e-3)
