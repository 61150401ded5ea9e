"""GPU parity tests at the ME-API / network / sampling-loop level (pytest -m gpu).

Tolerances: a single conv is within rtol/atol 1e-4 of the oracle (measured worst on the bench maps: 4.7e-6 on outputs of
scale 12); whole-network outputs (49 convs deep, O(0.1-1)) are compared at NET_RTOL / NET_ATOL below, which are <= 10x the
errors measured on the device; every test records what it achieved (conftest.record_parity).
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, build_seeded_models, diffusion_state_dict, noisy_scan_points, record_parity, small_scene
from oracle import me_cpu as me
from oracle import minkunet_cpu as net
from oracle.dpm_solver import DpmSolverSdeOracle

pytestmark = pytest.mark.gpu

# Whole-network tolerances, set from the errors MEASURED on the MI355X (profiles/r03_parity_errors.txt): one CFG step on the
# 180 000-point scan has max |eps error| 3.4e-7 .. 6.3e-7 at the four trajectory positions (|eps| up to 0.17 .. 0.29) and the
# teacher-forced T = 50 loop 3.3e-7 in eps / 7.9e-6 m in the points -- the bars are <= 10x those (round 2 ran with 1e-3 / 2e-3).
NET_RTOL, NET_ATOL = 1e-5, 3e-6
X_ATOL = 8e-5        # metres: points after one teacher-forced DPM-Solver++ update


@pytest.fixture(scope="module")
def models(device):
    enc, unet, refine = build_seeded_models(42)
    sd = diffusion_state_dict(enc, unet)
    return enc.to(device), unet.to(device), refine.to(device), sd


def to_field(points_np, device, divide_batch=True):
    """Device TensorField whose float coordinates were rounded ON THE CPU exactly as the oracle
    rounds them: torch.round(x / 0.05) differs between CPU (true division) and GPU for ~5 ppm of
    inputs, and parity tests must feed both sides identical voxel coordinates."""
    import lidiff_amd.MinkowskiEngine as ME
    pts = torch.from_numpy(points_np)[None] if points_np.ndim == 2 else torch.from_numpy(points_np)
    cpu = net.points_to_field(pts, divide_batch_col=divide_batch)
    return ME.TensorField(features=cpu.F.to(device), coordinates=cpu.coords_f.to(device),
                          quantization_mode=ME.SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE,
                          minkowski_algorithm=ME.MinkowskiAlgorithm.SPEED_OPTIMIZED, device=device)


def test_me_api_field_sparse_slice(device):
    import lidiff_amd.MinkowskiEngine as ME
    scan, noisy = small_scene()
    f = to_field(noisy, device)
    s = f.sparse()
    of = net.points_to_field(torch.from_numpy(noisy)[None])
    os_ = of.sparse()
    assert s.C.dtype == torch.int32 and np.array_equal(s.C.cpu().numpy(), os_.C)
    assert np.array_equal(f.inverse_mapping.cpu().numpy(), of.inverse)
    assert torch.allclose(s.F.cpu(), os_.F, atol=1e-5)
    back = s.slice(f)
    assert torch.allclose(back.F.cpu(), os_.F[torch.from_numpy(of.inverse)], atol=1e-5)
    assert torch.equal((s * 2.0).F, s.F * 2.0) and torch.equal((s + s).F, s.F * 2)
    assert ME.cat(s, s).F.shape[1] == 6
    other = to_field(scan, device).sparse()
    with pytest.raises(RuntimeError):
        ME.cat(s, other)
    with pytest.raises(RuntimeError):
        ME.TensorField(features=torch.zeros(4, 3), coordinates=torch.zeros(4, 4))     # CPU tensors: no fallback


def test_me_modules_unfused_vs_oracle(device):
    """MinkowskiConvolution / Transpose / BatchNorm / ReLU as plain modules (reference op order)."""
    import lidiff_amd.MinkowskiEngine as ME
    torch.manual_seed(0)
    _, noisy = small_scene(seed=9)
    conv3 = ME.MinkowskiConvolution(3, 32, kernel_size=3, stride=1, dimension=3)
    down = ME.MinkowskiConvolution(32, 32, kernel_size=2, stride=2, dimension=3)
    res1 = ME.MinkowskiConvolution(32, 64, kernel_size=1, stride=1, dimension=3)
    up = ME.MinkowskiConvolutionTranspose(64, 32, kernel_size=2, stride=2, dimension=3)
    bn = ME.MinkowskiBatchNorm(32).eval()
    assert conv3.kernel.shape == (27, 3, 32) and res1.kernel.shape == (32, 64) and up.kernel.shape == (8, 64, 32)
    mods = [m.to(device) for m in (conv3, down, res1, up, bn)]
    with torch.no_grad():
        x = to_field(noisy, device).sparse()
        y0 = ME.MinkowskiReLU()(bn(conv3(x)))
        y1 = down(y0)
        y2 = res1(y1)
        y3 = up(y2)
        assert y3.tensor_stride == 1 and y1.tensor_stride == 2
        cat = ME.cat(y3, y0)
    ox = net.points_to_field(torch.from_numpy(noisy)[None]).sparse()
    sd = {k: v.cpu() for k, v in bn.state_dict().items()}
    o0 = me.conv(ox, conv3.kernel.detach().cpu(), 3, 1)
    o0 = o0.replace(torch.relu(me.batch_norm_eval(o0.F, sd["bn.weight"], sd["bn.bias"], sd["bn.running_mean"],
                                                  sd["bn.running_var"])))
    o1 = me.conv(o0, down.kernel.detach().cpu(), 2, 2)
    o2 = me.conv(o1, res1.kernel.detach().cpu(), 1, 1)
    o3 = me.conv_transpose(o2, up.kernel.detach().cpu(), 2, 2)
    assert np.array_equal(y1.C.cpu().numpy(), o1.C)
    for got, want in ((y0, o0), (y1, o1), (y2, o2), (y3, o3)):
        assert torch.allclose(got.F.cpu(), want.F, rtol=1e-4, atol=1e-4)
    assert cat.F.shape == (o0.F.shape[0], 64)


def run_cfg(models, device, scan, noisy, fused, t_val=500):
    from lidiff_amd import minkunet as product
    enc, unet, _, _ = models
    with torch.no_grad(), product.fusion(fused):
        xf, cf, uf = to_field(noisy, device), to_field(scan, device), to_field(np.zeros_like(scan), device)
        t = torch.tensor([t_val], device=device)
        xs = xf.sparse()
        e_c = unet(xf, xs, enc(cf), t)
        e_u = unet(xf, xs, enc(uf), t)
        xf.coordinate_manager.check()
    return (e_u + 6.0 * (e_c - e_u)).reshape(1, -1, 3)


def test_golden_unet_cfg(device, models):
    g = np.load(os.path.join(GOLDEN, "unet_small.npz"))
    want = torch.from_numpy(g["eps"])
    for fused in (False, True):
        got = run_cfg(models, device, g["scan"], g["noisy"], fused).cpu()
        err = (got - want).abs().max().item()
        record_parity("golden_unet_cfg", fused=int(fused), max_abs_err=err, max_abs=want.abs().max().item())
        assert torch.allclose(got, want, rtol=NET_RTOL, atol=NET_ATOL), f"fused={fused}: max err {err}"


def test_golden_refine_unet(device, models):
    g = np.load(os.path.join(GOLDEN, "unet_small.npz"))
    refine = models[2]
    from lidiff_amd import minkunet as product
    for fused in (False, True):
        with torch.no_grad(), product.fusion(fused):
            got = refine(to_field(g["noisy"], device)).cpu()
        assert got.shape == (2000, 18)
        record_parity("golden_refine_unet", fused=int(fused), max_abs_err=(got - torch.from_numpy(g["refine"])).abs().max().item())
        assert torch.allclose(got, torch.from_numpy(g["refine"]), rtol=NET_RTOL, atol=NET_ATOL), f"fused={fused}"


def test_unet_batch2_vs_oracle(device, models):
    """B=2 (the training shape): batch-grouped rows, per-batch timestep embedding, match across
    batches.  Uses models.py's points_to_tensor convention (batch column not divided)."""
    enc, unet, _, sd = models
    s0, n0 = small_scene(seed=1, n=1000)
    s1, n1 = small_scene(seed=2, n=1000)
    noisy, scan = np.stack([n0, n1]), np.stack([s0, s1])
    t = torch.tensor([100, 900])
    with torch.no_grad():
        xf, cf = to_field(noisy, device, divide_batch=False), to_field(scan, device, divide_batch=False)
        from lidiff_amd import minkunet as product
        for fused in (False, True):
            with product.fusion(fused):
                got = unet(xf, xf.sparse(), enc(cf), t.to(device)).cpu()
            oxf = net.points_to_field(torch.from_numpy(noisy), divide_batch_col=False)
            ocf = net.points_to_field(torch.from_numpy(scan), divide_batch_col=False)
            want = net.denoise_forward(sd, oxf, oxf.sparse(), ocf, t).reshape(-1, 3)
            record_parity("unet_batch2", fused=int(fused), max_abs_err=(got - want).abs().max().item(), max_abs=want.abs().max().item())
            assert torch.allclose(got, want, rtol=NET_RTOL, atol=NET_ATOL), f"fused={fused}"


def test_fused_equals_unfused_realistic_sparsity(device, models, fps_scan):
    """36k points of the bundled scan at sigma=0.3: the fused execution plan (epilogue fusion,
    split-input convs, commuted MLPs, cached match) against the reference op order."""
    noisy = noisy_scan_points(fps_scan[:3600], 0.3, 0)
    scan = np.tile(fps_scan[:3600], (10, 1))
    a = run_cfg(models, device, scan, noisy, fused=True)
    b = run_cfg(models, device, scan, noisy, fused=False)
    record_parity("fused_vs_unfused_36k", max_abs_diff=(a - b).abs().max().item(), max_abs=b.abs().max().item())
    assert torch.allclose(a, b, rtol=NET_RTOL, atol=NET_ATOL), (a - b).abs().max().item()


def test_scheduler_on_device_vs_oracle(device):
    """Device scheduler vs the oracle evaluated LIVE on this machine (tight: both read the same
    fp32 schedule tables), and vs the committed golden trajectory (loose: the fp32 tables
    themselves -- torch.linspace/cumprod/log on the CPU -- differ by an ulp between the machine
    that generated the fixture and this one, which a 50-step trajectory amplifies to ~1e-5)."""
    from lidiff_amd.schedulers import DPMSolverMultistepScheduler
    g = np.load(os.path.join(GOLDEN, "dpm_trajectory.npz"))
    for n in (50, 8, 1):
        s = DPMSolverMultistepScheduler(num_train_timesteps=1000, beta_start=3.5e-5, beta_end=0.007,
                                        beta_schedule="linear", algorithm_type="sde-dpmsolver++", solver_order=2)
        s.set_timesteps(n)
        s.to(device)
        o = DpmSolverSdeOracle()
        o.set_timesteps(n)
        assert s.host_timesteps == g[f"ts{n}"].tolist() == o.timesteps.tolist()
        x = torch.from_numpy(g[f"traj{n}"][0]).to(device)
        xo = g[f"traj{n}"][0]
        for i, t in enumerate(s.host_timesteps):
            x = s.step(torch.from_numpy(g[f"eps{n}"][i]).to(device), t, x,
                       noise=torch.from_numpy(g[f"z{n}"][i]).to(device))["prev_sample"]
            xo = o.step(g[f"eps{n}"][i], t, xo, g[f"z{n}"][i])
            assert torch.allclose(x.cpu(), torch.from_numpy(xo), rtol=1e-10, atol=1e-10)
            assert torch.allclose(x.cpu(), torch.from_numpy(g[f"traj{n}"][i + 1]), rtol=1e-4, atol=1e-4)


def test_completion_loop_vs_oracle(device, models):
    """T=3 closed loop on a small scene with the scheduler noise injected from one shared tensor
    (device RNG != CPU RNG): device pipeline vs oracle networks + oracle DPM-Solver++."""
    from lidiff_amd.pipeline import DiffCompletion
    enc, unet, refine, sd = models
    pipe = DiffCompletion(denoising_steps=3, cond_weight=6.0, device=device)
    pipe.partial_enc, pipe.model, pipe.model_refine = enc, unet, refine
    scan_np, noisy_np = small_scene(seed=3, n=1500)
    rng = np.random.default_rng(0)
    z = rng.standard_normal((3, 1, scan_np.shape[0], 3))
    scan = torch.from_numpy(scan_np).double()[None].to(device)
    x_feats = torch.from_numpy(noisy_np).double()[None].to(device)
    pipe.new_scheduler()
    out = pipe.completion_loop(scan, pipe.points_to_tensor(x_feats), pipe.points_to_tensor(scan),
                               pipe.points_to_tensor(torch.zeros_like(scan)),
                               noises=[torch.from_numpy(z[i]).to(device) for i in range(3)])
    # oracle loop (pipeline:155-169 restated on the CPU)
    o = DpmSolverSdeOracle()
    ts = o.set_timesteps(3)
    x_init = scan_np.astype(np.float64)[None]
    x_t = noisy_np.astype(np.float64)[None]
    with torch.no_grad():
        for i, t in enumerate(ts):
            xf = net.points_to_field(torch.from_numpy(x_t).float())
            cf = net.points_to_field(torch.from_numpy(scan_np)[None])
            uf = net.points_to_field(torch.zeros(1, scan_np.shape[0], 3))
            eps = net.classfree_forward(sd, xf, cf, uf, torch.tensor([int(t)]), w=6.0).numpy()
            x_t = x_init + o.step(eps, t, xf.F.numpy().reshape(1, -1, 3) - x_init, z[i])
    want = x_t.astype(np.float32).reshape(-1, 3)
    # (A point whose coordinate sits within the fp32 noise of a voxel boundary may be voxelised differently on the two
    # sides -- GPU / CPU round(x / 0.05) disagree for ~5 ppm of inputs -- which legitimately changes that point's trajectory;
    # on this scene and seed no point is affected.)
    err = np.abs(out - want).max(axis=1)
    record_parity("closed_loop_T3_1500pts", share_above_5mm=float(np.mean(err > 5e-3)), share_above_0p1mm=float(np.mean(err > 1e-4)),
                  median_err_m=float(np.median(err)), max_err_m=float(err.max()))
    # measured on the MI355X: EVERY point within 3.8e-6 m after the three closed-loop steps (median 7e-7); round 2 accepted
    # 2 % of the points off by more than 5 mm here
    assert err.max() < 1e-4, (float(np.mean(err > 1e-4)), float(np.median(err)), float(err.max()))
    # encoding the step-invariant conditions once per scan changes nothing, bit for bit
    pipe.cache_condition = True
    pipe.new_scheduler()
    cached = pipe.completion_loop(scan, pipe.points_to_tensor(x_feats), pipe.points_to_tensor(scan),
                                  pipe.points_to_tensor(torch.zeros_like(scan)),
                                  noises=[torch.from_numpy(z[i]).to(device) for i in range(3)])
    dd = np.abs(cached - out).max(axis=1)
    # bit for bit: every kernel on the inference path is deterministic (the voxel mean sums in fixed point: order-free)
    assert np.array_equal(cached, out), (np.count_nonzero(dd), dd.size, dd.max())


def test_fused_step_boundary_equals_the_torch_sequence(device, models, fps_scan):
    """SURVEY.md 8(f) row 1 / pipeline:148-167: the boundary between two steps as ONE launch (lidiff_cfg_dpm_step: guidance,
    DPM-Solver++ update, the next field's points and voxel coordinates) and points_to_tensor as one launch
    (lidiff_points_to_field) give the SAME BITS as the ~25 torch launches they stand for -- every product and sum rounded where
    torch rounds it, x / 0.05 as the reciprocal multiplication torch's GPU kernels perform: (a) points_to_tensor on fp64 / fp32
    points with two batches, (b) a T = 6 closed loop (first-order first step, second-order afterwards) with injected noise and
    with the scheduler's own draws from the same generator state."""
    from lidiff_amd.pipeline import DiffCompletion
    enc, unet, refine, sd = models
    pipe = DiffCompletion(denoising_steps=6, cond_weight=6.0, device=device)
    pipe.partial_enc, pipe.model, pipe.model_refine = enc, unet, refine
    rng = np.random.default_rng(5)
    for dt in (torch.float64, torch.float32):
        pts = torch.from_numpy(rng.standard_normal((2, 5000, 3)) * 30.0).to(dt).to(device)
        got = {}
        for fused in (True, False):
            pipe.fused_step = fused
            f = pipe.points_to_tensor(pts)
            ci = f.C if f.C.dtype == torch.int32 else torch.floor(f.C).to(torch.int32)
            got[fused] = (f.F.contiguous().cpu(), ci.cpu())
        assert torch.equal(got[True][0], got[False][0]) and torch.equal(got[True][1], got[False][1]), dt
        assert got[True][1][:, 0].unique().tolist() == [0, 20]          # the batch column is divided by the resolution too
    scan_np = np.tile(fps_scan[:600].astype(np.float32), (10, 1))
    noisy_np = noisy_scan_points(fps_scan[:600], 1.0, 3)
    scan = torch.from_numpy(scan_np).double()[None].to(device)
    x_feats = torch.from_numpy(noisy_np).double()[None].to(device)
    z = [torch.from_numpy(rng.standard_normal((1, scan_np.shape[0], 3))).to(device) for _ in range(6)]
    outs = {}
    for mode in ("injected", "drawn"):
        for fused in (True, False):
            pipe.fused_step = fused
            pipe.new_scheduler()
            torch.manual_seed(77)
            outs[(mode, fused)] = pipe.completion_loop(scan, pipe.points_to_tensor(x_feats), pipe.points_to_tensor(scan),
                                                       pipe.points_to_tensor(torch.zeros_like(scan)),
                                                       noises=z if mode == "injected" else None)
        d = np.abs(outs[(mode, True)] - outs[(mode, False)])
        assert np.array_equal(outs[(mode, True)], outs[(mode, False)]), (mode, np.count_nonzero(d), float(d.max()))
    assert not np.array_equal(outs[("injected", True)], outs[("drawn", True)])
    pipe.fused_step = True


def test_overlapped_coordinate_pipeline_equals_the_serial_one(device, models, fps_scan):
    """DiffCompletion.overlap_maps: the coordinate pipeline of a field on a side stream, under another tensor's convolutions
    (the next step's conditions under the UNet, x_t's maps under the condition encoders).  Scheduling only: four closed-loop
    steps on the 180k-point scan give the same points with and without it, BIT FOR BIT -- twice, to give a stream hazard
    (memory handed back to one stream while the other still reads it) a chance to show.  (Round 2 had to allow last-bit
    differences here: the voxel mean used fp32 atomics, whose order depends on what else is running.)"""
    from lidiff_amd.pipeline import DiffCompletion
    enc, unet, refine, _ = models
    scan = torch.from_numpy(np.tile(fps_scan, (10, 1))).double()[None].to(device)
    g = torch.Generator(device="cpu").manual_seed(5)
    x0 = (scan.cpu() + torch.randn(scan.shape, generator=g, dtype=torch.float64)).to(device)
    zs = [torch.randn(scan.shape, generator=g, dtype=torch.float64).to(device) for _ in range(4)]
    outs = []
    for overlap in (False, True, True, False):
        pipe = DiffCompletion(denoising_steps=4, cond_weight=6.0, device=device)
        pipe.partial_enc, pipe.model, pipe.model_refine = enc, unet, refine
        pipe.overlap_maps = overlap
        pipe.new_scheduler()
        outs.append(pipe.completion_loop(scan, pipe.points_to_tensor(x0), pipe.points_to_tensor(scan),
                                         pipe.points_to_tensor(torch.zeros_like(scan)), noises=zs))
        torch.cuda.synchronize()
    assert np.isfinite(outs[0]).all()
    for o in outs[1:]:
        d = np.abs(o - outs[0]).max(axis=1)
        assert np.array_equal(o, outs[0]), (np.count_nonzero(d), d.size, d.max())


def test_two_forward_mode_and_weight_updates_under_the_side_stream(device, models, fps_scan):
    """(1) pair_cfg = False (two forwards per step, as the reference runs them) with the coordinate pipeline on the
    side stream must join that stream before the encoder reads a condition's maps -- closed loop equal, bit for bit, to the
    serial schedule; (2) after the weights change (load_state_dict between two scans) the next encoder pass must re-pack /
    re-fold on the MAIN stream before the side stream may run the unconditional encoder again -- equal to a fresh serial
    pipeline holding the new weights."""
    import copy
    from lidiff_amd.pipeline import DiffCompletion
    enc, unet, refine, _ = models
    scan = torch.from_numpy(np.tile(fps_scan[:6000], (10, 1))).double()[None].to(device)
    g = torch.Generator(device="cpu").manual_seed(7)
    x0 = (scan.cpu() + 0.5 * torch.randn(scan.shape, generator=g, dtype=torch.float64)).to(device)
    zs = [torch.randn(scan.shape, generator=g, dtype=torch.float64).to(device) for _ in range(3)]

    def run(pipe):
        pipe.new_scheduler()
        out = pipe.completion_loop(scan, pipe.points_to_tensor(x0), pipe.points_to_tensor(scan),
                                   pipe.points_to_tensor(torch.zeros_like(scan)), noises=zs)
        torch.cuda.synchronize()
        return out

    def make(overlap, pair, enc_, unet_):
        pipe = DiffCompletion(denoising_steps=3, cond_weight=6.0, device=device)
        pipe.partial_enc, pipe.model, pipe.model_refine = enc_, unet_, refine
        pipe.overlap_maps, pipe.pair_cfg = overlap, pair
        return pipe
    # (1) two-forward mode
    serial = run(make(False, False, enc, unet))
    for _ in range(2):
        assert np.array_equal(run(make(True, False, enc, unet)), serial)
    assert np.array_equal(run(make(True, True, enc, unet)), run(make(False, True, enc, unet)))
    # (2) weights replaced after the side stream has been in use
    enc2, unet2 = copy.deepcopy(enc), copy.deepcopy(unet)
    pipe = make(True, True, enc2, unet2)
    first = run(pipe)
    with torch.no_grad():
        new_sd = {k: (v * 1.05 if v.is_floating_point() and "running" not in k and "num_batches" not in k else v)
                  for k, v in enc2.state_dict().items()}
    enc2.load_state_dict(new_sd)
    second = run(pipe)                                                    # same pipeline object: _warm_state is stale now
    assert not np.array_equal(first, second)
    enc3 = copy.deepcopy(enc)
    enc3.load_state_dict(new_sd)
    assert np.array_equal(second, run(make(False, True, enc3, copy.deepcopy(unet))))


def test_training_steps_run_and_learn(device):
    """models.py:180-217 / models_refine.py:53-76 on the HIP path: forward in training mode through the ME shim,
    backward (dX on the conv kernel over the swapped maps, dW from rulebook + gathers + GEMM), Adam step.  The
    packed-weight cache must follow the in-place optimizer update, and the loss on a fixed batch must go down."""
    from lidiff_amd.diffusion import DiffusionPoints, RefineDiffusion
    torch.manual_seed(0)
    scan, _ = small_scene(seed=9, n=600)
    full = torch.from_numpy(np.stack([scan, scan[::-1].copy()]))                   # [2, 600, 3]
    part = full[:, :60].contiguous()
    mod = DiffusionPoints(device=device)
    opt, _ = mod.configure_optimizers()
    g = torch.Generator(device=device).manual_seed(1)
    losses = []
    for _ in range(3):
        g.manual_seed(1)                                                            # same noise / timesteps every step
        loss = mod.training_step({"pcd_full": full, "pcd_part": part}, generator=g)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        kg = mod.model.stage3[1].net[0].kernel.grad
        assert kg is not None and torch.isfinite(kg).all() and kg.abs().sum() > 0
        assert torch.isfinite(mod.model.stem[0].kernel.grad).all()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    ref = RefineDiffusion(device=device)
    batch = {"pcd_noise": full[:, :200].contiguous(), "pcd_full": full[:, :400].contiguous()}
    l0 = ref.training_step(batch)
    l0.backward()
    assert torch.isfinite(l0) and torch.isfinite(ref.model_refine.stage2[1].net[0].kernel.grad).all()


def test_cfg_pair_equals_two_forwards(device, models, fps_scan):
    """pipeline:148-153: the conditional / unconditional pair run as ONE stacked pass (every conv one launch with two
    replicas) must equal two separate forwards -- on the realistic sparsity of the 180k-point workload."""
    enc, unet, _, _ = models
    pts = noisy_scan_points(fps_scan, 0.5, 3)
    t = torch.tensor([500], device=device)
    with torch.no_grad():
        xf = to_field(pts, device)
        xs = xf.sparse()
        pc = enc(to_field(np.tile(fps_scan.astype(np.float32), (10, 1)), device))
        pu = enc(to_field(np.zeros((pts.shape[0], 3), np.float32), device))
        one_c, one_u = unet(xf, xs, pc, t), unet(xf, xs, pu, t)
        two_c, two_u = unet(xf, xs, (pc, pu), t)
    assert two_c.shape == one_c.shape == (pts.shape[0], 3)
    assert torch.allclose(two_c, one_c, rtol=1e-4, atol=1e-4) and torch.allclose(two_u, one_u, rtol=1e-4, atol=1e-4)
    gap = (one_c - one_u).abs().max().item()                      # the two conditions do differ ...
    assert gap > 0
    # ... and the stacked pass keeps them apart: each half matches ITS condition far better than the other one
    assert (two_c - one_c).abs().max().item() <= max(1e-6, 0.05 * gap), ((two_c - one_c).abs().max().item(), gap)
    assert (two_u - one_u).abs().max().item() <= max(1e-6, 0.05 * gap), ((two_u - one_u).abs().max().item(), gap)


def test_complete_scan_end_to_end(device, fps_scan):
    """DiffCompletion.complete_scan (pipeline:117-132) on the bundled scan: range filter, GPU farthest-point
    sampling, T = 2 CFG denoising steps (stacked pair), post-filter, refinement network -> [6 * P, 3] points."""
    from lidiff_amd.pipeline import DiffCompletion
    torch.manual_seed(0)
    pipe = DiffCompletion(denoising_steps=2, cond_weight=6.0, device=device,
                          hparams={"data": {"num_points": 20000}})
    gen = torch.Generator(device=device).manual_seed(3)
    refined, diffused = pipe.complete_scan(fps_scan.astype(np.float64), generator=gen)
    assert diffused.ndim == 2 and diffused.shape[1] == 3 and 0 < diffused.shape[0] <= 20000
    assert refined.shape == (6 * diffused.shape[0], 3)
    assert np.isfinite(refined).all() and np.isfinite(diffused).all()


@pytest.mark.parametrize("drop,precision", [(False, "32"), (True, "32")])
def test_training_step_loss_and_gradients_vs_oracle(device, drop, precision):
    """DiffusionPoints.training_step (models.py:180-217) on the HIP path -- train-mode BatchNorm, conv forward / dX / dW
    kernels, voxel-mean and slice backward, per-batch broadcast of the time embedding -- against the oracle's functional
    restatement differentiated by torch autograd on the CPU, with the step's random draws (noise, t, condition drop) shared.
    Bars: the loss within 1e-5 relative (measured 3e-7); the head's gradients (model.last.*) within 1e-4; EVERY parameter's
    gradient (322 tensors) with cosine >= 0.9999 and its norm within 2e-3 of the oracle's (measured worst: cosine 0.999999,
    norm 4.9e-4).  The scene is sized so that the coarse levels hold a few hundred voxels: on a 600-point scene train-mode
    BatchNorm over the ~20 voxels of stride 16 amplifies last-bit differences (atomic voxel mean / slice / dW sums, as in
    ME's own GPU path) until two runs of the SAME device step differ by 2.6e-2 in a BatchNorm bias gradient
    (measured in round 2) -- a property of the step, not of either implementation.
    (precision "bf16" is not pinned at this level: rounding to bf16 is discontinuous, a last-bit difference that crosses a
    rounding boundary becomes a 2^-9 one, and within ~3 layers the device and ANY emulation of it differ by the bf16 noise
    floor itself -- measured: device vs me.bf16_operands() emulation, cosine 0.92 on the first kernel's gradient, the same as
    bf16 vs fp32.  The bf16 kernels are pinned per layer at the fp32 bar instead: test_gpu_kernels.py, *bf16*.)"""
    from lidiff_amd.diffusion import DiffusionPoints
    torch.manual_seed(3)
    mod = DiffusionPoints(device=device, precision=precision)
    loss_tol, cos_bar, norm_bar, head_bar = 1e-5, 0.9999, 2e-3, 1e-4
    for m in mod.modules():                                    # away from the 1 / 0 initialisation of the BatchNorm affine
        if isinstance(m, torch.nn.BatchNorm1d):
            m.weight.data.uniform_(0.8, 1.2)
            m.bias.data.normal_(0, 0.1)
    mod.train()
    sd = {k: v.detach().cpu().clone() for k, v in mod.state_dict().items()}
    for k, v in sd.items():
        v.requires_grad_(v.is_floating_point() and "running" not in k)
    scan, _ = small_scene(seed=9, n=2000)
    full = torch.from_numpy(np.stack([scan, scan[::-1].copy() + np.float32(0.37)]))   # [2, 2000, 3]
    part = full[:, :200].contiguous()
    g = torch.Generator().manual_seed(1)
    noise = torch.randn(full.shape, generator=g)
    t = torch.tensor([700, 30])

    # both sides voxelise coordinates rounded on the CPU (GPU and CPU round(x / 0.05) differ for ~5 ppm of inputs, App. E)
    def cpu_rounded(points, mean=None, std=None):
        import lidiff_amd.MinkowskiEngine as ME
        cpu = net.points_to_field(points.detach().cpu().float(), divide_batch_col=False)
        return ME.TensorField(features=points.reshape(-1, 3).float().to(device), coordinates=cpu.coords_f.to(device),
                              quantization_mode=ME.SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE, device=device)
    mod.points_to_tensor = cpu_rounded
    loss = mod.training_step({"pcd_full": full, "pcd_part": part}, noise=noise, t=t, drop=drop)
    mod.zero_grad(set_to_none=True)
    loss.backward()
    loss_o, _ = net.training_loss(sd, full, part, noise, t, drop_condition=drop)
    names = [k for k, v in sd.items() if v.requires_grad]
    grads_o = torch.autograd.grad(loss_o, [sd[k] for k in names], allow_unused=True)
    l_d, l_o = float(loss.detach()), float(loss_o.detach())
    assert abs(l_d - l_o) <= loss_tol * abs(l_o), (l_d, l_o)
    params = dict(mod.named_parameters())
    worst_cos, worst_norm, compared = ("", 1.0), ("", 0.0), 0
    for k, go in zip(names, grads_o):
        gd = params[k].grad
        if go is None:
            assert gd is None or float(gd.abs().max()) == 0.0, k
            continue
        assert gd is not None, k
        gd, n_o = gd.detach().cpu(), float(go.norm())
        if drop and k.startswith("partial_enc."):
            # the zeroed condition is ONE voxel per batch with identical features: train-mode BatchNorm over B = 2 identical
            # rows has zero variance, its backward multiplies by 1 / sqrt(eps) = 316 per layer while the two rows'
            # gradients cancel analytically -- 25 layers deep that is inf - inf on any hardware (the reference's too):
            # nothing to compare.  The denoiser's own gradients (model.*) are finite and compared below.
            continue
        assert np.isfinite(n_o) and bool(torch.isfinite(gd).all()), k
        if n_o <= 1e-7:                                         # parameters the loss does not depend on
            assert float(gd.norm()) <= 1e-6, (k, float(gd.norm()))
            continue
        compared += 1
        cos = float((gd * go).sum() / (gd.norm() * go.norm()))
        nrel = abs(float(gd.norm()) - n_o) / n_o
        worst_cos = min(worst_cos, (k, cos), key=lambda q: q[1])
        worst_norm = max(worst_norm, (k, nrel), key=lambda q: q[1])
        assert cos >= cos_bar and nrel <= norm_bar, (k, cos, nrel, n_o)
        if k.startswith("model.last."):
            assert float((gd - go).norm()) <= head_bar * n_o, (k, float((gd - go).norm()) / n_o)
    assert compared >= (150 if drop else 300), compared
    print(f"training step drop={drop} precision={precision}: loss {l_d:.7f} vs oracle {l_o:.7f}; {compared} gradients; worst cosine "
          f"{worst_cos[1]:.6f} ({worst_cos[0]}), worst norm deviation {worst_norm[1]:.2e} ({worst_norm[0]})")


def test_refine_training_step_loss_and_gradients_vs_oracle(device):
    """RefineDiffusion.training_step (models_refine.py:53-76): voxelisation with the divided batch column, MinkUNet in
    training mode, Chamfer loss on the HIP nearest-neighbour kernel -- loss and every gradient against the oracle
    (KD-tree matches, torch autograd).  Same bars as the diffusion step."""
    import lidiff_amd.MinkowskiEngine as ME
    from lidiff_amd.diffusion import RefineDiffusion
    torch.manual_seed(5)
    ref = RefineDiffusion(device=device)
    for m in ref.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.weight.data.uniform_(0.8, 1.2)
            m.bias.data.normal_(0, 0.1)
    ref.train()
    sd = {k[len("model_refine."):]: v.detach().cpu().clone() for k, v in ref.state_dict().items()}
    for k, v in sd.items():
        v.requires_grad_(v.is_floating_point() and "running" not in k)
    scan, noisy = small_scene(seed=13, n=1500)
    pcd_noise = torch.from_numpy(np.stack([noisy, noisy[::-1].copy() + np.float32(0.21)]))
    pcd_full = torch.from_numpy(np.stack([scan, scan[::-1].copy() + np.float32(0.21)]))
    # CPU-rounded coordinates on both sides (App. E): patch the module's field construction for the test
    orig = ME.TensorField

    class field_cpu_rounded(orig):
        def __init__(self, features, coordinates, **kw):
            cpu = net.points_to_field(pcd_noise, divide_batch_col=True)
            super().__init__(features=features, coordinates=cpu.coords_f.to(device), **kw)
    ME.TensorField = field_cpu_rounded
    try:
        loss = ref.training_step({"pcd_noise": pcd_noise, "pcd_full": pcd_full})
    finally:
        ME.TensorField = orig
    ref.zero_grad(set_to_none=True)
    loss.backward()
    loss_o = net.refine_training_loss(sd, pcd_noise, pcd_full)
    names = [k for k, v in sd.items() if v.requires_grad]
    grads_o = torch.autograd.grad(loss_o, [sd[k] for k in names], allow_unused=True)
    l_d, l_o = float(loss.detach()), float(loss_o.detach())
    assert abs(l_d - l_o) <= 1e-5 * abs(l_o), (l_d, l_o)
    params = dict(ref.model_refine.named_parameters())
    compared, worst = 0, ("", 1.0)
    for k, go in zip(names, grads_o):
        if go is None or float(go.norm()) <= 1e-9:
            continue
        gd = params[k].grad.detach().cpu()
        cos = float((gd * go).sum() / (gd.norm() * go.norm()))
        nrel = abs(float(gd.norm()) - float(go.norm())) / float(go.norm())
        worst = min(worst, (k, cos), key=lambda q: q[1])
        assert cos >= 0.9999 and nrel <= 2e-3, (k, cos, nrel)
        compared += 1
    assert compared >= 150, compared                              # MinkUNet: 151 parameter tensors
    print(f"refine training step: loss {l_d:.7f} vs oracle {l_o:.7f}; {compared} gradients, worst cosine {worst[1]:.6f} ({worst[0]})")


def test_compat_aliases_serve_the_reference_imports_on_the_device(device):
    """lidiff_amd.compat.install(): the three third-party imports of the reference's files (`import MinkowskiEngine as ME`,
    `from pykeops.torch import LazyTensor`, `from diffusers import DPMSolverMultistepScheduler`) resolve to this library,
    and the one KeOps expression LiDiff writes (minkunet.py:403-418: squared distance of [M,1,4] against [1,N,4] rows,
    argKmin(1, dim=1)) runs the HIP arg-min with the oracle's indices (lowest index on ties).  The reference's files
    themselves run over these aliases in tests/test_reference_exec.py -- wherever /root/reference and a GPU are both
    present, which is neither the build container nor the GPU box; this test covers the alias layer on the device."""
    import sys
    import lidiff_amd.compat as compat
    names = compat.install(force=True)
    try:
        import MinkowskiEngine as ME
        from diffusers import DPMSolverMultistepScheduler
        from pykeops.torch import LazyTensor
        import lidiff_amd.MinkowskiEngine as OURS
        from lidiff_amd.schedulers import DPMSolverMultistepScheduler as OURS_SCHED
        assert ME is OURS and DPMSolverMultistepScheduler is OURS_SCHED
        rng = np.random.default_rng(2)
        full_c = np.concatenate([np.repeat(np.arange(2), 900)[:, None], rng.integers(-40, 40, (1800, 3))], axis=1)
        part_c = np.concatenate([np.repeat(np.arange(2), 60)[:, None], rng.integers(-40, 40, (120, 3))], axis=1)
        f = torch.from_numpy(full_c).float().to(device)
        q = torch.from_numpy(part_c).float().to(device)
        scale = float(f.max()) * 2.0
        f[:, 0] *= scale
        q[:, 0] *= scale
        idx = ((LazyTensor(f[:, None, :]) - LazyTensor(q[None, :, :])) ** 2).sum(-1).argKmin(1, dim=1)[:, 0]
        want = me.argmin_match(full_c, part_c)
        assert np.array_equal(idx.cpu().numpy().astype(np.int64), np.asarray(want, np.int64))
        with pytest.raises(NotImplementedError):
            (LazyTensor(f[:, None, :]) - LazyTensor(q[None, :, :])).sum(-1)
    finally:
        compat.uninstall(names)
        for n in names:
            assert n not in sys.modules


def test_bf16_training_step_tracks_the_fp32_step(device):
    """BASELINE configs[4] (train.py, bf16): DiffusionPoints(precision="bf16") runs the convolutions of the step -- forward
    and input gradient -- through lidiff_spconv_fwd_bf16 (all 32-channel-multiple layers on dense maps).  The kernel is
    pinned per layer -- forward, dX, dW -- to the oracle on bf16-rounded operands at the fp32 bar (test_gpu_kernels, *bf16*);
    a whole-step pin is not possible (see test_training_step_loss_and_gradients_vs_oracle).  This is the end-to-end sanity of the
    mode on a full-size (2 x 18 000-point) batch against the fp32 step on the same weights and draws: the loss within 1e-2
    relative and every gradient finite.  Gradient cosines are printed, with a loose floor only: at random initialisation
    the step is ill-conditioned with respect to 2^-9 operand rounding -- the CPU oracle's OWN bf16-emulated step has median
    gradient cosine 0.86 / worst 0.71 against its fp32 step on a 20 000-point scene (DESIGN.md, bf16 training)."""
    from lidiff_amd import ops
    from lidiff_amd.diffusion import DiffusionPoints
    torch.manual_seed(3)
    mod = DiffusionPoints(device=device)
    for m in mod.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.weight.data.uniform_(0.8, 1.2)
            m.bias.data.normal_(0, 0.1)
    mod.train()
    scan = np.load(os.path.join(GOLDEN, "scan_000123_fps18000.npy")).astype(np.float32)      # a real 18 000-point scan:
    rng = np.random.default_rng(5)                     # BatchNorm statistics over thousands of voxels at every level
    full = torch.from_numpy(np.stack([scan + np.float32(0.02) * rng.standard_normal(scan.shape).astype(np.float32)
                                      for _ in range(2)]))
    part = full[:, ::10].contiguous()
    g = torch.Generator().manual_seed(1)
    noise = torch.randn(full.shape, generator=g)
    t = torch.tensor([700, 30])
    out = {}
    for precision in ("32", "bf16"):
        mod.precision = precision
        prof = ops.ConvProfiler(variants=())
        ops.PROFILER = prof
        try:
            loss = mod.training_step({"pcd_full": full, "pcd_part": part}, noise=noise, t=t, drop=False)
            mod.zero_grad(set_to_none=True)
            loss.backward()
        finally:
            ops.PROFILER = None
        n_bf16 = sum(1 for v, *_ in prof.launches if v == "bf16")
        out[precision] = (float(loss.detach()), {k: p.grad.detach().clone() for k, p in mod.named_parameters()
                                                 if p.grad is not None}, n_bf16, len(prof.launches))
    assert out["32"][2] == 0
    assert out["bf16"][2] >= 0.3 * out["bf16"][3] > 0, out["bf16"][2:]          # the dense-map layers
    l32, lbf = out["32"][0], out["bf16"][0]
    assert abs(lbf - l32) <= 1e-2 * abs(l32), (lbf, l32)
    cosines = []
    for k, g32 in out["32"][1].items():
        gbf = out["bf16"][1][k]
        if float(g32.norm()) <= 1e-7:
            continue
        assert bool(torch.isfinite(gbf).all()), k
        cosines.append((float((gbf * g32).sum() / (gbf.norm() * g32.norm())), k))
    cosines.sort()
    med = cosines[len(cosines) // 2][0]
    print(f"bf16 step: loss {lbf:.6f} vs fp32 {l32:.6f} ({abs(lbf - l32) / abs(l32):.2e}); {len(cosines)} gradients, "
          f"median cosine {med:.5f}, worst {cosines[0][0]:.5f} ({cosines[0][1]}); bf16 launches {out['bf16'][2]} of {out['bf16'][3]}")
    assert len(cosines) >= 300 and med >= 0.6, cosines[:5]


_A18_TENSORS = ["model.stem.0.kernel", "model.stage3.1.net.0.kernel", "model.up1.1.0.net.0.kernel",
                "model.latemp_stage2.0.weight", "model.last.2.weight", "model.stage4.2.net.1.bn.weight",
                "partial_enc.stage2.1.net.0.kernel"]


def _a18_batches():
    """Four distinct B = 2 batches with their own draws (noise, t): shared by the two-rank run and its one-process check."""
    scan, _ = small_scene(seed=9, n=2000)
    g = torch.Generator().manual_seed(11)
    out = []
    for i in range(4):
        full = torch.from_numpy(np.stack([scan + np.float32(0.013 * i), scan[::-1].copy() + np.float32(0.37 + 0.02 * i)]))
        out.append({"pcd_full": full, "pcd_part": full[:, :200].contiguous(), "noise": torch.randn(full.shape, generator=g),
                    "t": torch.tensor([700 - 100 * i, 30 + 50 * i])})
    return out


def _a18_module(seed, dev):
    from lidiff_amd.diffusion import DiffusionPoints
    torch.manual_seed(seed)
    mod = DiffusionPoints(device=dev)
    step = mod.training_step
    mod.training_step = lambda batch, idx=0: step(batch, idx, noise=batch["noise"], t=batch["t"], drop=False)
    return mod


@pytest.mark.parametrize("precision", ["32", "bf16"])
def test_training_step_is_bit_reproducible(device, precision):
    """Two runs of the same DiffusionPoints training step (same weights, batch and draws) give the SAME loss and the same 322
    parameter gradients bit for bit, in fp32 and with bf16 convolution operands: every kernel of the step is deterministic
    (fixed-point voxel mean, output-stationary convolutions, slice-ordered dW, double-precision BatchNorm statistics in a fixed
    order, segment-sum scatter-adds, matches queued ahead on a side stream included)."""
    from lidiff_amd.diffusion import DiffusionPoints
    batch = _a18_batches()[1]
    runs = []
    for _ in range(2):
        torch.manual_seed(123)
        mod = DiffusionPoints(device=device, precision=precision)
        mod.train()
        loss = mod.training_step(batch, 0, noise=batch["noise"], t=batch["t"], drop=False)
        loss.backward()
        runs.append((loss.detach().cpu(), {k: v.grad.detach().cpu() for k, v in mod.named_parameters() if v.grad is not None}))
    assert torch.equal(runs[0][0], runs[1][0])
    assert runs[0][1].keys() == runs[1][1].keys() and len(runs[0][1]) > 300
    for k in runs[0][1]:
        assert torch.equal(runs[0][1][k], runs[1][1][k]), k


def test_bf16_shadow_rows_give_the_same_training_step_bit_for_bit(device):
    """bf16 activations in HBM (ops.BF16_ROWS, VERDICT r4 #4b): the bf16 convolutions -- forward, input gradient, weight gradient --
    gather the bf16 SHADOW of their inputs (one cast per tensor) instead of rounding the fp32 rows inside every kernel.  Rounding
    is the same function of the same fp32 value wherever it happens, and the kernels keep their products and order of sums: the
    loss and all parameter gradients of a DiffusionPoints step are bit-identical with the shadow rows and without."""
    from lidiff_amd import ops
    from lidiff_amd.diffusion import DiffusionPoints
    batch = _a18_batches()[1]
    runs = []
    keep, keep_wide = ops.BF16_ROWS, ops.BF16_WIDE
    ops.BF16_WIDE = False            # (the wide register-tile kernel sums in another order: the next test)
    try:
        for rows in (True, False):
            ops.BF16_ROWS = rows
            torch.manual_seed(123)
            mod = DiffusionPoints(device=device, precision="bf16")
            mod.train()
            loss = mod.training_step(batch, 0, noise=batch["noise"], t=batch["t"], drop=False)
            loss.backward()
            runs.append((loss.detach().cpu(), {k: v.grad.detach().cpu() for k, v in mod.named_parameters() if v.grad is not None}))
    finally:
        ops.BF16_ROWS, ops.BF16_WIDE = keep, keep_wide
    assert torch.equal(runs[0][0], runs[1][0])
    assert runs[0][1].keys() == runs[1][1].keys() and len(runs[0][1]) > 300
    for k in runs[0][1]:
        assert torch.equal(runs[0][1][k], runs[1][1][k]), k


def test_wide_bf16_tiles_give_the_same_training_step_up_to_summation_order(device):
    """ops.BF16_WIDE (the default of the bf16 step): layers of 128-multiple width on 256-row register tiles without pair lists --
    ONE fp32 sum per output over all offsets and channels instead of a sum per offset first.  Same operands, same products; per
    layer the two forms agree to 1e-4 (test_spconv_bf16_from_shadow_rows_is_bit_identical) and both are pinned to the oracle's
    emulation per kernel and per block.  A whole step is not comparable more tightly than bf16 steps are in general: a last-bit
    difference of an output that crosses a bf16 rounding boundary of the next layer's input becomes a 2^-9 one, and at random
    initialisation the step is ill-conditioned with respect to those (test_bf16_training_step_tracks_the_fp32_step: median
    gradient cosine 0.86 against fp32).  Asserted here: the step is bit-reproducible, its loss within 1e-2 relative of the
    pair-list kernels' step and the median parameter-gradient cosine >= 0.9 (printed: the achieved values)."""
    from lidiff_amd import ops
    from lidiff_amd.diffusion import DiffusionPoints
    batch = _a18_batches()[1]
    runs = []
    keep = ops.BF16_WIDE
    try:
        for wide in (True, True, False):
            ops.BF16_WIDE = wide
            torch.manual_seed(123)
            mod = DiffusionPoints(device=device, precision="bf16")
            mod.train()
            loss = mod.training_step(batch, 0, noise=batch["noise"], t=batch["t"], drop=False)
            loss.backward()
            runs.append((loss.detach().cpu().double(), {k: v.grad.detach().cpu().double() for k, v in mod.named_parameters() if v.grad is not None}))
    finally:
        ops.BF16_WIDE = keep
    assert torch.equal(runs[0][0], runs[1][0]) and all(torch.equal(runs[0][1][k], runs[1][1][k]) for k in runs[0][1])
    rel = abs(float(runs[0][0] - runs[2][0])) / abs(float(runs[2][0]))
    cos = []
    for k in runs[2][1]:
        a, b = runs[0][1][k].flatten(), runs[2][1][k].flatten()
        if float(b.norm()) > 1e-7:
            cos.append(float(torch.dot(a, b) / (a.norm() * b.norm())))
    cos.sort()
    print(f"wide vs pair-list bf16 step: loss {float(runs[0][0]):.6f} vs {float(runs[2][0]):.6f} ({rel:.2e}); {len(cos)} gradients, "
          f"median cosine {cos[len(cos) // 2]:.5f}, worst {cos[0]:.5f}")
    record_parity("bf16_wide_step", loss_rel=rel, median_cos=cos[len(cos) // 2], worst_cos=cos[0])
    assert rel <= 1e-2 and cos[len(cos) // 2] >= 0.9, (rel, cos[:3])


def _two_rank_gloo_one_gpu_worker(rank, world, port, q):
    import hashlib
    import torch.distributed as tdist
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from lidiff_amd import dist as ldist
    from lidiff_amd.diffusion import train_loop
    ldist.init_from_env("gloo")                                   # gloo carries device tensors (staged through the host)
    dev = torch.device("cuda", 0)                                 # both ranks share the one GPU of the box
    mod = _a18_module(100 + rank, dev)                            # different initial weights: the broadcast must fix that
    losses = train_loop(mod, _a18_batches(), steps=3, sync_bn=False)
    sd = {k: v.detach().float().cpu() for k, v in mod.named_parameters()}
    h = hashlib.sha1()
    for k in sorted(sd):
        h.update(sd[k].numpy().tobytes())
    q.put((rank, losses, h.hexdigest(), {k: sd[k].numpy() for k in _A18_TENSORS}))
    tdist.barrier()
    tdist.destroy_process_group()


def test_two_rank_train_loop_on_one_gpu_over_gloo(device):
    """SURVEY.md 8 row a18 (train.py:88-101) executed on the 1-GPU box: the real DiffusionPoints train_loop as TWO processes
    sharing cuda:0, gradients exchanged over gloo -- broadcast of rank 0's weights (and invalidation of the packed-weight
    caches), rank-sharded batches (step * 2 + rank), HIP forward / backward, bucketed all-reduce (SUM, / world), Adam.  After 3
    steps (a) both ranks hold BIT-IDENTICAL weights (sha1 over all 322 tensors), and (b) they are the weights ONE process gets
    from rank 0's initial weights when every step applies the mean of the two ranks' gradients: the update w3 - w0 of seven
    representative tensors is BIT-IDENTICAL to that run's (a two-rank sum is commutative, the division by the world size exact,
    and every kernel of the step deterministic).  (Until the scatter-adds of the backward -- slice, conditioning gathers -- became segment sums in a fixed order, Adam's
    sign-like first steps amplified their fp32-atomic noise: 0 .. 5.5 % of the elements flipped, cosines 0.9983 .. 0.99999; with
    every kernel of the step deterministic the two runs agree exactly.)
    BatchNorm statistics are per process here (sync_bn = False); the synchronised form is the next test."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_gloo_one_gpu_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict((r[0], r[1:]) for r in (q.get(timeout=900) for _ in range(2)))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(np.isfinite(res[r][0]).all() and len(res[r][0]) == 3 for r in (0, 1))
    assert res[0][1] == res[1][1], "ranks hold different weights after 3 steps"
    for k in _A18_TENSORS:
        assert np.array_equal(res[0][2][k], res[1][2][k]), k
    # the same three steps in one process: mean of the two ranks' gradients, then Adam
    mod = _a18_module(100, device)
    w0 = {k: v.detach().float().cpu().numpy().copy() for k, v in mod.named_parameters() if k in _A18_TENSORS}
    batches = _a18_batches()
    opt, _ = mod.configure_optimizers()
    mod.train()
    params = [p for p in mod.parameters() if p.requires_grad]
    for step in range(3):
        acc = None
        for r in range(2):
            opt.zero_grad(set_to_none=True)
            mod.training_step(batches[(step * 2 + r) % 4], step).backward()
            grads = [torch.zeros_like(p) if p.grad is None else p.grad.detach().clone() for p in params]
            acc = grads if acc is None else [a + g for a, g in zip(acc, grads)]
        for p, a in zip(params, acc):
            p.grad = a / 2
        opt.step()
    lr = mod.hparams["train"]["lr"]
    for k, v in mod.named_parameters():
        if k not in _A18_TENSORS:
            continue
        d_ref = v.detach().float().cpu().numpy() - w0[k]
        d_two = res[0][2][k] - w0[k]
        cos = float((d_ref * d_two).sum() / (np.linalg.norm(d_ref) * np.linalg.norm(d_two) + 1e-30))
        off = float(np.mean(np.abs(d_ref - d_two) > 0.1 * lr))
        worst = float(np.abs(d_ref - d_two).max())
        print(f"a18 {k}: update cosine {cos:.6f}, elements off by > lr/10: {100 * off:.3f} %, max |difference| {worst:.3e} (lr {lr:g})")
        assert np.linalg.norm(d_ref) > 0 and worst == 0.0, (k, cos, off, worst)        # bit for bit


def _syncbn_forward_backward(mod, batches, n_total):
    """DiffusionPoints.forward on the concatenation of `batches` in TRAINING mode and the gradient of a loss that is a plain sum
    over points (sum |eps - noise|^2 / (3 n_total)): separable over ranks, so the SUM of the ranks' gradients is the gradient of
    the one-process run on the concatenated batch (the training step's own loss has batch-level mean / std terms and is not)."""
    from lidiff_amd.diffusion import prebuild_maps
    full = torch.cat([b["pcd_full"] for b in batches]).to(mod.device)
    part = torch.cat([b["pcd_part"] for b in batches]).to(mod.device)
    noise = torch.cat([b["noise"] for b in batches]).to(mod.device)
    t = torch.cat([b["t"] for b in batches]).to(mod.device)
    x_full = prebuild_maps(mod.points_to_tensor(full + mod.q_sample(torch.zeros_like(full), t, noise)))
    x_part = prebuild_maps(mod.points_to_tensor(part))
    mod.train()
    mod.zero_grad(set_to_none=True)
    eps = mod.forward(x_full, x_full.sparse(), x_part, t)
    ((eps - noise) ** 2).sum().div(3.0 * n_total).backward()
    return eps.detach()


_SYNCBN_STATS = ["model.stem.1.bn", "model.stage2.1.net.1.bn", "model.stage4.2.net.4.bn", "model.up1.0.net.1.bn", "model.up4.1.1.net.4.bn",
                 "partial_enc.stage3.1.downsample.1.bn"]


def _two_rank_syncbn_one_gpu_worker(rank, world, port, q):
    import hashlib
    import torch.distributed as tdist
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import lidiff_amd.MinkowskiEngine as ME
    from lidiff_amd import dist as ldist
    from lidiff_amd import ops
    from lidiff_amd.diffusion import train_loop
    ldist.init_from_env("gloo")
    dev = torch.device("cuda", 0)
    batches = _a18_batches()
    n_total = sum(b["pcd_full"].shape[0] * b["pcd_full"].shape[1] for b in batches[:2])
    # (1) one forward + backward with synchronised statistics: rank r holds batch r of the pair
    mod = _a18_module(100, dev)                                   # same weights on both ranks
    ME.MinkowskiSyncBatchNorm.convert_sync_batchnorm(mod)
    n_sync = sum(type(m) is ops.SyncBatchNorm1d for m in mod.modules())
    eps = _syncbn_forward_backward(mod, [batches[rank]], n_total)
    grads = {}
    for k, p in mod.named_parameters():
        if k in _A18_TENSORS or k.endswith("stem.1.bn.weight") or k.endswith("stem.1.bn.bias"):
            g = p.grad.detach().clone()
            tdist.all_reduce(g)                                   # SUM over ranks = the gradient of the concatenated batch
            grads[k] = g.cpu().numpy()
    named = dict(mod.named_modules())
    stats = {k: (named[k].running_mean.cpu().numpy(), named[k].running_var.cpu().numpy()) for k in _SYNCBN_STATS}
    # (2) the real train_loop with sync_bn=True: conversion, broadcast, rank-sharded batches, HIP SyncBatchNorm, all-reduce, Adam
    mod2 = _a18_module(100 + rank, dev)
    losses = train_loop(mod2, batches, steps=2, sync_bn=True)
    h = hashlib.sha1()
    for k, v in sorted(mod2.state_dict().items()):                # parameters AND BatchNorm running statistics
        h.update(v.detach().cpu().numpy().tobytes())
    n_sync2 = sum(type(m) is ops.SyncBatchNorm1d for m in mod2.modules())
    # (3) the same loop in config 5's transport: bf16 convolution operands, bf16 gradient buckets on the wire
    from lidiff_amd.diffusion import DiffusionPoints
    torch.manual_seed(300 + rank)
    mod3 = DiffusionPoints(device=dev, precision="bf16")
    losses3 = train_loop(mod3, batches, steps=2, sync_bn=True, transport_dtype=torch.bfloat16)
    h3 = hashlib.sha1()
    for k, v in sorted(mod3.state_dict().items()):
        h3.update(v.detach().cpu().numpy().tobytes())
    q.put((rank, n_sync, eps.cpu().numpy(), grads, stats, losses, h.hexdigest(), n_sync2, losses3, h3.hexdigest()))
    tdist.barrier()
    tdist.destroy_process_group()


def test_two_rank_sync_batchnorm_on_one_gpu_over_gloo(device):
    """SURVEY.md 8 row a18, the SyncBatchNorm half (train.py:90 convert_sync_batchnorm): TWO processes sharing cuda:0, statistics
    exchanged over gloo, the normalisation on the norm.hip kernels (ops.SyncBatchNorm1d / _BatchNormTrain(group=...): fp64
    per-channel sums, one all-reduce per layer forward and backward, fused ReLU / shortcut forms kept).
    (1) Each rank runs DiffusionPoints.forward (training mode) on ITS batch of two scans; one process runs it with plain
    BatchNorm on the CONCATENATED batch of four.  Outputs (each rank's rows), BatchNorm running statistics and the summed
    gradients of a point-separable loss agree -- synchronised statistics ARE the statistics of the concatenated batch.
    (2) train_loop(sync_bn=True) for two steps: every tensor of the state dict -- weights and running statistics --
    bit-identical on the two ranks."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_syncbn_one_gpu_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict((r[0], r[1:]) for r in (q.get(timeout=900) for _ in range(2)))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[0][0] == res[1][0] == res[0][6] == res[1][6] and res[0][0] > 70        # all 74 BatchNorms are the sync kind
    # (2) the ranks of the real loop agree bit for bit, running statistics included
    assert np.isfinite(res[0][4]).all() and len(res[0][4]) == 2
    assert res[0][5] == res[1][5], "ranks hold different weights / running statistics after 2 sync-BN steps"
    assert np.isfinite(res[0][7]).all() and len(res[0][7]) == 2 and res[0][8] == res[1][8], \
        "ranks diverged in the bf16 loop (bf16 operands, bf16 gradient transport, synchronised BatchNorm)"
    # (1) against one process on the concatenated batch
    batches = _a18_batches()
    n_total = sum(b["pcd_full"].shape[0] * b["pcd_full"].shape[1] for b in batches[:2])
    mod = _a18_module(100, device)
    eps = _syncbn_forward_backward(mod, batches[:2], n_total).cpu().numpy()
    two = np.concatenate([res[0][1], res[1][1]])
    scale = float(np.abs(eps).max())
    err_out = float(np.abs(two - eps).max()) / scale
    named = dict(mod.named_modules())
    err_stat = 0.0
    for k in _SYNCBN_STATS:
        for r in (0, 1):
            assert np.array_equal(res[0][3][k][0], res[1][3][k][0]) and np.array_equal(res[0][3][k][1], res[1][3][k][1]), k
        for got, want in zip(res[0][3][k], (named[k].running_mean.cpu().numpy(), named[k].running_var.cpu().numpy())):
            err_stat = max(err_stat, float(np.abs(got - want).max() / (np.abs(want).max() + 1e-30)))
    err_grad, cos_min = 0.0, 1.0
    for k, p in mod.named_parameters():
        if k not in res[0][2]:
            continue
        assert np.array_equal(res[0][2][k], res[1][2][k]), k
        want, got = p.grad.detach().cpu().numpy(), res[0][2][k]
        err_grad = max(err_grad, float(np.abs(got - want).max() / (np.abs(want).max() + 1e-30)))
        cos_min = min(cos_min, float((got * want).sum() / (np.linalg.norm(got) * np.linalg.norm(want) + 1e-30)))
    record_parity("syncbn_two_ranks_vs_concatenated_batch", out_rel_err=err_out, running_stat_rel_err=err_stat,
                  grad_rel_err=err_grad, grad_cos_min=cos_min, n_sync_bn=res[0][0])
    assert err_stat <= 1e-6, err_stat
    assert err_out <= 1e-5 and err_grad <= 1e-4 and cos_min >= 0.99999, (err_out, err_grad, cos_min)


def _uneven_syncbn_worker(rank, world, port, q):
    import torch.distributed as tdist
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from lidiff_amd import dist as ldist
    from lidiff_amd import ops
    ldist.init_from_env("gloo")
    dev = torch.device("cuda", 0)
    out = {}
    for case, rows in _UNEVEN_ROWS.items():
        x_all, g_all = _uneven_inputs(case)
        lo = sum(rows[:rank])
        x = x_all[lo:lo + rows[rank]].to(dev).requires_grad_(True)
        bn = ops.SyncBatchNorm1d(x_all.shape[1]).to(dev).train()
        with torch.no_grad():
            bn.weight.copy_(torch.linspace(0.5, 1.5, x_all.shape[1]))
            bn.bias.copy_(torch.linspace(-1, 1, x_all.shape[1]))
        assert ops.bn_fused_applies(bn, x), "the synchronised path must not depend on the local row count"
        y = ops.batch_norm_train(x, bn, relu=(case == "zero"))
        y.backward(g_all[lo:lo + rows[rank]].to(dev))
        gw, gb = bn.weight.grad.clone(), bn.bias.grad.clone()
        tdist.all_reduce(gw)
        tdist.all_reduce(gb)
        out[case] = (y.detach().cpu().numpy(), x.grad.cpu().numpy(), gw.cpu().numpy(), gb.cpu().numpy(),
                     bn.running_mean.cpu().numpy(), bn.running_var.cpu().numpy())
    q.put((rank, out))
    tdist.barrier()
    tdist.destroy_process_group()


_UNEVEN_ROWS = {"one": (1, 700), "zero": (0, 300)}


def _uneven_inputs(case):
    g = torch.Generator().manual_seed(len(case))
    n = sum(_UNEVEN_ROWS[case])
    return torch.randn(n, 32, generator=g) * 2 + 0.5, torch.randn(n, 32, generator=g)


def test_sync_batchnorm_when_a_rank_holds_one_or_no_row(device):
    """the synchronised BatchNorm path is chosen from rank-invariant properties only, and its kernels take 0 / 1 local
    rows: two processes on cuda:0 over gloo, rank 0 holding ONE row (then NO row) of the layer, rank 1 the rest -- outputs, input
    gradients, summed weight / bias gradients and running statistics equal nn.BatchNorm1d on the concatenated rows (the collective
    pattern is the same on both ranks: nothing hangs)."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_uneven_syncbn_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for case in _UNEVEN_ROWS:
        x_all, g_all = _uneven_inputs(case)
        x = x_all.clone().double().requires_grad_(True)
        bn = torch.nn.BatchNorm1d(x_all.shape[1]).double().train()
        with torch.no_grad():
            bn.weight.copy_(torch.linspace(0.5, 1.5, x_all.shape[1]))
            bn.bias.copy_(torch.linspace(-1, 1, x_all.shape[1]))
        y = bn(x)
        if case == "zero":
            y = torch.relu(y)
        y.backward(g_all.double())
        got_y = np.concatenate([res[0][case][0], res[1][case][0]])
        got_dx = np.concatenate([res[0][case][1], res[1][case][1]])
        assert got_y.shape == tuple(y.shape)
        assert np.allclose(got_y, y.detach().numpy(), rtol=1e-5, atol=1e-5), case
        assert np.allclose(got_dx, x.grad.numpy(), rtol=1e-4, atol=1e-6), case
        for r in (0, 1):
            assert np.allclose(res[r][case][2], bn.weight.grad.numpy(), rtol=1e-4, atol=1e-5), case
            assert np.allclose(res[r][case][3], bn.bias.grad.numpy(), rtol=1e-4, atol=1e-5), case
            assert np.allclose(res[r][case][4], bn.running_mean.numpy(), rtol=1e-5, atol=1e-6), case
            assert np.allclose(res[r][case][5], bn.running_var.numpy(), rtol=1e-5, atol=1e-6), case


def _two_rank_train_worker(rank, world, port, q):
    import torch.distributed as tdist
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    from lidiff_amd import dist as ldist
    from lidiff_amd.diffusion import DiffusionPoints, train_loop
    ldist.init_from_env("nccl")
    dev = torch.device("cuda", rank)
    torch.manual_seed(100 + rank)                                 # different initial weights: the broadcast must fix that
    mod = DiffusionPoints(device=dev)
    scan, _ = small_scene(seed=9, n=600)
    full = torch.from_numpy(np.stack([scan, scan[::-1].copy()]))
    batches = [{"pcd_full": full + 0.01 * i, "pcd_part": full[:, :60].contiguous() + 0.01 * i} for i in range(4)]
    losses = train_loop(mod, batches, steps=3)
    from lidiff_amd.ops import SyncBatchNorm1d
    sync_bn = sum(type(m) is SyncBatchNorm1d for m in mod.modules())
    w = mod.model.stage3[1].net[0].kernel.detach().float().cpu()
    rm = mod.model.stem[1].bn.running_mean.detach().cpu()
    q.put((rank, losses, sync_bn, w.sum().item(), w.abs().sum().item(), rm.tolist()))
    tdist.destroy_process_group()


def test_two_rank_train_loop_rccl_syncbn(device):
    """train.py:88-101 on two GPUs: MinkowskiSyncBatchNorm conversion, broadcast of rank 0's weights, rank-sharded batches,
    bucketed gradient all-reduce over RCCL, Adam -- after 3 steps both ranks hold bit-identical weights and (SyncBatchNorm:
    statistics all-reduced) identical running means.  Skipped (and reported as skipped) on a box with one GPU."""
    import torch.multiprocessing as mp
    if torch.cuda.device_count() < 2:
        pytest.skip(f"needs 2 GPUs, this box shows {torch.cuda.device_count()}")
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_train_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict((r[0], r[1:]) for r in (q.get(timeout=600) for _ in range(2)))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[0][1] > 50 and res[0][1] == res[1][1]              # every BatchNorm became a SyncBatchNorm
    assert res[0][2:4] == res[1][2:4] and res[0][4] == res[1][4]  # weights and synced running statistics agree
    assert all(np.isfinite(res[r][0]).all() for r in (0, 1))


def test_bench_n_rank_path_with_two_ranks_sharing_the_gpu(device):
    """bench.py's N > 1 code path executed for real on the 1-GPU box: `python bench.py --gpus 2` (the form the driver's SCALE run
    starts) with LIDIFF_BENCH_SHARE_GPU=1 / LIDIFF_BENCH_BACKEND=gloo -- both ranks on cuda:0, rendezvous over gloo (RCCL cannot
    place two ranks on one device): launcher, rendezvous, the all-reduce that proves the ranks, barrier-bracketed timing with
    the max over ranks, ONE JSON line on stdout (native libraries' banners routed to stderr), whole-job aggregate value."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(LIDIFF_BENCH_SHARE_GPU="1", LIDIFF_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines[:3]
    js = json.loads(lines[0])
    assert js["n_gpus"] == 2 and js["rccl_ranks_seen"] == 2 and js["steps"] == 3 and js["scaling"] == "weak"
    assert js["value"] > 0 and abs(js["value"] - 2 * 3 / (js["ms_per_step"] * 3e-3)) < 1e-6 * js["value"]
    assert "roofline" in js and "cpu_baseline" not in js and "train" not in js          # N = 1 legs stay at N = 1


def test_bench_whole_scan_form_prints_one_line(device):
    """`python bench.py --pipeline --scans 1` (BASELINE configs[2] / [3]'s unit of work: whole scans through complete_scan) run for
    real: ONE JSON line with scans/s and the dtype the path computes in (the --dry-run form of tests/test_host.py stops before the
    line is assembled)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--pipeline", "--scans", "1"], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines[:3]
    js = json.loads(lines[0])
    assert js["unit"] == "scans/s" and js["scans"] == 1 and js["n_gpus"] == 1 and js["value"] > 0
    from lidiff_amd import ops
    assert abs(js["value"] * js["s_per_scan"] - 1.0) < 1e-6 and ("3 bf16 pieces" in js["dtype"]) == (ops.SPLIT_PIECES == 3)
