"""GPU parity at BASELINE.json's own configuration (pytest -m gpu): the workload bench.py times -- the bundled scan
(FPS 18 000, tiled x10 = 180 000 points) plus sigma * N(0, I), voxel 0.05 m -- checked against the CPU oracle, not
only through size-independent properties:

  (a) coordinate maps at sigma in {1.0, 0.2, 0.05}: unique voxels / inverse / first index, the four strided maps and
      parents, ks3 / ks2 / transposed neighbour tables and the ME-layout rulebook -- BIT-EXACT;
  (b) one sparse convolution per (level x channel pair) the networks really run on those maps, with the sparse-map
      hint the host really passes (the packed-stage / KS = 64 / 16-bit-list kernel variants that the real sparsity
      selects), against the oracle in float64 -- rtol / atol LAYER_TOL = 2e-5 (5x the worst measured);
  (c) BASELINE configs[0] (C1): ONE classifier-free-guided denoising step (timesteps = [999]) on the 180 000-point
      scan against oracle/minkunet_cpu.py, and the same at three later positions of the T = 50 trajectory (t = 300 / 100 /
      20) -- every point within test_gpu_network.NET_RTOL / NET_ATOL (<= 10x the errors measured on the MI355X,
      profiles/r03_parity_errors.txt); truth = the committed fixtures tests/golden/c1_t*.npz;
  (d) a T = 50 trajectory on a small scene with both sides starting every step from the SAME points (teacher
      forcing), so that voxel-boundary flips cannot hide real error: every point of every step within tolerance.

The oracle legs of (b) take a few minutes of host CPU; (c), (d) read tests/golden/ (tests/heavy_oracle.py recomputes when a
fixture does not belong to the inputs / the oracle sources at hand).  Achieved errors go to gpurun_out/parity_errors.jsonl.
"""
import numpy as np
import pytest
import torch

import heavy_oracle as heavy
from conftest import build_seeded_models, noisy_scan_points, record_parity
from oracle import me_cpu as me
from test_gpu_kernels import check_maps, dev_i32
from test_gpu_network import NET_ATOL, NET_RTOL, X_ATOL, to_field

pytestmark = pytest.mark.gpu

RES = 0.05
# one convolution on the bench maps vs the float64 oracle, rtol = atol: measured worst over the 56 cases 4.7e-6 on outputs of
# scale 12, i.e. 0.04 of the per-kernel bar 1e-4 that tests/test_gpu_kernels.py uses (profiles/r03_parity_errors.txt)
LAYER_TOL = 2e-5


def scan_coords(fps_scan, sigma, seed=0):
    """int32 [180000, 4] voxel coordinates of the noisy scan, rounded on the CPU as the oracle does (App. E)."""
    pts = noisy_scan_points(fps_scan, sigma, seed)
    c = torch.round(torch.from_numpy(pts) / RES).to(torch.int32).numpy()
    return np.concatenate([np.zeros((c.shape[0], 1), np.int32), c], axis=1)


# ---------------------------------------------------------------------------------------- (a)
@pytest.mark.parametrize("sigma", [1.0, 0.2, 0.05])
def test_maps_bit_exact_on_the_bench_scan(device, fps_scan, sigma):
    from lidiff_amd import ops
    coords = scan_coords(fps_scan, sigma)
    nbr3 = check_maps(coords, device)           # unique / inverse / first, 4 strided maps, parents, all tables
    uniq, _, _ = me.voxelize(coords)
    assert 160000 < uniq.shape[0] <= 180000
    pin, pout, ptr = ops.rulebook_compact(nbr3)
    o_in, o_out, o_ptr = me.rulebook_from_nbr(nbr3.cpu().numpy())
    assert np.array_equal(ptr.cpu().numpy(), o_ptr)
    assert np.array_equal(pin.cpu().numpy(), o_in) and np.array_equal(pout.cpu().numpy(), o_out)


# ---------------------------------------------------------------------------------------- (b)
# (level, kind, c_in, c_out, split of c_in for a fused ME.cat | 0): the convolutions of MinkUNetDiff / MinkGlobalEnc /
# MinkUNet (minkunet.py:155-368) per level; level l = tensor stride 2^l
LAYERS = [
    (0, "k3", 3, 32, 0), (0, "k3", 32, 32, 0), (0, "down", 32, 32, 0),
    (1, "k3", 32, 32, 0), (1, "k3", 32, 64, 0), (1, "down", 32, 32, 0), (1, "k1", 32, 64, 0),
    (2, "k3", 64, 64, 0), (2, "k3", 64, 128, 0), (2, "down", 64, 64, 0),
    (3, "k3", 128, 128, 0), (3, "k3", 128, 256, 0), (3, "k3", 256, 256, 0), (3, "down", 128, 128, 0),
    (4, "k3", 256, 256, 0),
    (3, "up", 256, 256, 0), (3, "k3", 384, 256, 256), (3, "k1", 384, 256, 256),
    (2, "up", 256, 128, 0), (2, "k3", 192, 128, 128), (2, "k3", 128, 128, 0),
    (1, "up", 128, 96, 0), (1, "k3", 128, 96, 96), (1, "k3", 96, 96, 0), (1, "k1", 128, 96, 96),
    (0, "up", 96, 96, 0), (0, "k3", 128, 96, 96), (0, "k3", 96, 96, 0),
]


class SceneMaps:
    """Coordinate maps of one noisy scan on the device (product manager: its own sparse-map hints) and on the CPU."""

    def __init__(self, fps_scan, sigma, device):
        import lidiff_amd.MinkowskiEngine as ME
        self.sigma = sigma
        coords = scan_coords(fps_scan, sigma)
        field = ME.TensorField(features=torch.zeros(coords.shape[0], 3, device=device),
                               coordinates=dev_i32(coords, device), device=device)
        field.sparse()
        self.mgr = field.coordinate_manager
        ts = 1
        for _ in range(4):
            ts = self.mgr.stride(ts, 2)
        self.mgr.check()

    def table(self, level, kind, cout=128):
        """(nbr on the device, m_in, m_out, sparse hint) exactly as _ConvBase.maps / sparse_hint produce them."""
        mgr, ts = self.mgr, 1 << level
        m = lambda t: mgr.maps[t].coords.shape[0]
        if kind == "k3":
            return mgr.kernel_map(ts, ts, 3), m(ts), m(ts), mgr.is_sparse_map(ts, ts, 3, c_out=cout)
        if kind == "down":
            return mgr.kernel_map(ts, 2 * ts, 2), m(ts), m(2 * ts), mgr.is_sparse_map(ts, 2 * ts, 2, c_out=cout)
        if kind == "up":
            return mgr.kernel_map(2 * ts, ts, 2, True), m(2 * ts), m(ts), mgr.is_sparse_map(2 * ts, ts, 2, True, c_out=cout)
        return None, m(ts), m(ts), False


@pytest.fixture(scope="module")
def scenes(device, fps_scan):
    cache = {}

    def get(sigma):
        if sigma not in cache:
            cache[sigma] = SceneMaps(fps_scan, sigma, device)
        return cache[sigma]
    return get


def _layer_case(device, scene, level, kind, cin, cout, split, replicas=1):
    from lidiff_amd import ops
    nbr, m_in, m_out, hint = scene.table(level, kind, cout)
    g = torch.Generator().manual_seed(1000 * level + cin + cout)
    k = 1 if nbr is None else nbr.shape[0]
    x = torch.randn(replicas * m_in, cin, generator=g)
    w = torch.randn(k, cin, cout, generator=g) / np.sqrt(cin * max(1, k // 3))
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    res = torch.randn(replicas * m_out, cout, generator=g)
    nbr_np = None if nbr is None else nbr.cpu().numpy()
    xd = x.to(device)
    got = ops.spconv_fwd(xd[:, :split].contiguous() if split else xd, w.to(device), nbr, m_out,
                         in_b=xd[:, split:].contiguous() if split else None, scale=scale.to(device),
                         shift=shift.to(device), residual=res.to(device), relu=True, sparse_map=hint,
                         replicas=replicas).cpu().double()
    # oracle in float64; the heaviest layers (> 5e10 multiply-adds: seconds of host time each) in float32 -- its own
    # rounding (~1e-6 relative) is far inside the tolerance
    pairs = m_out if nbr_np is None else int((nbr_np >= 0).sum())
    odt = torch.float64 if float(pairs) * cin * cout < 5e10 else torch.float32
    # the layers the fused plan runs on the split-operand kernel (ops.split3_layer: kernel_size 3, stride >= 4, widths it takes)
    # ALSO through that kernel, rows sorted by neighbour sets as the plan does -- same oracle, same bar as the native kernel
    got3 = None
    if kind == "k3" and (1 << level) >= ops.SPLIT3_MIN_STRIDE and ops.split3_conv_applies(split or cin, cin - split if split else 0, cout):
        nbr_s, order = ops.mask_sorted_map(nbr)
        got3 = ops.spconv_fwd_split3(xd[:, :split].contiguous() if split else xd, w.to(device), nbr_s, m_out,
                                     in_b=xd[:, split:].contiguous() if split else None, scale=scale.to(device),
                                     shift=shift.to(device), residual=res.to(device), relu=True, replicas=replicas,
                                     row_order=order).cpu().double()
    for r in range(replicas):
        want = me.conv_forward(x[r * m_in:(r + 1) * m_in].to(odt), (w if k > 1 else w[0]).to(odt), nbr_np).double()
        want = torch.relu(want * scale.double() + shift.double() + res[r * m_out:(r + 1) * m_out].double())
        d = (got[r * m_out:(r + 1) * m_out] - want).abs()
        err = d.max().item()
        frac = (d / (LAYER_TOL + LAYER_TOL * want.abs())).max().item()
        record_parity("conv_layer_on_bench_maps", sigma=scene.sigma, level=level, kind=kind, c_in=cin, c_out=cout, hint=int(hint),
                      replica=r, max_abs_err=err, max_abs_out=want.abs().max().item(), worst_tolerance_fraction=frac)
        assert torch.allclose(got[r * m_out:(r + 1) * m_out], want, rtol=LAYER_TOL, atol=LAYER_TOL), \
            f"level {level} {kind} {cin}->{cout} hint={hint} replica {r}: max err {err}"
        if got3 is not None:
            d3 = (got3[r * m_out:(r + 1) * m_out] - want).abs()
            record_parity("conv_layer_on_bench_maps_split3", sigma=scene.sigma, level=level, kind=kind, c_in=cin, c_out=cout,
                          replica=r, max_abs_err=d3.max().item(), max_abs_err_native=err, max_abs_out=want.abs().max().item(),
                          worst_tolerance_fraction=(d3 / (LAYER_TOL + LAYER_TOL * want.abs())).max().item())
            assert torch.allclose(got3[r * m_out:(r + 1) * m_out], want, rtol=LAYER_TOL, atol=LAYER_TOL), \
                f"split3 level {level} {cin}->{cout} replica {r}: max err {d3.max().item()} (native {err})"
    return hint


@pytest.mark.parametrize("level,kind,cin,cout,split", LAYERS)
def test_every_network_conv_on_the_bench_maps_sigma1(device, scenes, level, kind, cin, cout, split):
    _layer_case(device, scenes(1.0), level, kind, cin, cout, split)


@pytest.mark.parametrize("sigma", [0.2, 0.05])
@pytest.mark.parametrize("level,kind,cin,cout,split", [
    (0, "k3", 32, 32, 0), (0, "k3", 128, 96, 96), (1, "k3", 96, 96, 0), (1, "k3", 32, 64, 0), (2, "k3", 128, 128, 0),
    (2, "k3", 192, 128, 128), (3, "k3", 256, 256, 0), (4, "k3", 256, 256, 0), (1, "up", 128, 96, 0), (0, "down", 32, 32, 0)])
def test_network_convs_on_the_bench_maps_late_trajectory(device, scenes, sigma, level, kind, cin, cout, split):
    """The same layers where the sparsity regime differs from sigma = 1 (stride 1 gets dense, stride 8/16 small)."""
    _layer_case(device, scenes(sigma), level, kind, cin, cout, split)


def test_cfg_pair_replicas_on_the_bench_maps(device, scenes):
    """The stacked conditional / unconditional pair (two replicas per launch) as bench.py runs every conv."""
    for level, kind, cin, cout, split in [(3, "k3", 256, 256, 0), (0, "k3", 96, 96, 0), (1, "k3", 128, 96, 96),
                                          (2, "down", 64, 64, 0)]:
        _layer_case(device, scenes(1.0), level, kind, cin, cout, split, replicas=2)


def test_sparse_hint_covers_both_kernel_families(device, scenes):
    """The hint really changes along the trajectory (so both the dense and the packed-stage kernels are exercised
    above with the host's own decision)."""
    hints = {(s, l): scenes(s).table(l, "k3")[3] for s in (1.0, 0.05) for l in (0, 3)}
    assert hints[(1.0, 0)] and not hints[(1.0, 3)] and not hints[(0.05, 0)]


# ---------------------------------------------------------------------------------------- (c)
@pytest.mark.parametrize("t", heavy.C1_TIMESTEPS)
def test_c1_one_denoising_step_on_the_180k_scan_vs_oracle(device, fps_scan, t):
    """One classifier-free-guided forward on the 180 000-point scan, EVERY point against oracle/minkunet_cpu.py
    (fixtures tests/golden/c1_t*.npz, generated by tests/golden/make_golden.py --heavy).  t = 999 is BASELINE configs[0]
    (T = 1, timesteps = [999]); t = 300 / 100 / 20 are positions 35 / 45 / 49 of the T = 50 trajectory (sigma_t = 0.527 /
    0.195 / 0.047), where the sparse-map hints flip, centre + tail switches off and the stride-1 maps get dense."""
    from lidiff_amd.pipeline import DiffCompletion
    enc, unet, refine = build_seeded_models(42)
    pipe = DiffCompletion(denoising_steps=1 if t == 999 else 50, cond_weight=6.0, device=device)
    pipe.partial_enc, pipe.model = enc.to(device), unet.to(device)
    assert t in pipe.dpm_scheduler.host_timesteps
    scan, noisy = heavy.c1_inputs(fps_scan, t)
    with torch.no_grad():
        x_t = to_field(noisy, device)
        got = pipe.classfree_forward(x_t, to_field(scan, device), to_field(np.zeros_like(scan), device),
                                     torch.tensor([t], device=device)).cpu()
        hints = [x_t.coordinate_manager.is_sparse_map(1 << l, 1 << l, 3) for l in range(5)]
    want, src = heavy.c1_oracle(fps_scan, t)
    assert got.shape == want.shape == (1, 180000, 3)
    err = (got - want).abs()
    rel = err / (NET_ATOL + NET_RTOL * want.abs())                 # 1.0 = at the tolerance
    record_parity(f"c1_step_180k_t{t}", max_abs_err=err.max().item(), mean_abs_err=err.mean().item(),
                  max_abs_eps=want.abs().max().item(), worst_tolerance_fraction=rel.max().item(),
                  sparse_hints_per_level=[int(h) for h in hints], truth=src)
    print(f"C1 t={t}: max |eps| error {err.max().item():.2e}, mean {err.mean().item():.2e}, full scale {want.abs().max().item():.3f} ({src})")
    assert torch.allclose(got, want, rtol=NET_RTOL, atol=NET_ATOL), (err.max().item(), err.mean().item())


# ---------------------------------------------------------------------------------------- (d)
def test_t50_trajectory_teacher_forced_every_point(device):
    """completion_loop (pipeline:155-169), all 50 steps of the sde-dpmsolver++ trajectory on a 2 000-point scene
    (fixture tests/golden/t50_small.npz: the oracle's own closed loop).  The device starts step i from the oracle's points
    of step i - 1 (identical voxel coordinates), so the comparison is free of voxel-boundary flips and holds for 100 % of
    the points at every step."""
    from lidiff_amd.pipeline import DiffCompletion
    enc, unet, refine = build_seeded_models(42)
    pipe = DiffCompletion(denoising_steps=50, cond_weight=6.0, device=device)
    pipe.partial_enc, pipe.model = enc.to(device), unet.to(device)
    scan_np, zs, ts, traj, src = heavy.t50_oracle()
    assert pipe.dpm_scheduler.host_timesteps == ts and len(ts) == 50
    scan_d = torch.from_numpy(scan_np.astype(np.float64)[None]).to(device)
    worst_eps = worst_x = 0.0
    with torch.no_grad():
        for i, t in enumerate(ts):
            x, x_next, eps_o = traj["x"][i], traj["x"][i + 1], torch.from_numpy(traj["eps"][i])
            x_t = to_field(x.astype(np.float32)[0], device)                     # the SAME points, CPU-rounded coords
            eps_d = pipe.classfree_forward(x_t, to_field(scan_np, device), to_field(np.zeros_like(scan_np), device),
                                           torch.tensor([t], device=device))
            assert torch.allclose(eps_d.cpu(), eps_o, rtol=NET_RTOL, atol=NET_ATOL), (i, (eps_d.cpu() - eps_o).abs().max())
            x_dev = scan_d + pipe.dpm_scheduler.step(eps_d, t, x_t.F.reshape(1, -1, 3) - scan_d,
                                                     noise=torch.from_numpy(zs[i]).to(device))["prev_sample"]
            dx = np.abs(x_dev.cpu().numpy() - x_next).max()
            worst_eps = max(worst_eps, (eps_d.cpu() - eps_o).abs().max().item())
            worst_x = max(worst_x, dx)
            assert dx < X_ATOL, (i, dx)                                          # every point, metres
    record_parity("t50_teacher_forced_2000pts", worst_abs_eps_err=worst_eps, worst_abs_x_err_m=worst_x, truth=src)
    print(f"T=50 teacher-forced: worst |eps| error {worst_eps:.2e}, worst |x| error {worst_x:.2e} m ({src})")


# ---------------------------------------------------------------------------------------- (e)
def _chamfer(a, b, device):
    """utils/metrics.py:124-141 (ChamferDistance.update): (mean NN distance a -> b + mean NN distance b -> a) / 2, metres."""
    from lidiff_amd.evaluation import ChamferDistance
    cd = ChamferDistance(device=device)
    cd.update(a, b)
    return cd.compute()[0]


@pytest.mark.parametrize("gpu_rounding", [False, True])
def test_closed_loop_c2_chamfer_vs_oracle(device, fps_scan, gpu_rounding):
    """gpu_rounding = True: against tests/golden/closed_c2_gpu.npz -- the same oracle loop with every field
    voxelised the way the reference's DEVICE path rounds (x * 20.0f: heavy_oracle.points_to_field_gpu_rounding), which removes the
    one systematic difference between the two sides; what is left is fp32 summation order (|eps| errors of ~5e-7) acting on points
    within that distance of a voxel boundary.  The bars of this variant are the tight ones (below).
    BASELINE configs[1] END TO END, the metric's own acceptance quantity ("Chamfer vs ref", north_star: within 1e-3): the
    180 000-point bench scan, seeded weights, the CLOSED loop over all T = 50 sde-dpmsolver++ steps (every step voxelises the
    points the previous one produced: pipeline:155-169), postprocess_scan and the MinkUNet refinement forward
    (pipeline:117-132) -- DiffCompletion on the device against the oracle's run of the same loop with the same scheduler noise
    (tests/golden/closed_c2.npz, made by tests/golden/make_golden.py --closed from tests/heavy_oracle.closed_compute).
    Compared with the reference's metric, utils/metrics.py:124-141: Chamfer(product, oracle) <= 1e-3 m for the diffused cloud
    and for the refined cloud.  Unlike the teacher-forced tests nothing is re-synchronised here: a point within fp32 noise of a
    voxel boundary may be voxelised differently by the two sides (torch's GPU kernels evaluate x / 0.05 as
    x * 20.0f, the CPU divides -- ~5 ppm of the coordinates per step), which changes that point's -- and through the
    convolutions its neighbours' -- trajectory; the per-point shares and the positions kept along the trajectory are recorded."""
    from lidiff_amd.pipeline import DiffCompletion
    sd, sd_refine = heavy.seeded_state_dict(), heavy.seeded_refine_state_dict()
    key = heavy.closed_key(fps_scan, sd, sd_refine, gpu_rounding)
    name = heavy.closed_name(gpu_rounding)
    assert heavy.golden_status(name, key) == (True, True, True), \
        f"tests/golden/{name}.npz is absent or stale: python tests/golden/make_golden.py --closed[-gpu-rounding] (~100 min of CPU)"
    with np.load(heavy.golden_path(name)) as z:
        want = {k: z[k] for k in z.files if not k.endswith("_sha1")}
    enc, unet, refine = build_seeded_models(42)
    pipe = DiffCompletion(denoising_steps=heavy.CLOSED_STEPS, cond_weight=6.0, device=device)
    pipe.partial_enc, pipe.model, pipe.model_refine = enc.to(device), unet.to(device), refine.to(device)
    scan_np, noisy_np = heavy.closed_inputs(fps_scan)
    n = scan_np.shape[0]
    scan = torch.from_numpy(scan_np).double()[None].to(device)
    x_t = pipe.points_to_tensor(torch.from_numpy(noisy_np).double()[None].to(device))
    x_cond, x_uncond = pipe.points_to_tensor(scan), pipe.points_to_tensor(torch.zeros_like(scan))
    pipe.new_scheduler()
    ts = pipe.dpm_scheduler.host_timesteps
    assert len(ts) == heavy.CLOSED_STEPS
    along = {}
    for i, t_int in enumerate(ts):
        x_t, x_cond, x_uncond = pipe.denoise_step(scan, x_t, x_cond, x_uncond, t_int, torch.from_numpy(heavy.closed_noise(i, n)).to(device),
                                                  next_t=ts[i + 1] if i + 1 < len(ts) else None)
        if i + 1 in heavy.CLOSED_KEEP:
            got = x_t.F.contiguous().cpu().numpy()
            err = np.abs(got - want[f"x{i + 1}"]).max(axis=1)
            along[i + 1] = (float(np.mean(err <= 1e-4)), float(np.mean(err <= 1e-3)), float(np.median(err)), float(err.max()))
    assert pipe.read_free_check() is None              # (steps 2.. ran without host reads: what they assumed held)
    x_t.coordinate_manager.check()
    completed = x_t.F.contiguous().cpu().numpy()
    assert completed.shape == want["completed"].shape == (n, 3) and np.isfinite(completed).all()
    err = np.abs(completed - want["completed"]).max(axis=1)
    cd_diff = _chamfer(completed, want["completed"], device)
    # post-filter + refinement (pipeline:123-130), both sides from their OWN diffused cloud
    post = pipe.postprocess_scan(completed, scan_np[None].astype(np.float64))
    post_o = heavy.postprocess_scan(want["completed"], scan_np[None].astype(np.float64))
    assert post_o.shape[0] == int(want["post_rows"][0]) == want["refine_offset"].shape[0]
    offset = pipe.refine_forward(pipe.points_to_tensor(torch.as_tensor(post[None, :, :]))).reshape(-1, 6, 3).cpu().numpy()
    refined = (post[:, None, :] + offset).reshape(-1, 3)
    refined_o = (post_o[:, None, :] + want["refine_offset"].reshape(-1, 6, 3)).reshape(-1, 3)
    cd_ref = _chamfer(refined, refined_o, device)
    scale = float(np.abs(want["completed"] - scan_np).max())
    record_parity("closed_loop_c2_180k_T50" + ("_device_rounding" if gpu_rounding else ""), chamfer_diffused_m=cd_diff, chamfer_refined_m=cd_ref,
                  share_within_0p1mm=float(np.mean(err <= 1e-4)), share_within_1mm=float(np.mean(err <= 1e-3)),
                  share_within_5mm=float(np.mean(err <= 5e-3)), points_beyond_0p1mm=int(np.sum(err > 1e-4)),
                  median_err_m=float(np.median(err)), max_err_m=float(err.max()), max_offset_m=scale,
                  post_rows=int(post.shape[0]), post_rows_oracle=int(post_o.shape[0]),
                  **{f"step{k}_share_0p1mm": v[0] for k, v in along.items()}, **{f"step{k}_max_err_m": v[3] for k, v in along.items()})
    print(f"closed loop C2: Chamfer diffused {cd_diff:.3e} m, refined {cd_ref:.3e} m; points within 0.1 mm {np.mean(err <= 1e-4):.6f}, "
          f"1 mm {np.mean(err <= 1e-3):.6f}; median {np.median(err):.2e} m, max {err.max():.2e} m; along the trajectory "
          f"(step: share <= 0.1 mm, share <= 1 mm, median, max) {along}; post-filter rows {post.shape[0]} vs {post_o.shape[0]}")
    assert cd_diff <= 1e-3, cd_diff
    assert cd_ref <= 1e-3, cd_ref
    # Per point.  Measured in round 5 against BOTH fixtures: Chamfer 1.33e-5 / 1.34e-5 m; after step 1 EVERY point
    # within 3.8e-6 m; at the end 99.88 % of the points within 0.1 mm (212 / 207 beyond it), 99.99 % within 1 mm, worst 7.2 mm --
    # the same numbers with the CPU's rounding and with the device's (the two oracles themselves differ in 27 points): what moves
    # the remaining points is not the rounding rule but fp32 summation order (|eps| errors ~5e-7, positions ~1e-6 m per step)
    # carrying a point across a voxel boundary on one side only -- from there the point's, and through the convolutions its
    # neighbours', inputs differ.  No float64 oracle can be matched point for point by ANY fp32 execution of a closed loop
    # (tools/closed_loop_sensitivity.py: the device against itself from x_T + 1e-6 m ends at the same Chamfer distance).  The bars
    # below are 10x tighter than north_star's on the cloud and bound every point; a kernel fault that touches a few hundred
    # points (0.2 % of them) fails the 0.1 mm share.
    assert cd_diff <= 1e-4 and cd_ref <= 1e-4, (cd_diff, cd_ref)
    assert along[1][3] <= 1e-5, along[1]                               # one step: every point
    assert along[5][0] >= 0.9999 and along[10][0] >= 0.9998, along
    assert np.mean(err <= 1e-4) >= 0.998 and np.mean(err <= 1e-3) >= 0.9995 and err.max() <= 2e-2, (np.mean(err <= 1e-4), err.max())
