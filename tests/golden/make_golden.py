"""Generates the committed fixtures under tests/golden/ (run from the repo root, in the build
container, where /root/reference exists):

    python tests/golden/make_golden.py

The reference ships no golden vectors and none of its native dependencies can be imported
(SURVEY.md 4, 8c), so these vectors come from the CPU oracle under oracle/ -- PARITY UNPINNED.
They pin the oracle against regressions and give the GPU parity tests fixed inputs/outputs.
  scan_000123_range_filtered.npy : bundled scan -> 3.5 m < |p| < 50 m, float32 [119035,3] (exact)
  scan_000123_fps18000.npy : bundled scan lidiff/Datasets/test/000123.ply -> 3.5 m < |p| < 50 m
                             -> greedy farthest-point sampling of 18 000 points (index 0 first),
                             float32 [18000,3]  (the preprocessing of pipeline:92-99)
  coords_small.npz         : voxelize / stride maps / kernel maps / rulebook of a seeded cloud
  conv_small.npz           : sparse conv forward (ks 3, 2/stride 2, transposed, ks 1) on it
  dpm_trajectory.npz       : DPM-Solver++ (sde, 2nd order) trajectory with injected noise
  unet_small.npz           : MinkGlobalEnc + MinkUNetDiff CFG output and MinkUNet output on a
                             2 000-point cloud, weights from torch.manual_seed(42)
  c1_t*.npz, t50_small.npz : (--heavy) the whole-network oracle results of the BASELINE-configuration tests, see make_heavy()
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
HERE = os.path.dirname(os.path.abspath(__file__))

from conftest import build_seeded_models, diffusion_state_dict, random_cloud, small_scene  # noqa: E402
from oracle import me_cpu as me  # noqa: E402
from oracle import minkunet_cpu as net  # noqa: E402
from oracle.dpm_solver import DpmSolverSdeOracle  # noqa: E402


def fps_numpy(pts, n):
    sel = np.empty(n, np.int64)
    dist = np.full(pts.shape[0], np.inf)
    far = 0
    for i in range(n):
        sel[i] = far
        dist = np.minimum(dist, ((pts - pts[far]) ** 2).sum(-1))
        far = int(np.argmax(dist))
    return sel


def make_scan():
    """The bundled scan through preprocess_scan (pipeline:92-99): range filter (119 035 points, committed as
    scan_000123_range_filtered.npy -- the file's float64 values are exactly float32) and FPS to 18 000."""
    from lidiff_amd.pipeline import read_ply_points
    from oracle.fps_cpu import farthest_point_sample, range_filter
    pts = range_filter(read_ply_points("/root/reference/lidiff/Datasets/test/000123.ply"))
    assert np.array_equal(pts.astype(np.float32).astype(np.float64), pts)
    np.save(os.path.join(HERE, "scan_000123_range_filtered.npy"), pts.astype(np.float32))
    keep = farthest_point_sample(pts, 18000)
    np.save(os.path.join(HERE, "scan_000123_fps18000.npy"), pts[keep].astype(np.float32))
    print("scan:", pts.shape, "->", keep.shape)


def make_coords():
    c = random_cloud(4000, 12, seed=1, batch=2)
    uniq, inv, first = me.voxelize(c)
    out = {"coords": c, "uniq": uniq, "inverse": inv, "first_idx": first}
    cur, ts = uniq, 1
    for lvl in range(1, 5):
        coarse, parent = me.stride_map(cur, ts * 2)
        out[f"coarse{lvl}"], out[f"parent{lvl}"] = coarse, parent
        out[f"nbr_down{lvl}"] = me.kernel_map(cur, coarse, 2, ts)
        cur, ts = coarse, ts * 2
    out["nbr3_l0"] = me.kernel_map(uniq, uniq, 3, 1)
    out["nbr3_l2"] = me.kernel_map(out["coarse2"], out["coarse2"], 3, 4)
    pin, pout, ptr = me.rulebook_from_nbr(out["nbr3_l0"])
    out.update(rb_in=pin, rb_out=pout, rb_ptr=ptr)
    np.savez_compressed(os.path.join(HERE, "coords_small.npz"), **out)
    return out


def make_conv(cm):
    g = torch.Generator().manual_seed(7)
    m0, m1 = cm["uniq"].shape[0], cm["coarse1"].shape[0]
    x0 = torch.randn(m0, 32, generator=g)
    w3 = torch.randn(27, 32, 64, generator=g) * 0.1
    w2 = torch.randn(8, 32, 32, generator=g) * 0.1
    wt = torch.randn(8, 32, 96, generator=g) * 0.1
    w1 = torch.randn(32, 64, generator=g) * 0.1
    y3 = me.conv_forward(x0, w3, cm["nbr3_l0"])
    yd = me.conv_forward(x0, w2, cm["nbr_down1"])
    up = me.transpose_kernel_map(cm["nbr_down1"], m0)
    yu = me.conv_forward(yd, wt, up)
    y1 = me.conv_forward(x0, w1, None)
    np.savez_compressed(os.path.join(HERE, "conv_small.npz"), x0=x0.numpy(), w3=w3.numpy(), w2=w2.numpy(),
                        wt=wt.numpy(), w1=w1.numpy(), y3=y3.numpy(), yd=yd.numpy(), yu=yu.numpy(),
                        y1=y1.numpy(), nbr_up1=up)
    assert yd.shape[0] == m1


def make_dpm():
    rng = np.random.default_rng(3)
    out = {}
    for n in (50, 8, 1):
        o = DpmSolverSdeOracle()
        ts = o.set_timesteps(n)
        x = rng.standard_normal((1, 64, 3))
        eps = rng.standard_normal((len(ts), 1, 64, 3))
        z = rng.standard_normal((len(ts), 1, 64, 3))
        traj = [x]
        for i, t in enumerate(ts):
            traj.append(o.step(eps[i], t, traj[-1], z[i]))
        out[f"ts{n}"], out[f"eps{n}"], out[f"z{n}"], out[f"traj{n}"] = ts, eps, z, np.stack(traj)
    np.savez_compressed(os.path.join(HERE, "dpm_trajectory.npz"), **out)


def reference_unet_outputs():
    """The CFG output and the refinement output of unet_small.npz computed by the REFERENCE'S OWN CODE
    (/root/reference/lidiff/models/minkunet.py, imported unmodified by tests/ref_exec.py) over the oracle's ME / KeOps
    stand-ins: classfree_forward of tools/diff_completion_pipeline.py:140-153 written out with the reference modules."""
    import ref_exec
    enc, unet, refine = build_seeded_models(42)
    ref, mods = ref_exec.reference_minkunet("oracle")
    ME = mods["MinkowskiEngine"]
    r_enc, r_unet = ref.MinkGlobalEnc(in_channels=3, out_channels=96), ref.MinkUNetDiff(in_channels=3, out_channels=96)
    r_ref = ref.MinkUNet(in_channels=3, out_channels=18)
    r_enc.load_state_dict(enc.state_dict()), r_unet.load_state_dict(unet.state_dict()), r_ref.load_state_dict(refine.state_dict())
    r_enc.eval(), r_unet.eval(), r_ref.eval()
    scan, noisy = small_scene()

    def field(points):
        cpu = net.points_to_field(points)
        return ME.TensorField(features=cpu.F, coordinates=cpu.coords_f,
                              quantization_mode=ME.SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE,
                              minkowski_algorithm=ME.MinkowskiAlgorithm.SPEED_OPTIMIZED)

    with torch.no_grad():
        t = torch.tensor([500])
        xf = field(torch.from_numpy(noisy)[None])
        xs = xf.sparse()
        e_c = r_unet(xf, xs, r_enc(field(torch.from_numpy(scan)[None])), t).reshape(1, -1, 3)
        e_u = r_unet(xf, xs, r_enc(field(torch.zeros(1, scan.shape[0], 3))), t).reshape(1, -1, 3)
        eps = e_u + 6.0 * (e_c - e_u)
        off = r_ref(field(torch.from_numpy(noisy)[None]))
    return eps.numpy(), off.numpy()


def make_unet():
    """unet_small.npz: produced by the reference's own networks when /root/reference is mounted (and cross-checked
    against the functional oracle), by the functional oracle otherwise."""
    import ref_exec
    enc, unet, refine = build_seeded_models(42)
    sd = diffusion_state_dict(enc, unet)
    scan, noisy = small_scene()
    with torch.no_grad():
        xf = net.points_to_field(torch.from_numpy(noisy)[None])
        cf = net.points_to_field(torch.from_numpy(scan)[None])
        uf = net.points_to_field(torch.zeros(1, scan.shape[0], 3))
        t = torch.tensor([500])
        eps = net.classfree_forward(sd, xf, cf, uf, t, w=6.0).numpy()
        rf = net.points_to_field(torch.from_numpy(noisy)[None])
        off = net.unet_refine_forward(refine.state_dict(), rf).numpy()
    source = "oracle/minkunet_cpu.py"
    if ref_exec.have_reference():
        r_eps, r_off = reference_unet_outputs()
        assert np.allclose(r_eps, eps, rtol=1e-5, atol=1e-5) and np.allclose(r_off, off, rtol=1e-5, atol=1e-5)
        eps, off, source = r_eps, r_off, "/root/reference/lidiff/models/minkunet.py over oracle/me_shim.py"
    np.savez_compressed(os.path.join(HERE, "unet_small.npz"), scan=scan, noisy=noisy, eps=eps, refine=off,
                        source=np.array(source))
    print("unet eps", eps.shape, float(np.abs(eps).mean()), "refine", off.shape, "from", source)


def make_heavy():
    """python tests/golden/make_golden.py --heavy: the slow whole-network oracle legs of tests/test_gpu_baseline.py
    (tests/heavy_oracle.py) as tracked fixtures -- one CFG step on the 180 000-point bench scan at four trajectory positions
    (c1_t999 / c1_t300 / c1_t100 / c1_t20 .npz, ~45 s of CPU each) and the 50-step closed loop on 2 000 points (t50_small.npz)."""
    import time
    import heavy_oracle as heavy
    fps = np.load(os.path.join(HERE, "scan_000123_fps18000.npy"))
    sd = heavy.seeded_state_dict()
    for t in heavy.C1_TIMESTEPS:
        t0 = time.time()
        out = heavy.c1_compute(fps, t, sd)
        heavy.save_golden(f"c1_t{t}", heavy.c1_key(fps, t, sd), {"eps": out["eps"].astype(np.float32), "t": np.array(t)})
        print(f"c1_t{t}: max |eps| {np.abs(out['eps']).max():.4f}, {time.time() - t0:.0f} s", flush=True)
    t0 = time.time()
    traj = heavy.t50_compute(sd)
    heavy.save_golden("t50_small", heavy.t50_key(sd), {"eps": traj["eps"].astype(np.float32), "x": traj["x"]})
    print(f"t50_small: {time.time() - t0:.0f} s", flush=True)


def make_closed(gpu_rounding=False):
    """python tests/golden/make_golden.py --closed: the oracle's END-TO-END run of BASELINE configs[1] -- closed loop over all
    T = 50 steps on the 180 000-point bench scan with seeded weights and shared scheduler noise, postprocess_scan, MinkUNet
    refinement (tests/heavy_oracle.closed_compute; ~2 min of CPU per step on 8 cores) -> tests/golden/closed_c2.npz, the truth of
    tests/test_gpu_baseline.py::test_closed_loop_c2_chamfer_vs_oracle."""
    import time
    import heavy_oracle as heavy
    fps = np.load(os.path.join(HERE, "scan_000123_fps18000.npy"))
    sd, sdr = heavy.seeded_state_dict(), heavy.seeded_refine_state_dict()
    t0 = time.time()
    scan = np.tile(fps.astype(np.float64), (10, 1))

    def log(i, t, eps, xo):
        off = xo[0] - scan
        print(f"step {i} t={t} max |eps| {np.abs(eps).max():.4f} offsets std {off.std():.4f} max {np.abs(off).max():.2f} "
              f"elapsed {time.time() - t0:.0f} s", flush=True)
    out = heavy.closed_compute(fps, sd, sdr, log=log, gpu_rounding=gpu_rounding)
    heavy.save_golden(heavy.closed_name(gpu_rounding), heavy.closed_key(fps, sd, sdr, gpu_rounding), out)
    print(heavy.closed_name(gpu_rounding) + ":", {k: v.shape for k, v in out.items()}, f"{time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    if "--closed-gpu-rounding" in sys.argv:      # the same loop voxelised as the reference's DEVICE path rounds (x * 20.0f)
        make_closed(gpu_rounding=True)
        sys.exit(0)
    if "--closed" in sys.argv:
        make_closed()
        sys.exit(0)
    if "--heavy" in sys.argv:
        make_heavy()
        sys.exit(0)
    if os.path.exists("/root/reference/lidiff/Datasets/test/000123.ply") and "--no-scan" not in sys.argv:
        make_scan()
    cm = make_coords()
    make_conv(cm)
    make_dpm()
    make_unet()
