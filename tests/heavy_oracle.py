"""The slow whole-network ORACLE legs of the BASELINE-configuration parity tests (tests/test_gpu_baseline.py), shared by
the tests and by tests/golden/make_golden.py --heavy, which commits their results as tracked fixtures:

  tests/golden/c1_t{999,300,100,20}.npz   eps [1, 180000, 3] of ONE classifier-free-guided denoising step on the 180 000-point
                                           bench scan at four positions of the T = 50 trajectory (sigma_t = 0.985 / 0.527 /
                                           0.195 / 0.047: the sparsity regimes the kernels switch between) -- oracle/minkunet_cpu.py;
  tests/golden/t50_small.npz               the oracle's own closed loop over all 50 steps on a 2 000-point scene: eps and points
                                           of every step.

Every file records sha1 digests of (a) the exact input bytes (points, a sample of the seeded weights) and (b) the oracle's
source files.  A test uses a fixture only when both digests match what it would feed / run itself; otherwise it recomputes
(memoised under tests/.oracle_cache/, git-ignored).  tests/test_oracle.py::test_heavy_goldens_are_current fails when a
fixture is stale, and recomputes the cheapest one on the CPU to pin fixture == oracle on every run of the CPU suite.
"""
import glob
import hashlib
import os

import numpy as np
import torch

from conftest import (GOLDEN, ROOT, build_seeded_models, diffusion_state_dict, noisy_scan_points, oracle_cached, small_scene,
                      state_dict_arrays)
from oracle import minkunet_cpu as net
from oracle.dpm_solver import DpmSolverSdeOracle

# positions 0, 35, 45, 49 of the T = 50 trajectory; the noise level of each as a LITERAL (sigma_t of the beta in [3.5e-5, 0.007]
# linear schedule rounded to three digits: 0.5271 / 0.1950 / 0.0469; 1.0 at t = 999 as round 1 defined configs[0]) -- computed
# values differ in the last bit between host CPUs, which would change the input bytes and orphan the fixtures
C1_TIMESTEPS = (999, 300, 100, 20)
C1_SIGMAS = {999: 1.0, 300: 0.527, 100: 0.195, 20: 0.047}


def oracle_digest() -> str:
    h = hashlib.sha1()
    for src in sorted(glob.glob(os.path.join(ROOT, "oracle", "*.py"))):
        h.update(open(src, "rb").read())
    return h.hexdigest()


def input_digest(arrays) -> str:
    h = hashlib.sha1()
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str(a.dtype).encode() + str(a.shape).encode() + a.tobytes())
    return h.hexdigest()


def seeded_state_dict():
    enc, unet, _ = build_seeded_models(42)
    return diffusion_state_dict(enc, unet)


def c1_inputs(fps_scan, t: int = 999):
    """(condition scan [180000,3], noisy points [180000,3]) of the step at timestep t: the bundled scan tiled x10 plus
    sigma_t * N(0, I) (seed 0) -- t = 999 is BASELINE configs[0] / the bench's first trajectory position."""
    scan = np.tile(fps_scan.astype(np.float32), (10, 1))
    return scan, noisy_scan_points(fps_scan, C1_SIGMAS[t], 0)


def c1_key(fps_scan, t, sd):
    scan, noisy = c1_inputs(fps_scan, t)
    return [noisy, scan, np.array([t])] + state_dict_arrays(sd)


def c1_compute(fps_scan, t, sd):
    scan, noisy = c1_inputs(fps_scan, t)
    with torch.no_grad():
        return {"eps": net.classfree_forward(sd, net.points_to_field(torch.from_numpy(noisy)[None]),
                                             net.points_to_field(torch.from_numpy(scan)[None]),
                                             net.points_to_field(torch.zeros(1, scan.shape[0], 3)),
                                             torch.tensor([t]), w=6.0).numpy()}


def t50_setup():
    scan_np, noisy_np = small_scene(seed=21, n=2000)
    zs = np.random.default_rng(4).standard_normal((50, 1) + scan_np.shape)
    return scan_np, noisy_np, zs


def t50_key(sd):
    scan_np, noisy_np, zs = t50_setup()
    return [scan_np, noisy_np, zs[0]] + state_dict_arrays(sd)


def t50_compute(sd, steps=50):
    scan_np, noisy_np, zs = t50_setup()
    o = DpmSolverSdeOracle()
    ts = o.set_timesteps(50)
    x_init = scan_np.astype(np.float64)[None]
    cond_o = net.points_to_field(torch.from_numpy(scan_np)[None])
    zero_o = net.points_to_field(torch.zeros(1, scan_np.shape[0], 3))
    xo = noisy_np.astype(np.float64)[None]
    xs, eps_all = [xo], []
    with torch.no_grad():
        for i, t in enumerate(ts[:steps]):
            xf = net.points_to_field(torch.from_numpy(xo).float())
            eps = net.classfree_forward(sd, xf, cond_o, zero_o, torch.tensor([int(t)]), w=6.0)
            xo = x_init + o.step(eps.numpy(), int(t), xf.F.numpy().reshape(1, -1, 3) - x_init, zs[i])
            eps_all.append(eps.numpy())
            xs.append(xo)
    return {"eps": np.stack(eps_all), "x": np.stack(xs)}


def golden_path(name: str) -> str:
    return os.path.join(GOLDEN, name + ".npz")


def save_golden(name: str, key_arrays, data: dict):
    np.savez_compressed(golden_path(name), inputs_sha1=np.array(input_digest(key_arrays)),
                        oracle_sha1=np.array(oracle_digest()), **data)


def golden_status(name: str, key_arrays):
    """(exists, inputs match, oracle sources match)"""
    path = golden_path(name)
    if not os.path.exists(path):
        return False, False, False
    with np.load(path) as z:
        return True, str(z["inputs_sha1"]) == input_digest(key_arrays), str(z["oracle_sha1"]) == oracle_digest()


def golden_or_compute(name: str, key_arrays, compute):
    """The committed fixture when it belongs to exactly these inputs and this oracle; else the oracle, run now (memoised)."""
    ok = golden_status(name, key_arrays)
    if all(ok):
        with np.load(golden_path(name)) as z:
            return {k: z[k] for k in z.files if not k.endswith("_sha1")}, "fixture " + os.path.relpath(golden_path(name), ROOT)
    return oracle_cached(name, key_arrays, compute), "oracle run here (fixture absent or stale: %s)" % (ok,)


def c1_oracle(fps_scan, t: int = 999):
    sd = seeded_state_dict()
    out, src = golden_or_compute(f"c1_t{t}", c1_key(fps_scan, t, sd), lambda: c1_compute(fps_scan, t, sd))
    return torch.from_numpy(out["eps"]), src


def t50_oracle():
    sd = seeded_state_dict()
    scan_np, _, zs = t50_setup()
    o = DpmSolverSdeOracle()
    ts = [int(t) for t in o.set_timesteps(50)]
    traj, src = golden_or_compute("t50_small", t50_key(sd), lambda: t50_compute(sd))
    return scan_np, zs, ts, traj, src


# ----------------------------------------------------------------------------------------------------------------------------
# closed_c2: BASELINE configs[1] END TO END on the oracle -- the 180 000-point bench scan, seeded weights, the closed loop over
# all T = 50 DPM-Solver++ steps with the scheduler's noise draws shared (pipeline:155-169), then postprocess_scan + the MinkUNet
# refinement forward (pipeline:117-132).  tests/test_gpu_baseline.py runs DiffCompletion.completion_loop + refine_forward on
# the same inputs and noise and compares the two completions by the reference's own metric (utils/metrics.py:124-141,
# ChamferDistance) -- the one the BASELINE metric names ("Chamfer vs ref", north_star: within 1e-3).
# ----------------------------------------------------------------------------------------------------------------------------
CLOSED_STEPS = 50
CLOSED_KEEP = (1, 5, 10, 25, 40, 50)      # trajectory positions whose points are kept (float32) to locate any divergence
CLOSED_NOISE_SEED = 11


def closed_noise(i: int, n_points: int) -> np.ndarray:
    """The scheduler's N(0, I) draw of step i, float64 [1, n, 3] -- one generator per step, so that either side can make step
    i's draw without holding all 50 (216 MB)."""
    return np.random.default_rng([CLOSED_NOISE_SEED, i]).standard_normal((1, n_points, 3))


def seeded_refine_state_dict():
    _, _, refine = build_seeded_models(42)
    return {k: v for k, v in refine.state_dict().items()}


def closed_inputs(fps_scan):
    """(condition scan [180000, 3] float32, x_T [180000, 3] float32): what complete_scan builds at pipeline:118-119 --
    the FPS scan tiled x10 plus N(0, I) (seed 0; the same x_T as the c1_t999 fixture)."""
    return c1_inputs(fps_scan, 999)


def closed_key(fps_scan, sd, sd_refine, gpu_rounding=False):
    scan, noisy = closed_inputs(fps_scan)
    mode = [np.array([20.0], np.float32)] if gpu_rounding else []          # the rounding mode belongs to the fixture's identity
    return ([noisy, scan, closed_noise(0, 4), np.array([CLOSED_STEPS, CLOSED_NOISE_SEED])] + mode + state_dict_arrays(sd)
            + state_dict_arrays(sd_refine))


def points_to_field_gpu_rounding(points: torch.Tensor, resolution=0.05):
    """DiffCompletion.points_to_tensor (pipeline:68-84) with the voxel index rounded the way the REFERENCE'S OWN DEVICE path
    rounds it.  LiDiff is a CUDA program: `x_coord / resolution` with a float32 CUDA tensor and a Python scalar runs torch's
    GPU true-divide kernel, which multiplies by the reciprocal computed once in float32 (`a * (1.0f / 0.05f)` = `a * 20.0f`);
    torch's CPU kernel -- what oracle.minkunet_cpu.points_to_field executes -- divides.  The two differ in the last bit for
    ~5 ppm of the coordinates, i.e. a point within one ulp of a cell boundary lands in the neighbouring voxel.  This variant
    restates the device arithmetic (float32 product, round half to even, batch column included) so that the closed loop can be
    compared POINT FOR POINT with the product; the division variant stays the 'reference on the CPU' fixture."""
    from oracle import me_cpu as me
    feats = me.batched_coordinates(list(points), dtype=torch.float32)
    inv = torch.tensor(1.0, dtype=torch.float32) / torch.tensor(resolution, dtype=torch.float32)
    assert inv.item() == 20.0
    coord = torch.round(feats * inv)
    return me.CpuTensorField(feats[:, 1:].contiguous(), coord)


def postprocess_scan(completed, input_scan, max_range=50.0):
    """DiffCompletion.postprocess_scan pipeline:107-115 (numpy; torch's std is the unbiased one)."""
    dist = np.sqrt(np.sum(completed ** 2, -1))
    post = completed[dist < max_range]
    z = torch.from_numpy(np.ascontiguousarray(input_scan[..., 2]))
    max_z, min_z = z.max().item(), (z.mean() - 2 * z.std()).item()
    return post[(post[:, 2] < max_z) & (post[:, 2] > min_z)]


def closed_compute(fps_scan, sd, sd_refine, steps=CLOSED_STEPS, log=None, gpu_rounding=False):
    """gpu_rounding: every field of the loop (x_t, conditions, the refinement input) is voxelised by
    points_to_field_gpu_rounding instead of the oracle's CPU division -> tests/golden/closed_c2_gpu.npz."""
    scan_np, noisy_np = closed_inputs(fps_scan)
    to_field = points_to_field_gpu_rounding if gpu_rounding else net.points_to_field
    n = scan_np.shape[0]
    o = DpmSolverSdeOracle()
    ts = o.set_timesteps(CLOSED_STEPS)
    x_init = scan_np.astype(np.float64)[None]
    cond_o = to_field(torch.from_numpy(scan_np)[None])
    zero_o = to_field(torch.zeros(1, n, 3))
    xo = noisy_np.astype(np.float64)[None]
    out = {}
    with torch.no_grad():
        for i, t in enumerate(ts[:steps]):
            xf = to_field(torch.from_numpy(xo).float())
            eps = net.classfree_forward(sd, xf, cond_o, zero_o, torch.tensor([int(t)]), w=6.0)
            xo = x_init + o.step(eps.numpy(), int(t), xf.F.numpy().reshape(1, -1, 3) - x_init, closed_noise(i, n))
            if i + 1 in CLOSED_KEEP:
                out[f"x{i + 1}"] = xo[0].astype(np.float32)
            if log is not None:
                log(i, int(t), eps.numpy(), xo)
        # complete_scan pipeline:123-130: x_t.F of the LAST field (float32 features), post-filter, refinement forward
        completed = to_field(torch.from_numpy(xo).float()).F.numpy()
        post = postprocess_scan(completed, x_init)
        offset = net.unet_refine_forward(sd_refine, to_field(torch.from_numpy(post)[None])).numpy()
    out["completed"] = completed.astype(np.float32)
    out["post_rows"] = np.array([post.shape[0]])
    out["refine_offset"] = offset.astype(np.float32)            # [P, 18]; the refined cloud is post[:, None] + offset.reshape(-1, 6, 3)
    return out


def closed_name(gpu_rounding=False):
    return "closed_c2_gpu" if gpu_rounding else "closed_c2"


def closed_oracle(fps_scan, gpu_rounding=False):
    sd, sd_refine = seeded_state_dict(), seeded_refine_state_dict()
    return golden_or_compute(closed_name(gpu_rounding), closed_key(fps_scan, sd, sd_refine, gpu_rounding),
                             lambda: closed_compute(fps_scan, sd, sd_refine, gpu_rounding=gpu_rounding))
