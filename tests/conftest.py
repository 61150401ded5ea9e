import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The oracle legs are many small torch-CPU / numpy ops: on a 128-thread box the default thread count makes them ~20x
# SLOWER than on 8 cores (measured: the T = 50 oracle trajectory 632 s on the GPU box against 29 s in the build
# container), so the test session caps the CPU thread pool.  bench.py's cpu_baseline is not affected (own process).
torch.set_num_threads(min(torch.get_num_threads(), 16))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    # A session takes 2-3 minutes.  Should it ever sit somewhere for a quarter of an hour (a stalled disk under a fixture load was
    # seen once in the build container), the stacks of all threads go to stderr -- every 15 minutes, without failing anything.
    import faulthandler
    faulthandler.dump_traceback_later(900, repeat=True)


def pytest_sessionfinish(session, exitstatus):
    import faulthandler
    faulthandler.cancel_dump_traceback_later()


def has_gpu() -> bool:
    return torch.cuda.is_available()


@pytest.fixture(scope="session")
def device():
    if not has_gpu():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")


@pytest.fixture(scope="session")
def fps_scan():
    """18 000 farthest-point samples of the reference's bundled scan (tests/golden/make_golden.py)."""
    return np.load(os.path.join(GOLDEN, "scan_000123_fps18000.npy"))


def random_cloud(n, extent, seed, batch=1, dup=0.3):
    """Integer voxel coordinates [n,4] with duplicates, negative values and `batch` batches
    (rows grouped by ascending batch as LiDiff produces them)."""
    rng = np.random.default_rng(seed)
    c = rng.integers(-extent, extent + 1, size=(n, 3))
    ndup = int(n * dup)
    if ndup:
        c[rng.integers(0, n, ndup)] = c[rng.integers(0, n, ndup)]
    b = np.sort(rng.integers(0, batch, size=n))
    return np.concatenate([b[:, None], c], axis=1).astype(np.int32)


def noisy_scan_points(fps_scan, sigma, seed, n_rep=10):
    """The bench/parity workload of BASELINE.md section 2: tile the 18k scan x10 and add
    sigma * N(0, I) offsets (float32, metres)."""
    rng = np.random.default_rng(seed)
    base = np.tile(fps_scan.astype(np.float32), (n_rep, 1))
    return (base + sigma * rng.standard_normal(base.shape).astype(np.float32)).astype(np.float32)


def small_scene(seed=5, n=2000):
    """Tiny stand-in for a scan: n/10 'partial' points tiled x10 plus noise (float32 metres)."""
    rng = np.random.default_rng(seed)
    part = (rng.standard_normal((n // 10, 3)) * np.array([4.0, 4.0, 0.5])).astype(np.float32)
    scan = np.tile(part, (10, 1))
    noisy = (scan + 0.3 * rng.standard_normal(scan.shape)).astype(np.float32)
    return scan, noisy


def build_seeded_models(seed=42):
    """MinkGlobalEnc, MinkUNetDiff, MinkUNet(18) with seeded weights and non-trivial BatchNorm
    statistics (CPU tensors).  Shared by make_golden.py and the parity tests."""
    from lidiff_amd import minkunet as product
    torch.manual_seed(seed)
    enc = product.MinkGlobalEnc(in_channels=3)
    unet = product.MinkUNetDiff(in_channels=3)
    refine = product.MinkUNet(in_channels=3, out_channels=18)
    for mod in (enc, unet, refine):
        for m in mod.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.data.uniform_(0.8, 1.2)
                m.bias.data.normal_(0, 0.1)
    return enc.eval(), unet.eval(), refine.eval()


def diffusion_state_dict(enc, unet):
    sd = {"partial_enc." + k: v for k, v in enc.state_dict().items()}
    sd.update({"model." + k: v for k, v in unet.state_dict().items()})
    return sd


def oracle_cached(name: str, key_arrays, compute):
    """Memo of an expensive ORACLE result (a dict of numpy arrays) under tests/.oracle_cache/ (git-ignored, it does
    travel to the GPU box with the snapshot): keyed on the oracle's source files and the exact input bytes, so a hit
    is the value the oracle would compute.  Absent / stale cache -> computed in place (the driver's fresh boxes).
    Only slow whole-network oracle legs use it; it saves GPU-box minutes, it never replaces the comparison."""
    import glob
    import hashlib
    h = hashlib.sha1(name.encode())
    for src in sorted(glob.glob(os.path.join(ROOT, "oracle", "*.py"))):
        h.update(open(src, "rb").read())
    for a in key_arrays:
        a = np.ascontiguousarray(a)
        h.update(str(a.dtype).encode() + str(a.shape).encode() + a.tobytes())
    path = os.path.join(ROOT, "tests", ".oracle_cache", f"{name}_{h.hexdigest()[:20]}.npz")
    if os.path.exists(path):
        with np.load(path) as z:
            return {k: z[k] for k in z.files}
    out = compute()
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        np.savez(path + ".tmp.npz", **out)
        os.replace(path + ".tmp.npz", path)
    except OSError:
        pass
    return out


def state_dict_arrays(sd):
    """A few weight tensors as cache-key material (seeded models: the seed decides all of them)."""
    keys = sorted(sd.keys())
    return [sd[k].detach().cpu().numpy() for k in (keys[0], keys[len(keys) // 2], keys[-1])]


def record_parity(name: str, **values):
    """Append the ACHIEVED error of a parity test as one JSON line to $LIDIFF_PARITY_LOG (default
    gpurun_out/parity_errors.jsonl, which gpurun merges back): the source of profiles/rNN_parity_errors.txt and of the
    tolerances the tests assert (<= 10x the measured values).  Never fails a test."""
    import json
    path = os.environ.get("LIDIFF_PARITY_LOG", os.path.join(ROOT, "gpurun_out", "parity_errors.jsonl"))
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "a") as f:
            f.write(json.dumps({"test": name, **{k: (float(v) if isinstance(v, (int, float, np.floating)) else v)
                                                 for k, v in values.items()}}) + "\n")
    except OSError:
        pass
