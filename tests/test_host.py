"""CPU tests (no GPU): host logic of the product package and the C-ABI library's surface.
No compute kernel is launched here."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT


def test_library_builds_loads_and_exports_every_declared_symbol():
    from lidiff_amd import _lib
    from lidiff_amd.csrc import build
    path = build.build()
    assert os.path.exists(path)
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "lidiff_amd.h")).read()
    declared = set(re.findall(r"\b(lidiff_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), f"{name} not exported"
    # the ctypes mirror must take exactly the parameters the header declares (pointer / integer / float classes too)
    import ctypes as C
    flat = re.sub(r"/\*.*?\*/", " ", header, flags=re.S)
    for name, params in re.findall(r"\b(lidiff_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", flat):
        params = [q.strip() for q in params.split(",") if q.strip() and q.strip() != "void"]
        argtypes = _lib.SIGNATURES[name][1]
        assert len(params) == len(argtypes), (name, params, argtypes)
        for q, a in zip(params, argtypes):
            want = C.c_void_p if "*" in q else C.c_float if q.startswith("float") else C.c_double if q.startswith("double") else None
            assert (a is want) if want is not None else a in (C.c_int32, C.c_int64), (name, q, a)
            if want is None:
                assert a is (C.c_int64 if q.startswith("int64_t") else C.c_int32), (name, q, a)
    assert lib.lidiff_abi_version() == _lib.ABI_VERSION == 28
    assert lib.lidiff_hash_capacity(180000) == 524288 and lib.lidiff_hash_capacity(1) == 1024
    assert lib.lidiff_unique_workspace_bytes(1000) >= 1000 * 4
    # host-side argument validation reaches the error string without touching a device
    rc = lib.lidiff_spconv_fwd(None, 0, None, 0, None, None, 1, 0, 0, 32, None, None, None, None, 0, None, 1, 0, None, None, None, 0,
                               None, None)
    assert rc != 0 and b"lidiff_spconv_fwd" in lib.lidiff_last_error()


def test_operators_refuse_cpu_tensors():
    import lidiff_amd.MinkowskiEngine as ME
    from lidiff_amd import ops
    with pytest.raises(RuntimeError, match="GPU only"):
        ops.coords_floor(torch.zeros(4, 4))
    with pytest.raises(RuntimeError, match="GPU only"):
        ME.TensorField(features=torch.zeros(4, 3), coordinates=torch.zeros(4, 4))


def test_no_product_import_of_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "lidiff_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f
                assert "import oracle" not in src and "from oracle" not in src, f


def expected_keys_unet(with_cond: bool, out_channels: int):
    cs = [32, 32, 64, 128, 256, 256, 128, 96, 96]
    exp = {}

    def bn(p, c):
        for k in ("weight", "bias", "running_mean", "running_var"):
            exp[f"{p}.bn.{k}"] = (c,)
        exp[f"{p}.bn.num_batches_tracked"] = ()

    def res(p, a, b):
        exp[f"{p}.net.0.kernel"] = (27, a, b); bn(f"{p}.net.1", b)
        exp[f"{p}.net.3.kernel"] = (27, b, b); bn(f"{p}.net.4", b)
        if a != b:
            exp[f"{p}.downsample.0.kernel"] = (a, b); bn(f"{p}.downsample.1", b)

    def lin(p, a, b):
        exp[f"{p}.weight"] = (b, a); exp[f"{p}.bias"] = (b,)

    exp["stem.0.kernel"] = (27, 3, 32); bn("stem.1", 32); exp["stem.3.kernel"] = (27, 32, 32); bn("stem.4", 32)
    for n in range(4):
        p = f"stage{n + 1}"
        exp[f"{p}.0.net.0.kernel"] = (8, cs[n], cs[n]); bn(f"{p}.0.net.1", cs[n])
        res(f"{p}.1", cs[n], cs[n + 1]); res(f"{p}.2", cs[n + 1], cs[n + 1])
    for j in range(4):
        p = f"up{j + 1}"
        exp[f"{p}.0.net.0.kernel"] = (8, cs[4 + j], cs[5 + j]); bn(f"{p}.0.net.1", cs[5 + j])
        res(f"{p}.1.0", cs[5 + j] + cs[3 - j], cs[5 + j]); res(f"{p}.1.1", cs[5 + j], cs[5 + j])
    lin("last.0", 96, 20); lin("last.2", 20, out_channels)
    if with_cond:
        hid = {"stage1": (32, 256), "stage2": (32, 256), "stage3": (64, 256), "stage4": (128, 256),
               "up1": (256, 256), "up2": (256, 256), "up3": (128, 128), "up4": (96, 96)}
        for name, (cx, h) in hid.items():
            lin(f"latent_{name}.0", 256, 256); lin(f"latent_{name}.2", 256, 256)
            lin(f"latemp_{name}.0", 512, h); lin(f"latemp_{name}.2", h, cx)
            lin(f"{name}_temp.0", 96, 96); lin(f"{name}_temp.2", 96, 256)
    return exp


def test_state_dict_is_checkpoint_compatible():
    """Key names and shapes of the reference's module tree (SURVEY.md 8b, Appendix C counts)."""
    from lidiff_amd import minkunet
    unet = minkunet.MinkUNetDiff(in_channels=3, out_channels=96)
    sd = {k: tuple(v.shape) for k, v in unet.state_dict().items()}
    assert sd == expected_keys_unet(True, 3)
    assert sd["stage2.1.downsample.0.kernel"] == (32, 64) and sd["up1.0.net.0.kernel"] == (8, 256, 256)
    assert sd["latemp_up1.0.weight"] == (256, 512) and sd["stem.1.bn.running_mean"] == (32,)
    refine = minkunet.MinkUNet(in_channels=3, out_channels=18)
    assert {k: tuple(v.shape) for k, v in refine.state_dict().items()} == expected_keys_unet(False, 18)
    enc = minkunet.MinkGlobalEnc(in_channels=3, out_channels=96)
    enc_keys = {k for k in expected_keys_unet(False, 3) if k.startswith(("stem", "stage"))}
    assert set(enc.state_dict()) == enc_keys
    n = lambda m: sum(p.numel() for p in m.parameters())
    assert n(unet) + n(enc) == 32_672_467 and n(refine) == 21_722_926
    # BatchNorm init of weight_initialization (minkunet.py:128-132)
    assert torch.all(unet.stem[1].bn.weight == 1) and torch.all(unet.stem[1].bn.bias == 0)
    # a reference-format checkpoint round-trips through the pipeline's loading convention
    ck = {"model." + k: v for k, v in unet.state_dict().items()}
    holder = torch.nn.Module()
    holder.model = minkunet.MinkUNetDiff(in_channels=3)
    missing = holder.load_state_dict(ck, strict=False)
    assert not missing.missing_keys and not missing.unexpected_keys


def test_sync_batchnorm_conversion():
    import lidiff_amd.MinkowskiEngine as ME
    from lidiff_amd import minkunet
    net = minkunet.MinkGlobalEnc(in_channels=3)
    net.stem[1].bn.running_mean.fill_(0.25)
    net = ME.MinkowskiSyncBatchNorm.convert_sync_batchnorm(net)
    bns = [m for m in net.modules() if isinstance(m, ME.MinkowskiBatchNorm)]
    from lidiff_amd.ops import SyncBatchNorm1d
    # the sync variant's child is an nn.BatchNorm1d subclass on the norm.hip kernels (not torch's SyncBatchNorm): same keys,
    # LiDiff's weight_initialization (minkunet.py:128-132, isinstance(m, nn.BatchNorm1d)) still finds it
    assert bns and all(isinstance(m, ME.MinkowskiSyncBatchNorm) and type(m.bn) is SyncBatchNorm1d
                       and isinstance(m.bn, torch.nn.BatchNorm1d) for m in bns)
    assert torch.all(net.stem[1].bn.running_mean == 0.25)
    assert "stem.1.bn.weight" in net.state_dict()


def test_scheduler_matches_oracle_on_cpu():
    from lidiff_amd.schedulers import DPMSolverMultistepScheduler
    g = np.load(os.path.join(GOLDEN, "dpm_trajectory.npz"))
    from oracle.dpm_solver import DpmSolverSdeOracle
    for n in (50, 8, 1):
        s = DPMSolverMultistepScheduler(num_train_timesteps=1000, beta_start=3.5e-5, beta_end=0.007,
                                        beta_schedule="linear", algorithm_type="sde-dpmsolver++", solver_order=2)
        s.set_timesteps(n)
        assert s.timesteps.tolist() == g[f"ts{n}"].tolist()
        x = torch.from_numpy(g[f"traj{n}"][0])
        o, xo = DpmSolverSdeOracle(), g[f"traj{n}"][0]
        o.set_timesteps(n)
        for i, t in enumerate(s.host_timesteps):
            xo = o.step(g[f"eps{n}"][i], t, xo, g[f"z{n}"][i])
            x = s.step(torch.from_numpy(g[f"eps{n}"][i]), torch.tensor(t), x, noise=torch.from_numpy(g[f"z{n}"][i]))["prev_sample"]
            assert x.dtype == torch.float64
            assert torch.allclose(x, torch.from_numpy(xo), rtol=1e-10, atol=1e-10)     # live oracle, same tables
            assert torch.allclose(x, torch.from_numpy(g[f"traj{n}"][i + 1]), rtol=1e-4, atol=1e-4)
    for name in ("timesteps", "betas", "alphas", "alphas_cumprod", "alpha_t", "sigma_t", "lambda_t", "sigmas"):
        assert isinstance(getattr(s, name), torch.Tensor)          # pipeline:58-66 moves exactly these
    # without injected noise the draw comes from torch's RNG
    s.set_timesteps(4)
    torch.manual_seed(0)
    a = s.step(torch.zeros(1, 8, 3), 999, torch.ones(1, 8, 3))["prev_sample"]
    assert a.shape == (1, 8, 3) and torch.isfinite(a).all()


def test_batched_coordinates_and_ply_io(tmp_path):
    import lidiff_amd.MinkowskiEngine as ME
    from lidiff_amd.pipeline import load_pcd, read_ply_points, write_ply_points
    a, b = torch.rand(5, 3), torch.rand(7, 3)
    bc = ME.utils.batched_coordinates([a, b], dtype=torch.float32)
    assert bc.shape == (12, 4) and bc[:5, 0].eq(0).all() and bc[5:, 0].eq(1).all() and torch.equal(bc[5:, 1:], b)
    pts = np.random.default_rng(0).standard_normal((100, 3))
    write_ply_points(str(tmp_path / "a.ply"), pts)
    assert np.array_equal(read_ply_points(str(tmp_path / "a.ply")), pts)
    with open(tmp_path / "b.ply", "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex 2\nproperty float x\nproperty float y\nproperty float z\n"
                "property uchar red\nend_header\n1 2 3 9\n4 5 6 9\n")
    assert read_ply_points(str(tmp_path / "b.ply")).tolist() == [[1, 2, 3], [4, 5, 6]]
    np.arange(8, dtype=np.float32).tofile(tmp_path / "c.bin")
    assert load_pcd(str(tmp_path / "c.bin")).tolist() == [[0, 1, 2], [4, 5, 6]]
    with pytest.raises(ValueError):
        load_pcd("x.pcd")
    ref_ply = "/root/reference/lidiff/Datasets/test/000123.ply"
    if os.path.exists(ref_ply):                        # build container only
        assert read_ply_points(ref_ply).shape == (125773, 3)


def test_farthest_point_sample_and_golden_scan(fps_scan):
    from lidiff_amd.pipeline import farthest_point_sample
    pts = torch.from_numpy(np.random.default_rng(1).standard_normal((300, 3)))
    sel = farthest_point_sample(pts, 20).numpy()
    assert sel[0] == 0 and len(set(sel.tolist())) == 20
    d = ((pts[:, None] - pts[sel[:1]][None]) ** 2).sum(-1).squeeze(1)
    assert sel[1] == int(d.argmax())
    assert fps_scan.shape == (18000, 3) and fps_scan.dtype == np.float32
    r = np.sqrt((fps_scan ** 2).sum(1))
    assert r.min() > 3.5 and r.max() < 50.0


def test_q_sample_schedule_constants():
    from lidiff_amd.diffusion import linear_beta_schedule
    b = linear_beta_schedule(1000, 3.5e-5, 0.007)
    acp = torch.cumprod(1 - b, 0)
    assert abs(float(torch.sqrt(1 - acp[999])) - 0.985) < 2e-3


def test_repeat_segments_matches_repeat_interleave_autograd():
    """The per-batch broadcast of the time embedding (minkunet.py:427-428): same forward and gradient as
    torch.repeat_interleave, with a segmented-sum backward."""
    from lidiff_amd.minkunet import _RepeatSegments
    g = torch.Generator().manual_seed(0)
    t = torch.randn(3, 5, generator=g, dtype=torch.float64, requires_grad=True)
    counts = [4, 0, 7]
    up = torch.randn(sum(counts), 5, generator=g, dtype=torch.float64)
    out = _RepeatSegments.apply(t, counts)
    ref_in = t.detach().clone().requires_grad_(True)
    ref = torch.repeat_interleave(ref_in, torch.tensor(counts), dim=0)
    assert torch.equal(out, ref)
    (out * up).sum().backward()
    (ref * up).sum().backward()
    assert torch.allclose(t.grad, ref_in.grad, rtol=1e-12, atol=0)


def test_batch_rows_matches_repeat_interleave_autograd():
    """The same broadcast from the map's batch column (no rows-per-batch read back to the host): forward and gradient of
    torch.repeat_interleave over batch-grouped rows, an empty batch included."""
    from lidiff_amd.minkunet import _BatchRows
    g = torch.Generator().manual_seed(1)
    t = torch.randn(3, 5, generator=g, dtype=torch.float64, requires_grad=True)
    counts = [4, 0, 7]
    bidx = torch.repeat_interleave(torch.arange(3), torch.tensor(counts))
    up = torch.randn(sum(counts), 5, generator=g, dtype=torch.float64)
    out = _BatchRows.apply(t, bidx)
    ref_in = t.detach().clone().requires_grad_(True)
    ref = torch.repeat_interleave(ref_in, torch.tensor(counts), dim=0)
    assert torch.equal(out, ref)
    (out * up).sum().backward()
    (ref * up).sum().backward()
    assert torch.allclose(t.grad, ref_in.grad, rtol=1e-12, atol=0)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` with WORLD_SIZE unset (how the driver starts the N = 1 line, and how a SCALE run in
    the same form would start N > 1) spawns N ranks itself: the --dry-run leg runs the launcher, the rendezvous on
    127.0.0.1, one all-reduce and the barrier / max-over-ranks timing on the CPU over gloo.  Without GPUs the real
    run is refused with a clear message instead of a stack trace."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                        # ONE line, from rank 0
    js = json.loads(lines[0])
    assert js["n_gpus"] == 2 and js["rccl_ranks_seen"] == 2 and js["dry_run"] and js["shards"] == [[0], [1]]
    assert js["elapsed_s"] >= 0.02                                # the max over ranks (rank 1 sleeps 20 ms)
    # the whole-scan form (configs[2] / [3]): bench.py --gpus N --pipeline --scans S shards N x S scans round-robin
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--pipeline", "--scans", "3", "--dry-run"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    js = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert js["pipeline"] and js["n_gpus"] == 2 and js["shards"] == [[0, 2, 4], [1, 3, 5]]
    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and "only 0 GPU(s) visible" in r.stderr


def test_step_boundary_kernels_contain_no_fused_multiply_add(tmp_path):
    """step.hip promises the torch sequence bit for bit -- every product and sum rounded on its own (the claim has to be
    checked on the generated code, whatever route built it): the gfx950 listing of the file, compiled as build.py compiles it,
    holds no v_fma / v_fmac / v_mad instruction; and the same source WITHOUT the flag and pragma would (the check can fail)."""
    import shutil
    import subprocess
    import sys
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    from lidiff_amd.csrc import build as hip_build
    src = os.path.join(ROOT, "lidiff_amd", "csrc", "step.hip")
    assert "-ffp-contract=off" in hip_build.EXTRA_FLAGS["step.hip"]

    def listing(path_in, flags, name):
        out = os.path.join(tmp_path, name)
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "--cuda-device-only", "-S", path_in,
                            "-o", out, "-I", os.path.dirname(src)] + flags, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        return subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_asm_regs.py"), "--no-fma", out],
                              capture_output=True, text=True, timeout=120)
    ok = listing(src, hip_build.EXTRA_FLAGS["step.hip"], "step.s")
    assert ok.returncode == 0 and ok.stdout.startswith("0 fused"), ok.stdout[:1500]
    loose = os.path.join(tmp_path, "step_contract.hip")
    with open(loose, "w") as f:
        f.write(open(src).read().replace("#pragma clang fp contract(off)", ""))
    bad = listing(loose, ["-ffp-contract=fast"], "step_contract.s")
    assert bad.returncode == 1 and not bad.stdout.startswith("0 fused"), "the check cannot see a contracted multiply-add"


def test_step_plan_is_pure_until_commit():
    """DPMSolverMultistepScheduler.step_plan() (the host half of the fused step boundary) changes nothing; commit(x0) applies the
    bookkeeping step() does -- a planned step whose launch raised leaves the scheduler usable, and a second-order
    plan without a committed predecessor is refused instead of silently falling back to first order."""
    import copy
    from lidiff_amd.schedulers import DPMSolverMultistepScheduler
    mk = lambda: DPMSolverMultistepScheduler(num_train_timesteps=1000, beta_start=3.5e-5, beta_end=0.007, beta_schedule="linear",
                                             algorithm_type="sde-dpmsolver++", solver_order=2)
    s, ref = mk(), mk()
    s.set_timesteps(50), ref.set_timesteps(50)
    ts = s.host_timesteps
    x = torch.zeros(1, 4, 3, dtype=torch.float64)
    for i, t in enumerate(ts[:4]):
        before = (list(s.model_outputs), s.lower_order_nums)
        p1, p2 = s.step_plan(t), s.step_plan(t)                       # planning twice == planning once
        assert (list(s.model_outputs), s.lower_order_nums) == before
        assert {k: v for k, v in p1.items() if k != "m_prev"} == {k: v for k, v in p2.items() if k != "m_prev"}
        assert (p1["m_prev"] is None) == (i == 0)
        x0 = torch.full_like(x, float(i + 1))
        s.commit(x0)
        ref.step(torch.zeros_like(x), t, x, noise=torch.zeros_like(x))
        assert s.lower_order_nums == ref.lower_order_nums
        assert [m is None for m in s.model_outputs] == [m is None for m in ref.model_outputs]
        if i:
            assert p1["m_prev"] is not None and float(p1["m_prev"].flatten()[0]) == float(i)
    broken = copy.copy(s)
    broken.model_outputs = [None, None]
    with pytest.raises(RuntimeError, match="previous data prediction"):
        broken.step_plan(ts[4])


@pytest.mark.parametrize("source", ["spconv_bf16.hip"])
def test_asm_kernel_isa_never_reads_an_in_flight_register(tmp_path, source):
    """spconv_bf16.hip requests its LDS fragments with inline asm and waits for them with counted s_waitcnt, so the
    compiler believes the destination registers valid the moment they are requested.  tools/check_asm_regs.py scans
    the generated gfx950 ISA for any instruction that reads such a register between its request and its wait (a copy
    or spill placed there by the register allocator silently multiplies stale data) -- and for spills at all."""
    import subprocess
    import sys
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "lidiff_amd", "csrc", source)
    out = os.path.join(tmp_path, "dense.s")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "--cuda-device-only", "-S", src,
                        "-o", out, "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "ScratchSize [bytes/lane]: 0" in r.stderr and "VGPRs Spill: 0" in r.stderr
    assert not re.search(r"(ScratchSize \[bytes/lane\]|VGPRs Spill): [1-9]", r.stderr)
    chk = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_asm_regs.py"), out], capture_output=True,
                         text=True, timeout=120)
    assert chk.returncode == 0 and chk.stdout.startswith("0 suspicious"), chk.stdout[:2000]


def test_split_operand_rule_respects_the_descriptor_range():
    """ops.split3_layer -- which layers of the fused plan take the split-operand kernel: tensor stride >= 4, widths the kernel takes,
    enough tiles to fill the chip, and split matrices that fit the kernel's 2 GiB buffer descriptors (a batch so large that
    rows x channels x 2 bytes x pieces passes 2^31 keeps the native kernel instead of failing in the C ABI's argument check)."""
    from lidiff_amd import ops
    assert ops.split3_layer(8, 120000, 2, 256, 0, 256, m_bound=180000)
    assert not ops.split3_layer(2, 120000, 2, 256, 0, 256, m_bound=180000)          # stride 2: low-density levels
    assert not ops.split3_layer(8, 120000, 2, 256, 0, 96, m_bound=180000)           # C_out % 64
    assert not ops.split3_layer(8, 20000, 1, 256, 0, 256, m_bound=180000)           # < 256 tiles
    assert not ops.split3_layer(8, 1300000, 2, 256, 0, 256, m_bound=1400000)        # 1.4 M rows x 256 x 6 B = 2.15 GB > 2^31
    with ops.split_pieces(2):
        assert ops.split3_layer(8, 1300000, 2, 256, 0, 256, m_bound=1400000)        # ... 4 B per element: fits
