#!/usr/bin/env python
"""bench.py -- denoising steps/sec on a 180k-point scan (BASELINE.json metric), MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]         (N > 1: spawns its N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W   (the driver's form; same ranks, same line)

Workload = BASELINE.json configs[1] ("single scan, T=50 DPM-Solver++ steps, fp32, 1x MI355X"):
the reference's bundled scan (lidiff/Datasets/test/000123.ply -> range filter -> FPS 18 000,
committed as tests/golden/scan_000123_fps18000.npy) tiled x10 = 180 000 points, voxel 0.05 m,
CFG weight 6.0, seeded random-init weights of the reference architecture (no checkpoints exist
offline).  One STEP = one iteration of completion_loop
(/root/reference/lidiff/tools/diff_completion_pipeline.py:158-167): voxelise x_t, x_cond and
x_uncond, two MinkGlobalEnc + MinkUNetDiff forwards, the DPM-Solver++ update, re-voxelisation.
Because trained weights are unavailable, the point offsets entering step i are sigma_i * N(0, I)
with sigma_i the scheduler's sigma_t at the T=50 trajectory's timestep i
-- the sparsity trajectory a trained model sees; K < 50 steps sample that trajectory evenly.
Inputs are resident in HBM before the timed region.  N > 1: every rank denoises its own scan
(different noise seed), no collective on the data path ("scaling": "weak").

Rank 0 prints ONE JSON line (see DESIGN.md section "Measurement").
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3       # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0     # same guide, "Peak BF16/FP16 MFMA ~2.5 PF dense"
PEAK_HBM_GBS = 8000.0
# what the path computes in when ops.SPLIT3 is on (the default): said in full, because it is NOT the plain fp32 MFMA everywhere
DTYPE_SPLIT3 = ("f32 in / f32 out / f32 accumulate everywhere; kernel_size-3 convolutions on maps of tensor stride >= 4 with C_out % 64 == 0 "
                "that fill the chip (>= 256 tiles), and the 1 x 1 shortcut with C_in >= 256 there: each fp32 operand cut into 3 bf16 pieces (exact sum), the 6 piece products with i + j <= 2 on "
                "v_mfma_f32_16x16x32_bf16, dropped terms < 2^-24 |x w| -- error vs float64 equal to the native fp32 MFMA kernel's "
                "(tests/test_gpu_kernels.py::test_spconv_split3_is_an_fp32_convolution; the whole GPU parity suite runs in this mode at "
                "the native kernel's bars); every other kernel: native fp32 (v_mfma_f32_16x16x4_f32 / VALU). LIDIFF_SPLIT3=0: native "
                "fp32 MFMA everywhere -- that number is `native_fp32` in this line")
# ... and when the opt-in two-piece fp16 mode is switched on (LIDIFF_SPLIT_PIECES=2): never the default, said in full as well
DTYPE_F16X2 = ("f32 in / f32 out / f32 accumulate; the split-operand layers (kernel_size-3 convolutions on maps of tensor stride >= 4 that fill "
               "the chip, the widest 1 x 1 shortcut) with each fp32 operand cut into 2 fp16 pieces (22 bits of the operand; weights pre-scaled "
               "by a power of two), the 3 products x0 w1, x1 w0, x0 w0 on v_mfma_f32_16x16x32_f16 -- half the matrix work of the default; "
               "OPT-IN (LIDIFF_SPLIT_PIECES=2 / ops.split_pieces(2)): measured errors against float64 equal the native fp32 MFMA kernel's "
               "(profiles/r06_f16x2.txt; the whole GPU suite passes under the switch), but the operands are 22-bit, not 24-bit, and fp16's "
               "range applies to the features (a scan that exceeds 65504 is redone on three bf16 pieces) -- which is why it is not the default")


def path_dtype(ops):
    return "f32" if not ops.SPLIT3 else DTYPE_F16X2 if ops.SPLIT_PIECES == 2 else DTYPE_SPLIT3


N_POINTS = 180000
T_STEPS = 50


def load_scan():
    path = os.path.join(ROOT, "tests", "golden", "scan_000123_fps18000.npy")
    return np.load(path).astype(np.float32)


def trajectory_index(j: int, k: int) -> int:
    """Which of the 50 trajectory positions step j of a K-step run uses."""
    return (j * T_STEPS // k) % T_STEPS if k < T_STEPS else j % T_STEPS


def build_pipeline(device, seed=42):
    from lidiff_amd.pipeline import DiffCompletion
    torch.manual_seed(seed)
    pipe = DiffCompletion(denoising_steps=T_STEPS, cond_weight=6.0, device=device)
    for m in pipe.modules():        # non-trivial eval-mode BatchNorm statistics
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    return pipe


def make_inputs(pipe, scan_np, steps, seed, device):
    """x_init [1,N,3] (float64 like the reference, App. D.1) and, per step, the noisy points."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    scan = torch.from_numpy(np.tile(scan_np, (10, 1))).double()[None]
    sig = pipe.dpm_scheduler.sigma_t.cpu()
    ts = pipe.dpm_scheduler.host_timesteps
    xs, tvals = [], []
    for j in range(steps):
        t = ts[trajectory_index(j, steps)]
        xs.append((scan + float(sig[t]) * torch.randn(scan.shape, generator=g, dtype=torch.float64)).to(device))
        tvals.append(t)
    return scan.to(device), xs, tvals


def run_steps(pipe, x_init, xs, tvals, first, count, cache_condition=False):
    """`count` denoising steps starting at schedule position `first`.  cache_condition: the SURVEY.md 8(f) row-1
    variant (conditions encoded once per scan) -- reported beside the metric, never as `value`."""
    # fields carry their role in the loop: with DiffCompletion.read_free only the FIRST pyramid of a role is built with a host read
    # (what completion_loop does for a scan); the check of what the read-free steps assumed is part of the timed work (below)
    pipe.read_free_reset()
    x_t = pipe.points_to_tensor(xs[first], role="x_t")
    x_cond = pipe.points_to_tensor(x_init, role="cond")
    x_uncond = pipe.points_to_tensor(torch.zeros_like(x_init), role="uncond")
    parts = pipe.encode_conditions(x_cond, x_uncond) if cache_condition else None
    for j in range(first, first + count):
        t = torch.full((1,), tvals[j], dtype=torch.int64, device=x_init.device)
        e_cond, e_uncond = pipe.classfree_pair(x_t, x_cond, x_uncond, t, parts, t_host=tvals[j])
        if j == first or tvals[j] >= tvals[j - 1]:
            pipe.new_scheduler()                       # a new scan's trajectory starts
        # the step's own boundary exactly as completion_loop runs it (guidance, DPM-Solver++ update with a fresh draw, the new
        # points' field) -- computed and dropped, because the open loop prescribes the next points: one MORE field construction
        # per step than the closed loop has, never less
        _ = pipe.step_boundary(x_init, x_t, e_cond, e_uncond, tvals[j])
        nxt = xs[j + 1] if j + 1 < len(xs) else xs[j]
        x_t = pipe.points_to_tensor(nxt, role="x_t")   # open loop: next sigma's points (see docstring)
        if parts is None:
            x_cond, x_uncond = pipe.reset_partial_pcd(x_cond, x_uncond, next_t=tvals[j + 1] if j + 1 < first + count else None)
    why = pipe.read_free_check()
    from lidiff_amd import ops
    ops.split_check()
    if why is not None:
        raise RuntimeError("host-read-free steps voided: " + why + " (LIDIFF_READ_FREE=0 runs every step with its host read)")
    return x_t


def cpu_baseline(scan_np, seed=42, threads=0):
    """The "MinkowskiEngine CPU path" timed on this box's host cores: the C++17 / OpenMP restatement
    of ME's CPU algorithm (oracle/cpp/me_cpu_ref.cpp: hash-map coordinate maps, per-offset hash-probe kernel maps, per-offset
    gather -> MKL SGEMM -> scatter-add in ascending kernel index) under the oracle's network code (torch-CPU Linear /
    BatchNorm, as the reference's own Python runs them).  Sample = BASELINE configs[0] (C1): ONE full denoising step of
    the 180k-point workload at T = 1 (t = 999, sigma = 0.985: the densest maps of the trajectory) -- CFG pair, DPM-Solver++
    update, re-voxelisation.  A bounded sample, not extrapolated."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import build_seeded_models, diffusion_state_dict
    from oracle import me_cpp
    from oracle import me_cpu as me
    from oracle import minkunet_cpu as net
    from oracle.dpm_solver import DpmSolverSdeOracle
    enc, unet, _ = build_seeded_models(seed)
    sd = diffusion_state_dict(enc, unet)
    o = DpmSolverSdeOracle()
    ts = o.set_timesteps(1)
    t = int(ts[0])
    rng = np.random.default_rng(seed)
    scan = np.tile(scan_np, (10, 1)).astype(np.float64)[None]
    x = scan + o.sigma_t[t] * rng.standard_normal(scan.shape)
    z = rng.standard_normal(scan.shape)
    if threads:
        me_cpp.set_threads(min(threads, os.cpu_count() or threads))
    cores = me_cpp.threads()
    model = "unknown CPU"
    try:
        model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except (OSError, StopIteration):
        pass
    with torch.no_grad(), me.use_cpp():
        t0 = time.perf_counter()
        xf = net.points_to_field(torch.from_numpy(x).float())
        cf = net.points_to_field(torch.from_numpy(scan).float())
        uf = net.points_to_field(torch.zeros(1, scan.shape[1], 3))
        eps = net.classfree_forward(sd, xf, cf, uf, torch.tensor([t]), w=6.0).numpy()
        x_new = scan + o.step(eps, t, xf.F.numpy().reshape(1, -1, 3) - scan, z)
        net.points_to_field(torch.from_numpy(x_new).float()).sparse()
        dt = time.perf_counter() - t0
    return {"value": 1.0 / dt, "unit": "steps/s", "cores": cores, "kind": "port", "cpu": model,
            "sample": f"C1: 1 full denoising step (CFG pair + DPM-Solver++ update + re-voxelisation) of the same 180k-point "
                      f"scan at T=1 (t={t}, sigma={o.sigma_t[t]:.3f}), {dt:.1f} s wall; C++17/OpenMP restatement of "
                      f"MinkowskiEngine's CPU algorithm with MKL SGEMM (oracle/cpp/me_cpu_ref.cpp), torch-CPU MLPs, "
                      f"{cores} OpenMP threads / {torch.get_num_threads()} torch threads on {model}"}


_STDOUT = None            # the real stdout once main() has pointed fd 1 at stderr


def emit(line: str):
    out = _STDOUT or sys.stdout
    out.write(line + "\n")
    out.flush()


VARIANT_KERNELS = {"bn128": ("<128, 8, 1",), "bn96": ("<128, 6, 1", "<128, 3, 2"), "bn64": ("<128, 4, 2",),
                   "bn32": ("<128, 2, 4",), "bn16": ("<128, 1, 8",), "rows": ("spconv_rows_kernel",), "thin": ("spconv_thin_kernel",),
                   "split3": ("spconv_fwd_split3_kernel",)}
COORD_KERNELS = ("insert_kernel", "flag_count_kernel", "scan_write_kernel", "inverse_kernel", "mean_", "kernel_map_",
                 "floor_kernel")


def pmc_profile():
    """The newest committed PMC passes (tools/pmc_bench.sh -> profiles/rNN_pmc_traffic.json) or (None, None)."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True)
    return (json.load(open(paths[0])), os.path.relpath(paths[0], ROOT)) if paths else (None, None)


def mfma_busy_profile():
    """MFMA-pipe occupancy per layer class from the newest committed PMC session (tools/pmc_mfma.sh ->
    profiles/rNN_pmc_mfma.json): SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) -- static, like `traffic`."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_mfma.json")), reverse=True)
    if not paths:
        return None
    js = json.load(open(paths[0]))
    return {"source": "static: " + os.path.relpath(paths[0], ROOT) + " (rocprofv3 PMC over tools/conv_probe.py, one process per layer, "
                      "bench scan at sigma 1, CFG pair stacked; counters cannot be read inside the timed process)",
            "layers": {k: {"mfma_busy": round(v["mfma_busy"], 3), "tflops": v["tflops"], "kernel": v["kernel"][:48]}
                       for k, v in js["layers"].items()}}


def dense_gemm_tflops(dev, m=8192, k=8192, n=16384, iters=10):
    """What a dense bf16 GEMM reaches on this chip through torch.matmul (hipBLASLt): the practical ceiling of the bf16 matrix pipe."""
    a = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
    b = torch.randn(k, n, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        a @ b
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        a @ b
    e.record()
    torch.cuda.synchronize()
    return 2.0 * m * k * n * iters / (s.elapsed_time(e) * 1e-3) / 1e12


def traffic_from_profile(variants, per="launch"):
    """HBM-side bytes of the conv variants `variants` from the committed PMC passes: FETCH_SIZE x2 (gfx950 correction) +
    WRITE_SIZE, summed over their kernel instantiations -- per launch, or in total over the profiled command.  PMC counters
    cannot be collected from inside this process, so this is a STATIC number of the same kernels on the same command,
    labelled as such in the line.  (None, None) if no profile holds those kernels."""
    js, src = pmc_profile()
    if js is None:
        return None, None
    if "kernels" not in js:                                    # round-1/2 format: the dominant kernel only
        want = VARIANT_KERNELS.get(variants[0], ("?",))[0]
        return (js["traffic_bytes_per_launch"] / 1e9, src) if want in js.get("kernel", "") else (None, None)
    keys = [p for v in variants for p in VARIANT_KERNELS.get(v, ())]
    rows = [v for k, v in js["kernels"].items() if ("spconv_fwd_kernel" in k or "spconv_rows_kernel" in k or "spconv_thin_kernel" in k or "spconv_fwd_split3_kernel" in k)
            and any(p in k for p in keys)]
    if not rows:
        return None, None
    total = sum(r["traffic_bytes_total"] for r in rows)
    return (total / sum(r["launches"] for r in rows) if per == "launch" else total) / 1e9, src


def coords_traffic_from_profile():
    """(GB per denoising step moved by the coordinate kernels of ALL three fields of a step, source) from the same passes."""
    js, src = pmc_profile()
    if js is None or "kernels" not in js:
        return None, None
    steps = [int(x) for x in __import__("re").findall(r"--steps (\d+) --warmup (\d+)", js.get("command", ""))[0]] \
        if "--steps" in js.get("command", "") else None
    rows = [v for k, v in js["kernels"].items() if any(c in k for c in COORD_KERNELS)]
    if not rows or not steps:
        return None, None
    return sum(r["traffic_bytes_total"] for r in rows) / sum(steps) / 1e9, src


def coords_roofline(scan_np, device, iters=5):
    """The coordinate pipeline of ONE 180k-point x_t on its own, timed with HIP events on the launch stream: voxel hash
    (unique / inverse / mean), four strided maps, five kernel_size-3 tables, four kernel_size-2 tables and their four
    transposed tables -- the HBM-bound integer part of a step.  Algorithmic bytes: voxelise 16N + 12N +
    8N + 28M0; strided map 16 M_l + 16 M_(l+1) + 4 M_l; kernel map 16 (M_in + M_out) + 8 P."""
    import lidiff_amd.MinkowskiEngine as ME
    rng = np.random.default_rng(0)
    pts = torch.from_numpy(np.tile(scan_np, (10, 1)) + rng.standard_normal((N_POINTS, 3)).astype(np.float32)).to(device)
    coord = torch.cat([torch.zeros(N_POINTS, 1, device=device), torch.round(pts / 0.05)], 1)

    def build():
        f = ME.TensorField(features=pts, coordinates=coord, device=device)
        f.coordinate_manager.pyramid = True          # the pipeline's path: device-side row counts, ONE host read (ops.build_pyramid)
        f.sparse()
        mgr = f.coordinate_manager
        ts = 1
        for _ in range(4):
            mgr.kernel_map(ts, ts, 3)
            nxt = mgr.stride(ts, 2)
            mgr.kernel_map(ts, nxt, 2)
            mgr.kernel_map(nxt, ts, 2, True)
            ts = nxt
        mgr.kernel_map(ts, ts, 3)
        return mgr
    mgr = build()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        build()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    m = [mgr.maps[1 << l].coords.shape[0] for l in range(5)]
    nbytes = 36.0 * N_POINTS + 28.0 * m[0]
    for l in range(5):
        p3 = int((mgr.kernel_map(1 << l, 1 << l, 3) >= 0).sum())
        nbytes += 32.0 * m[l] + 8.0 * p3
        if l < 4:
            nbytes += 36.0 * m[l] + 16.0 * m[l + 1]                      # strided map
            nbytes += 2 * (16.0 * (m[l] + m[l + 1]) + 8.0 * m[l])        # ks2 table and its transpose (P = M_l)
    gbs = nbytes / (ms * 1e-3) / 1e9
    return {"kernel": "coordinate pipeline of one 180000-point x_t (sigma 1): voxel hash + mean, 4 strided maps, 5 ks3 / 4 ks2 / "
                      "4 transposed kernel maps (coords.hip; the pyramid's row counts stay on the device: one host read of all sizes)",
            "bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
            "traffic": coords_traffic_from_profile()[0],
            "traffic_unit": "GB per denoising step moved by the coordinate kernels of ALL three fields of a step (x_t, x_cond, x_uncond: "
                            "hash insert / flag / scan / inverse / mean / kernel maps; rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE; static: "
                            + str(coords_traffic_from_profile()[1]) + ") -- `achieved` is one x_t's pyramid alone",
            "ms": ms, "algorithmic_mbytes": nbytes / 1e6, "voxels_per_level": m, "host_reads": 1}


def train_leg(scan_np, device, steps=3, warmup=1):
    """Beside the metric, never `value`: BASELINE configs[4]'s per-GPU shape -- DiffusionPoints.training_step (models.py:180-217)
    forward + loss + backward + Adam on B = 2 scans of 180 000 points with 18 000-point partial scans, random-init weights,
    fp32 and bf16 (bf16 conv / MLP GEMM operands, fp32 accumulation and state) -- timed with HIP events on the launch stream.
    Per precision: ms per step and per phase, and the time inside the sparse-conv kernels by role (forward + input-gradient
    launches, weight-gradient launches; events around every launch: a few % of overhead, stated)."""
    from lidiff_amd import ops
    from lidiff_amd.diffusion import DiffusionPoints
    rng = np.random.default_rng(0)
    scan = scan_np.astype(np.float32)
    batches = []
    for _ in range(2):
        part = np.stack([scan + 0.01 * rng.standard_normal(scan.shape).astype(np.float32) for _ in range(2)])
        full = np.tile(part, (1, 10, 1)) + 0.05 * rng.standard_normal((2, 10 * scan.shape[0], 3)).astype(np.float32)
        batches.append({"pcd_full": torch.from_numpy(full), "pcd_part": torch.from_numpy(part)})
    ev = lambda: torch.cuda.Event(enable_timing=True)
    out = {"workload": "configs[4] per-GPU shape: B = 2 x 180000 points, 18000-point partial scans, forward + loss + backward "
                       "+ Adam, random-init weights (tools/train_probe.py is the same step)", "steps": steps, "warmup": warmup}
    import torch.distributed as tdist
    import lidiff_amd.MinkowskiEngine as ME
    own_group = False
    for precision in ("32", "bf16", "bf16_syncbn"):
        sync = precision.endswith("_syncbn")
        precision = precision.split("_")[0]
        torch.manual_seed(0)
        module = DiffusionPoints(device=device, precision=precision)
        if sync:
            # configs[4] runs under convert_sync_batchnorm (train.py:90): the same step with every BatchNorm on the synchronised
            # path of norm.hip -- per-channel fp64 sums, ONE all-reduce per layer forward and backward over RCCL -- in a process
            # group of this one rank (the 1-GPU box): all the launches and collectives of the multi-rank step, minus the wire
            if not tdist.is_initialized():
                import socket
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    port = sk.getsockname()[1]
                tdist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                         device_id=device)
                own_group = True
            ops.SyncBatchNorm1d.sync_single_rank = True
            ME.MinkowskiSyncBatchNorm.convert_sync_batchnorm(module)
        module.train()
        opt, _ = module.configure_optimizers()
        gen = torch.Generator(device=device).manual_seed(1)
        tot = {"forward_loss_ms": 0.0, "backward_ms": 0.0, "optimizer_ms": 0.0}
        torch.cuda.reset_peak_memory_stats()
        conv_ms = dw_ms = conv_flops = 0.0
        for step in range(warmup + steps):
            timed = step >= warmup
            prof = ops.ConvProfiler(None) if timed else None
            ops.PROFILER = prof
            e = [ev() for _ in range(4)]
            e[0].record()
            loss = module.training_step(batches[step % 2], step, generator=gen)
            e[1].record()
            opt.zero_grad(set_to_none=True)
            loss.backward()
            e[2].record()
            opt.step()
            e[3].record()
            torch.cuda.synchronize()
            ops.PROFILER = None
            if timed:
                tot["forward_loss_ms"] += e[0].elapsed_time(e[1])
                tot["backward_ms"] += e[1].elapsed_time(e[2])
                tot["optimizer_ms"] += e[2].elapsed_time(e[3])
                conv_ms += sum(a.elapsed_time(b) for _, a, b, *_ in prof.launches if a is not None)
                dw_ms += prof.dw_ms()
                conv_flops += sum(d["flops_timed"] for d in prof.summary().values())        # 2 P C_in C_out per forward / dX launch
        ms = {k: v / steps for k, v in tot.items()}
        total = sum(ms.values())
        out[("f32" if precision == "32" else "bf16") + ("_syncbn" if sync else "")] = {
            "ms_per_step": total, "steps_per_s": 1e3 / total, "scans_per_s": 2e3 / total, **ms,
            "conv_fwd_dx_kernels_ms": conv_ms / steps, "conv_dw_kernels_ms": dw_ms / steps,
            # the step's convolution classes against the matrix peak of their operand type (algorithmic flops 2 P C_in
            # C_out per launch, HIP-event time of the launches; the weight gradient does the forward launches' flops once more)
            "roofline_conv_fwd_dx": _mfma_roofline(conv_flops / steps, conv_ms / steps, precision),
            "roofline_conv_dw": _mfma_roofline(0.5 * conv_flops / steps, dw_ms / steps, precision),
            "other_ms": total - (conv_ms + dw_ms) / steps,
            "peak_memory_gib": torch.cuda.max_memory_allocated() / 2 ** 30, "final_loss": float(loss.detach())}
        if sync:
            out["bf16_syncbn"]["sync_batchnorm_layers"] = sum(type(m) is ops.SyncBatchNorm1d for m in module.modules())
            out["bf16_syncbn"]["collectives_per_step"] = 2 * out["bf16_syncbn"]["sync_batchnorm_layers"]
            out["bf16_syncbn"]["backend"] = tdist.get_backend()
            ops.SyncBatchNorm1d.sync_single_rank = False
        del module, opt, loss
        torch.cuda.empty_cache()
    if own_group:
        tdist.destroy_process_group()
    out["note"] = ("other_ms = BatchNorm (norm.hip), conditioning / head MLP GEMMs (hipBLASLt), coordinate maps, loss, Adam and "
                   "launch gaps; data-parallel training adds one bucketed gradient all-reduce per step (lidiff_amd/dist.py), launched "
                   "per bucket from backward hooks on gradients that live in the buckets")
    # the step by kernel class: static, from the newest committed rocprofv3 kernel trace of tools/train_probe.py
    # (tools/gpu_train_profile.sh + tools/train_classes.py) -- counters / traces cannot be taken inside the timed process
    import glob
    for key, tag in (("bf16", "bf16"), ("f32", "32")):
        paths = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_train_kernel_classes_{tag}.json")), reverse=True)
        if paths and key in out:
            with open(paths[0]) as f:
                js = json.load(f)
            out[key]["kernel_classes_ms_per_step"] = {k: round(v["ms_per_step"], 2) for k, v in js["classes"].items()}
            out[key]["kernel_classes_source"] = "static: " + os.path.relpath(paths[0], ROOT) + " (device-busy %.1f ms per step)" % js["device_busy_ms_per_step"]
    out["bf16_vs_fp32_note"] = ("bf16 is pinned per kernel and per block against the oracle's emulation (profiles/r05_parity_errors.txt, "
                                "bf16_block: 1 - cos <= 2.7e-4 on every gradient); the whole bf16 step tracks the fp32 step with a median "
                                "parameter-gradient cosine of 0.86 (worst 0.71) at random initialisation "
                                "(tests/test_gpu_network.py::test_bf16_training_step_tracks_the_fp32_step)")
    return out


def _mfma_roofline(flops_per_step, ms_per_step, precision):
    peak = 157.3 if precision == "32" else 2500.0            # dense fp32 / bf16 MFMA peak (MI355X_MICROARCH.md), TFLOP/s
    tf = flops_per_step / (ms_per_step * 1e-3) / 1e12 if ms_per_step > 0 else 0.0
    return {"bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak,
            "gflop_per_step": flops_per_step / 1e9, "ms_per_step": ms_per_step}


def train_refine_leg(scan_np, device, items=2, steps=2, warmup=1):
    """Beside the metric, never `value`: RefineDiffusion.training_step (models_refine.py:53-76) at config_refine.yaml's per-item
    size -- 180 000 noisy points in, 6 x 180 000 predicted against 2 x 180 000 target points in the Chamfer loss (the grid search
    of lidiff_nn_dist_grid: the exhaustive one needs 7.8e11 distance evaluations per item) -- B = `items` (the config's 8 is a
    matter of memory-time only: items are independent in the loss and share nothing but BatchNorm statistics; round 6: the bench
    runs the config's B = 8 when the device has 160 GiB free)."""
    from lidiff_amd import ops
    from lidiff_amd.diffusion import RefineDiffusion, chamfer_distance
    rng = np.random.default_rng(1)
    base = np.tile(scan_np.astype(np.float32), (10, 1))
    full = np.stack([np.concatenate([base, base + 0.04 * rng.standard_normal(base.shape).astype(np.float32)]) for _ in range(items)])
    noise = np.stack([base + 0.1 * rng.standard_normal(base.shape).astype(np.float32) for _ in range(items)])
    batch = {"pcd_noise": torch.from_numpy(noise), "pcd_full": torch.from_numpy(full)}
    torch.manual_seed(0)
    module = RefineDiffusion(device=device)
    module.train()
    opt = module.configure_optimizers()
    ev = lambda: torch.cuda.Event(enable_timing=True)
    tot = [0.0, 0.0, 0.0]
    torch.cuda.reset_peak_memory_stats()
    for step in range(warmup + steps):
        e = [ev() for _ in range(4)]
        e[0].record()
        loss = module.training_step(batch, step)
        e[1].record()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        e[2].record()
        opt.step()
        e[3].record()
        torch.cuda.synchronize()
        if step >= warmup:
            for i in range(3):
                tot[i] += e[i].elapsed_time(e[i + 1])
    # the loss alone (both directed searches + the differentiable distances), forward + backward, per item
    pred = (torch.from_numpy(np.repeat(noise[:1], 6, axis=1)).to(device) + 0.05 * torch.randn(1, 6 * base.shape[0], 3, device=device))
    pred.requires_grad_(True)
    tgt = torch.from_numpy(full[:1]).to(device)
    cd = []
    for _ in range(3):
        a, b = ev(), ev()
        a.record()
        chamfer_distance(pred, tgt).backward()
        b.record()
        torch.cuda.synchronize()
        cd.append(a.elapsed_time(b))
    total = sum(tot) / steps
    out = {"workload": f"config_refine.yaml per-item shape, B = {items}: 180000 noisy points -> MinkUNet -> 6 x 180000 predicted vs "
                       "2 x 180000 target points, Chamfer loss (pytorch3d defaults), backward, Adam; random-init weights",
           "ms_per_step": total, "ms_per_item": total / items, "forward_loss_ms": tot[0] / steps, "backward_ms": tot[1] / steps,
           "optimizer_ms": tot[2] / steps, "chamfer_fwd_bwd_ms_per_item": min(cd),
           "nearest_neighbour": f"lidiff_nn_dist_grid, cell {ops.NN_GRID_CELL} m (exact: bit-identical to the exhaustive search)",
           "peak_memory_gib": torch.cuda.max_memory_allocated() / 2 ** 30, "final_loss": float(loss.detach()), "steps": steps}
    del module, opt, loss
    torch.cuda.empty_cache()
    return out


def load_raw_scan():
    """The reference's bundled scan after its range filter (119 035 points): what preprocess_scan's FPS takes in."""
    return np.load(os.path.join(ROOT, "tests", "golden", "scan_000123_range_filtered.npy")).astype(np.float64)


def pipeline_leg(pipe, device, seeds, warm=True):
    """DiffCompletion.complete_scan (pipeline:117-132, timed per scan as pipeline:198-203 does): range filter + farthest-point
    sampling to 18 000 points, x10 + N(0, I), the CLOSED T = 50 CFG denoising loop, post-filter, refinement forward -- seeded
    random-init weights, one scan per seed.  Returns the per-scan seconds and the phase sums."""
    raw = load_raw_scan()
    if warm:                                                   # kernels / allocator / packed weights: one short scan, untimed
        saved = pipe.hparams["diff"]["s_steps"]
        pipe.hparams["diff"]["s_steps"] = 2
        pipe.complete_scan(raw, generator=torch.Generator(device=device).manual_seed(1))
        pipe.hparams["diff"]["s_steps"] = saved
        torch.cuda.synchronize()
    per_scan, phases, rows = [], {}, []
    for seed in seeds:
        gen = torch.Generator(device=device).manual_seed(int(seed))
        torch.manual_seed(int(seed))                           # the scheduler's draws (torch.randn without a generator, as upstream)
        t0 = time.perf_counter()
        refined, diffused = pipe.complete_scan(raw, generator=gen, timings=phases)
        torch.cuda.synchronize()
        per_scan.append(time.perf_counter() - t0)
        rows.append((int(diffused.shape[0]), int(refined.shape[0]), bool(np.isfinite(refined).all())))
    return per_scan, phases, rows


def pipeline_main(args, rank, world, device):
    """`bench.py --gpus N --pipeline [--scans S]`: BASELINE configs[2] / [3]'s unit of work -- whole scans (FPS + T = 50 +
    refinement), scans sharded round-robin over the ranks (dist.shard_items: scan i -> rank i mod N), no collective on the data
    path; barrier + synchronize on both sides, MAX over ranks, ONE line from rank 0: scans/s of the whole job and s/scan."""
    from lidiff_amd import dist as ldist
    from lidiff_amd import ops
    ranks_seen = int(ldist.sum_over_ranks(1.0, device=device))
    pipe = build_pipeline(device)
    total = world * args.scans
    mine = ldist.shard_items(total, rank, world)
    with torch.no_grad():
        pipeline_leg(pipe, device, [], warm=True)
        ldist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        per_scan, phases, rows = pipeline_leg(pipe, device, [5000 + i for i in mine], warm=False)
        torch.cuda.synchronize()
        ldist.barrier()
        elapsed = ldist.max_over_ranks(time.perf_counter() - t0, device=device)
    if rank != 0:
        return
    emit(json.dumps({
        "metric": "completed scans/sec (FPS + T=50 CFG denoising + refinement) on 180k-pt scans", "value": total / elapsed,
        "unit": "scans/s", "n_gpus": world, "rccl_ranks_seen": ranks_seen, "visible_devices": torch.cuda.device_count(),
        "scans": total, "scans_per_gpu": args.scans, "s_per_scan": elapsed / args.scans, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": path_dtype(ops), "data": "synthetic",
        "denoising_steps_per_s": total * T_STEPS / elapsed,
        "config": {"workload": "configs[2]: independent 180000-point scans (bundled scan, own noise seed each) sharded one per GPU, "
                               "DiffCompletion.complete_scan = range filter + FPS 18000 + T=50 sde-dpmsolver++ CFG loop (closed) + "
                               "post-filter + MinkUNet refinement, fp32, seeded random-init weights",
                   "parallelism": f"scan-sharded x{world}, no data-path collective", "shard_of_rank0": mine},
        "rank0": {"s_per_scan": per_scan, "phases_s": phases, "rows_diffused_refined_finite": rows}}))


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` outside a torchrun environment: re-execute under torch.distributed.run with one
    rank per GPU on 127.0.0.1 (the form the driver uses itself) and hand its exit code back.  A box with fewer
    visible GPUs than ranks is refused here, with a clear message, before anything is launched."""
    import socket
    import subprocess
    if not args.dry_run and os.environ.get("LIDIFF_BENCH_SHARE_GPU") != "1":
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible on this box "
                             f"(torch.cuda.device_count() = {have}); one rank per GPU, no oversubscription")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def dry_run(args, rank, world):
    """The N-rank plumbing without kernels (CPU, gloo): rendezvous, one all-reduce, barrier-bracketed timing with
    the max over ranks, one JSON line from rank 0 -- what tests/test_host.py runs in the build container."""
    import torch.distributed as tdist
    from lidiff_amd import dist as ldist
    seen = int(ldist.sum_over_ranks(1.0))
    ldist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (1 + rank))                       # "steps" of unequal length: the slowest rank sets the time
    ldist.barrier()
    elapsed = ldist.max_over_ranks(time.perf_counter() - t0)
    if rank == 0:
        print(json.dumps({"metric": "denoising steps/sec on 180k-pt scan", "dry_run": True, "n_gpus": world,
                          "rccl_ranks_seen": seen, "backend": tdist.get_backend() if world > 1 else None,
                          "steps": args.steps, "warmup": args.warmup, "elapsed_s": elapsed, "pipeline": bool(args.pipeline),
                          "shards": [ldist.shard_items(world * (args.scans if args.pipeline else 1), r, world) for r in range(world)]}))
    if world > 1:
        tdist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-kernel-events", action="store_true", help="skip per-launch HIP events (roofline leg)")
    ap.add_argument("--all-variants", action="store_true",
                    help="time every sparse-conv launch, not only the dominant (BN=128) variant: more events in the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=32,
                    help="threads of the CPU baseline (0 = all the box has).  Default 32: on the 128-thread EPYC 9575F of the "
                         "GPU box the leg takes 23.6 s with 32 threads, 37.8 s with 64, 81.7 s with 128 "
                         "(profiles/r02_cpu_baseline_threads.txt) -- the fastest configuration is the one reported")
    ap.add_argument("--no-coords-roofline", action="store_true", help="skip the coordinate-pipeline (HBM-bound) roofline leg")
    ap.add_argument("--cached-condition", action="store_true",
                    help="also time the same steps with the step-invariant conditions encoded once (reported beside the metric)")
    ap.add_argument("--layer-table", default="",
                    help="write the per-layer table of the all-variants pass (ms per step, pairs, TFLOP/s per conv shape) to this file")
    ap.add_argument("--no-train", action="store_true",
                    help="skip the 'train' leg (configs[4]'s per-GPU training step, fp32 and bf16: 4 steps each)")
    ap.add_argument("--no-alt", action="store_true",
                    help="skip the 'native_fp32' leg (the same steps with every layer on the native fp32-MFMA kernel)")
    ap.add_argument("--pipeline", action="store_true",
                    help="time whole scans instead of denoising steps: DiffCompletion.complete_scan (FPS + T = 50 + refinement) on "
                         "--scans scans per rank, sharded over the ranks (BASELINE configs[2] / [3]); prints scans/s and s/scan")
    ap.add_argument("--scans", type=int, default=1, help="--pipeline: scans per rank")
    ap.add_argument("--no-closed-loop", action="store_true",
                    help="skip the 'closed_loop' leg (one complete_scan with seeded weights: s per scan, beside the metric)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous / timing plumbing only, on the CPU over gloo (no kernels): CI of the N > 1 path")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args, sys.argv[1:]))
    from lidiff_amd import dist as ldist
    if not args.dry_run:
        # ONE line on stdout: native libraries write there too (RCCL prints its version banner when a communicator is created), so the
        # process's fd 1 points at stderr for the whole run and the JSON line goes to the saved descriptor at the end
        global _STDOUT
        sys.stdout.flush()
        _STDOUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
    # (LIDIFF_BENCH_BACKEND=gloo with LIDIFF_BENCH_SHARE_GPU=1: the N-rank code path rehearsed on a box with fewer GPUs than ranks --
    # the ranks share device 0 and rendezvous over gloo; RCCL itself cannot place two ranks on one device.  Testing only.)
    rank, world, local = ldist.init_from_env("gloo" if args.dry_run else os.environ.get("LIDIFF_BENCH_BACKEND", "nccl"))
    if os.environ.get("LIDIFF_BENCH_SHARE_GPU") == "1" and not args.dry_run:
        local = local % max(1, torch.cuda.device_count())
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size must equal --gpus")
    if args.dry_run:
        return dry_run(args, rank, world)
    from lidiff_amd import ops
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    if local >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local} but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:       # N ranks share the host: each keeps to its share of the cores (the step's host side is one Python thread)
        torch.set_num_threads(max(1, (os.cpu_count() or world) // world))
    if args.pipeline:
        return pipeline_main(args, rank, world, device)
    # one tiny collective over RCCL before anything is timed: every rank must be seen (the data path itself has none)
    ranks_seen = int(ldist.sum_over_ranks(1.0, device=device))

    scan_np = load_scan()
    pipe = build_pipeline(device)
    total = args.warmup + args.steps
    x_init, xs, tvals = make_inputs(pipe, scan_np, args.steps, seed=1000 + rank, device=device)
    # warmup steps reuse the first positions of the schedule
    wx, wt = xs[:max(1, min(args.warmup, len(xs)))], tvals[:max(1, min(args.warmup, len(xs)))]

    with torch.no_grad():
        for w in range(args.warmup):
            run_steps(pipe, x_init, wx, wt, w % len(wx), 1)
        torch.cuda.synchronize()
        # (every 3rd launch of the dominant variant carries events: 40 launches per step, so over the steps every layer is
        # sampled; all of them cost ~0.4 ms per step of event records in the timed region)
        prof = None if args.no_kernel_events else ops.ConvProfiler(None if args.all_variants else {"bn128", "split3"},
                                                                   sample=1 if args.all_variants else 3)
        ops.PROFILER = prof
        ldist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        x_last = run_steps(pipe, x_init, xs, tvals, 0, args.steps)
        torch.cuda.synchronize()
        ldist.barrier()
        elapsed = time.perf_counter() - t0
        ops.PROFILER = None
        x_last.coordinate_manager.check()
        elapsed_cached = 0.0
        if args.cached_condition:      # beside the metric: the same K steps with the conditions encoded once (not `value`)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run_steps(pipe, x_init, xs, tvals, 0, args.steps, cache_condition=True)
            torch.cuda.synchronize()
            elapsed_cached = time.perf_counter() - t0
        # a second, untimed-for-the-metric pass of the same K steps with EVERY conv launch carrying events: the narrow
        # variants (the low-density stride-1/2 layers) get their own roofline entry, bound by HBM (SURVEY.md 8d bytes)
        vprof = None
        if world == 1 and not args.no_kernel_events and not args.all_variants:
            vprof = ops.ConvProfiler(None)
            ops.PROFILER = vprof
            run_steps(pipe, x_init, xs, tvals, 0, args.steps)
            torch.cuda.synchronize()
            ops.PROFILER = None
            if args.layer_table:
                with open(args.layer_table, "w") as f:
                    f.write(f"# per conv shape over {args.steps} steps (all launches timed with HIP events): python bench.py --layer-table\n")
                    for r in ops.layer_table(vprof, args.steps):
                        f.write(f"{r['variant']:<6} k={r['k']:<2} {r['c_in']:>3}->{r['c_out']:<3} launches/step {r['launches_per_step']:5.1f}  "
                                f"ms/step {r['ms_per_step']:6.3f}  avg {r['avg_us']:7.1f} us  rows {r['avg_rows']:9.0f}  pairs {r['avg_pairs']:10.0f}  "
                                f"{r['tflops']:6.1f} TFLOP/s\n")
        # the dominant variant once more with NOTHING running beside it (every stream trick off: one queue): its launch durations
        # in the timed pass above include whatever the side streams run concurrently (the next step's condition encoders, x_t's
        # map building) -- good for the step, but it makes the kernel look slower than it is
        sprof = None
        if world == 1 and not args.no_kernel_events and prof is not None:
            saved = {k: getattr(pipe, k) for k in ("overlap_maps", "lazy_x_t", "encode_ahead")}
            pipe.overlap_maps = pipe.lazy_x_t = pipe.encode_ahead = False
            run_steps(pipe, x_init, wx, wt, 0, 1)
            psum = prof.summary()
            dom_variant = max(psum, key=lambda v: psum[v]["ms"])
            sprof = ops.ConvProfiler({dom_variant})
            ops.PROFILER = sprof
            run_steps(pipe, x_init, xs, tvals, 0, args.steps)
            torch.cuda.synchronize()
            ops.PROFILER = None
            for k, v in saved.items():
                setattr(pipe, k, v)
        # beside the metric: the same K steps with EVERY layer on the native fp32-MFMA kernel (ops.split3(False) / LIDIFF_SPLIT3=0)
        alt = None
        if world == 1 and not args.no_alt and ops.SPLIT3:
            with ops.split3(False):
                run_steps(pipe, x_init, wx, wt, 0, 1)
                aprof = None if args.no_kernel_events else ops.ConvProfiler({"bn128"}, sample=3)
                ops.PROFILER = aprof
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                run_steps(pipe, x_init, xs, tvals, 0, args.steps)
                torch.cuda.synchronize()
                alt = (time.perf_counter() - t0, aprof)
                ops.PROFILER = None
        # ... and with the split-operand layers on TWO fp16 pieces per operand (opt-in, ops.split_pieces(2) / LIDIFF_SPLIT_PIECES=2)
        alt16 = None
        if world == 1 and not args.no_alt and ops.SPLIT3 and ops.SPLIT_PIECES == 3:
            with ops.split_pieces(2):
                run_steps(pipe, x_init, wx, wt, 0, 1)
                hprof = None if args.no_kernel_events else ops.ConvProfiler({"split3"}, sample=3)
                ops.PROFILER = hprof
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                run_steps(pipe, x_init, xs, tvals, 0, args.steps)
                torch.cuda.synchronize()
                alt16 = (time.perf_counter() - t0, hprof)
                ops.PROFILER = None
    elapsed = ldist.max_over_ranks(elapsed, device=device)
    if args.cached_condition:
        elapsed_cached = ldist.max_over_ranks(elapsed_cached, device=device)

    if rank != 0:
        return
    out = {
        "metric": "denoising steps/sec on 180k-pt scan", "value": world * args.steps / elapsed, "unit": "steps/s",
        "n_gpus": world, "rccl_ranks_seen": ranks_seen, "visible_devices": torch.cuda.device_count(),
        "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": path_dtype(ops), "data": "synthetic",
        "config": {"workload": "configs[1]: one 180000-point scan (bundled scan FPS 18000 x10), voxel 0.05 m, "
                               "T=50 sde-dpmsolver++ trajectory, CFG w=6 (2 forwards/step), fp32, "
                               "random-init weights, offsets sigma_t*N(0,I) per step",
                   "points": N_POINTS, "trajectory_positions": [trajectory_index(j, args.steps) for j in range(args.steps)],
                   "scans_per_gpu": 1, "parallelism": f"scan-sharded x{world}, no data-path collective",
                   "host_reads_per_step": 0 if pipe.read_free else 3,
                   "host_reads_note": ("DiffCompletion.read_free: from every role's second pyramid on no map size reaches the host inside a "
                                       "step (row counts stay on the device, sizes arrive through pinned memory one step later and are "
                                       "validated inside the timed region); the first step of a run reads them (3 pyramids)"
                                       if pipe.read_free else "LIDIFF_READ_FREE=0: one blocking read per coordinate pyramid")},
    }
    if args.cached_condition:
        out["cached_condition"] = {
            "value": world * args.steps / elapsed_cached, "unit": "steps/s",
            "ms_per_step": 1e3 * elapsed_cached / args.steps,
            "note": "same steps with partial_enc(x_cond), partial_enc(x_uncond) encoded once per scan (SURVEY.md 8(f) "
                    "row 1, DiffCompletion.cache_condition; bit-identical outputs); informational, the metric above "
                    "recomputes them every step as the reference does"}
    if prof is not None:
        summ = prof.summary()
        dom = max(summ, key=lambda v: summ[v]["ms"])
        d = summ[dom]
        tflops = d["flops_timed"] / (d["ms"] * 1e-3) / 1e12
        traffic, traffic_src = traffic_from_profile([dom])
        step_flops = sum(v["flops"] for v in summ.values()) / args.steps
        s3 = dom == "split3"
        # split3: the algorithm's matrix work is SIX bf16 x bf16 products per fp32 product -- `achieved` counts those against the
        # dense bf16 MFMA peak; the fp32-equivalent rate (2 P C_in C_out / time) and what the pipe executes (rows without a
        # neighbour are multiplied as zeros) stand beside it
        ach, peak = (6.0 * tflops, PEAK_BF16_MFMA_TFLOPS) if s3 else (tflops, PEAK_F32_MFMA_TFLOPS)
        out["roofline"] = {
            "kernel": ("spconv_fwd_split3_kernel<128> (256 x 128 register tiles, v_mfma_f32_16x16x32_bf16, 6 products per block)" if s3
                       else f"spconv_fwd_kernel, BN={dom[2:]} output-channel tile ({dom})"), "bound": "mfma",
            "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
            "achieved_note": ("algorithmic: 6 bf16 MFMA products per fp32 product x 2 P C_in C_out per launch / HIP-event time, against the "
                              "dense bf16 MFMA peak" if s3 else "algorithmic 2 P C_in C_out per launch / HIP-event time, against the fp32 MFMA peak"),
            "fp32_equivalent_tflops": tflops, "fp32_mfma_peak": PEAK_F32_MFMA_TFLOPS, "fp32_equivalent_over_fp32_mfma_peak": tflops / PEAK_F32_MFMA_TFLOPS,
            "all_rows_all_offsets_tflops": d["mfma_flops_timed"] / (d["ms"] * 1e-3) / 1e12 if s3 else None,
            "all_rows_all_offsets_note": ("6 products x every row of every 256-row tile x every offset / time: what the matrix pipe WOULD run "
                                          "without the kernel's block masks (16-row blocks that lack an offset are skipped: 30 % of them at stride "
                                          "8, 70 % at stride 4 under mask-sorted rows) -- an upper bound of the executed rate, not the rate") if s3 else None,
            "executed": None if not s3 else (lambda ex, gemm: {
                "tflops": ex, "dense_gemm_bf16_tflops": gemm, "over_dense_gemm": ex / gemm,
                "note": "6 products x the 16-row blocks the kernel multiplies (those that hold a neighbour under the offset) / time: the "
                        "rate the matrix pipe runs at in this kernel, next to what torch.matmul (hipBLASLt) reaches on a dense bf16 "
                        "[8192 x 8192] @ [8192 x 16384] product measured in this process -- the practical ceiling of the pipe on this "
                        "chip (its nominal peak is `peak`)"})(d.get("executed_flops_timed", 0.0) / (d["ms"] * 1e-3) / 1e12, dense_gemm_tflops(device)),
            "traffic": traffic, "launches": d["timed"], "avg_us": 1e3 * d["ms"] / max(1, d["timed"]),
            "traffic_unit": "GB per launch (HBM-side, rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE)",
            "traffic_source": None if traffic is None else f"static: {traffic_src} (tools/pmc_bench.sh over this bench "
                                                            "command; counters cannot be read inside the timed process)",
            "mfma_busy": mfma_busy_profile(),
            "step_frac": step_flops / (elapsed / args.steps) / 1e12 / PEAK_F32_MFMA_TFLOPS,
            "step_frac_note": "all convolution flops of a step (fp32-equivalent, 2 P C_in C_out) / wall time / the fp32 MFMA peak -- with "
                              "the dense levels on the bf16 pipe this is a rate, not a fraction of a roofline",
            "step_conv_gflop": step_flops / 1e9,
            "algorithmic_gflop_per_launch": d["flops"] / d["launches"] / 1e9,
            "algorithmic_gbytes_per_launch": d["bytes"] / d["launches"] / 1e9,
            "algorithmic_hbm_gbs": d["bytes_timed"] / (d["ms"] * 1e-3) / 1e9,
            "sampling": "every 3rd launch of the variant carried HIP events (`launches`); flops / bytes of exactly those launches",
            "serial": None if sprof is None else (lambda q: {
                "achieved": q["flops_timed"] / (q["ms"] * 1e-3) / 1e12, "frac": q["flops_timed"] / (q["ms"] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                "avg_us": 1e3 * q["ms"] / max(1, q["timed"]), "launches": q["timed"],
                "note": "the same launches in a pass of the same steps with every stream overlap off (one queue: nothing runs beside "
                        "the kernel); `achieved` above is from the timed pass, where the side streams' kernels share the chip"})(
                sprof.summary()[dom_variant]),
            "timed_variants": sorted(k for k, v in summ.items() if v["timed"]),
            "conv_ms_per_step_timed_variants": sum(v["ms"] * v["launches"] / max(1, v["timed"]) for v in summ.values() if v["timed"]) / args.steps,
            "variants": {k: {"launches": v["launches"], "timed": v["timed"], "ms": round(v["ms"], 3),
                             "tflops": v["flops_timed"] / (v["ms"] * 1e-3) / 1e12,
                             "alg_gbs": v["bytes_timed"] / (v["ms"] * 1e-3) / 1e9} for k, v in summ.items() if v["timed"]},
        }
    if vprof is not None or (prof is not None and args.all_variants):
        # every launch that is NOT the dominant 128-column tile kernel, split by what bounds it: arithmetic
        # intensity of the launch (algorithmic flops / algorithmic bytes) against the ridge point 157.3 TFLOP/s / 8 TB/s = 19.7
        # FLOP/B -- the 64 / 96-column kernel_size-3 layers and the wide row-kernel launches are MFMA-bound, the 32-channel
        # layers, the stems and the narrow row-kernel launches are HBM-bound; each class against ITS peak
        narrow_variants = ("bn96", "bn64", "bn32", "bn16", "rows", "thin") + (("bn128",) if ops.SPLIT3 else ())
        split = ops.launches_by_bound(vprof or prof, PEAK_F32_MFMA_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9), narrow_variants)
        tiny = split.pop("tiny")
        if tiny["launches"]:
            out["roofline_narrow_tiny"] = {
                "kernel": "launches with fewer than 32768 output rows (replicas included): the condition encoders (18000-point partial scan, "
                          "5800 voxels at stride 16) and the one-voxel unconditional branch -- at most one 128-row tile per compute unit",
                "bound": "launch latency (no roofline applies: the grid cannot fill the chip); queued on the side stream one step ahead",
                "ms_per_step": tiny["ms"] / args.steps, "launches_per_step": tiny["launches"] / args.steps,
                "avg_us": 1e3 * tiny["ms"] / tiny["launches"],
                "algorithmic_gflop_per_step": tiny["flops"] / args.steps / 1e9, "algorithmic_mbytes_per_step": tiny["bytes"] / args.steps / 1e6}
        for bound, d in split.items():
            if not d["launches"]:
                continue
            mfma = bound == "mfma"
            ach = d["flops"] / (d["ms"] * 1e-3) / 1e12 if mfma else d["bytes"] / (d["ms"] * 1e-3) / 1e9
            peak = PEAK_F32_MFMA_TFLOPS if mfma else PEAK_HBM_GBS
            vs = sorted({k[0] for k in d["shapes"]})
            layers = []
            for (variant, kk, c_in, c_out), r in sorted(d["shapes"].items(), key=lambda kv: -kv[1]["ms"]):
                a_ = r["flops"] / (r["ms"] * 1e-3) / 1e12 if mfma else r["bytes"] / (r["ms"] * 1e-3) / 1e9
                layers.append({"kernel": variant, "k": kk, "c_in": c_in, "c_out": c_out, "launches_per_step": r["launches"] / args.steps,
                               "ms_per_step": round(r["ms"] / args.steps, 4), "achieved": round(a_, 2), "frac": round(a_ / peak, 4),
                               "flop_per_byte": round(r["flops"] / r["bytes"], 1)})
            out["roofline_narrow_" + bound] = {
                "kernel": ("launches outside the 128-column tile kernel with arithmetic intensity >= 19.7 FLOP/B and >= 32768 output rows (bn64 / "
                           "bn96 kernel_size-3 tiles at stride 2-4, wide spconv_rows_kernel launches)" if mfma else
                           "launches outside the 128-column tile kernel with arithmetic intensity < 19.7 FLOP/B and >= 32768 output rows "
                           "(32-channel layers, stems, low-density stride-1/2 tail passes, narrow spconv_rows_kernel launches)"),
                "bound": bound, "achieved": ach, "peak": peak, "unit": "TFLOP/s" if mfma else "GB/s", "frac": ach / peak,
                "ms_per_step": d["ms"] / args.steps, "launches_per_step": d["launches"] / args.steps,
                "algorithmic_gflop_per_launch": d["flops"] / d["launches"] / 1e9,
                "algorithmic_gbytes_per_launch": d["bytes"] / d["launches"] / 1e9,
                "traffic": traffic_from_profile(vs)[0],
                "traffic_unit": "GB per launch averaged over ALL launches of the kernel variants " + "/".join(vs) + " (a variant serves "
                                "both classes; HBM-side, rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE; static: " + str(traffic_from_profile(vs)[1]) + ")",
                "note": "algorithmic flops 2 P C_in C_out and bytes 4 (M_in C_in + M_out C_out) + 4 K C_in C_out + 8 P per launch / "
                        "HIP-event time, from a second pass of the same steps with every conv launch timed (not the pass behind `value`)",
                "layers": layers}
    if alt is not None:
        a_elapsed, aprof = alt
        out["native_fp32"] = {
            "dtype": "f32: every convolution on the native fp32 matrix instruction (v_mfma_f32_16x16x4_f32), ops.split3(False) / LIDIFF_SPLIT3=0",
            "value": args.steps / a_elapsed, "unit": "steps/s", "ms_per_step": 1e3 * a_elapsed / args.steps,
            "note": "the same K steps, same process, right after the timed pass; `value` above is the default configuration"}
        if aprof is not None:
            d = aprof.summary().get("bn128")
            if d and d["ms"] > 0:
                tf = d["flops_timed"] / (d["ms"] * 1e-3) / 1e12
                out["native_fp32"]["roofline"] = {
                    "kernel": "spconv_fwd_kernel, BN=128 output-channel tile (bn128)", "bound": "mfma", "achieved": tf,
                    "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_F32_MFMA_TFLOPS, "launches": d["timed"],
                    "avg_us": 1e3 * d["ms"] / max(1, d["timed"])}
    if alt16 is not None:
        h_elapsed, hprof = alt16
        out["f16x2"] = {
            "dtype": DTYPE_F16X2,
            "value": args.steps / h_elapsed, "unit": "steps/s", "ms_per_step": 1e3 * h_elapsed / args.steps,
            "note": "the same K steps, same process; informational, never `value`"}
        if hprof is not None:
            d = hprof.summary().get("split3")
            if d and d["ms"] > 0:
                tf = d["flops_timed"] / (d["ms"] * 1e-3) / 1e12
                out["f16x2"]["roofline"] = {
                    "kernel": "spconv_fwd_split3_kernel<128, 2> (3 products per block on v_mfma_f32_16x16x32_f16)", "bound": "mfma",
                    "achieved": 3.0 * tf, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": 3.0 * tf / PEAK_BF16_MFMA_TFLOPS,
                    "fp32_equivalent_tflops": tf, "executed_tflops": d.get("executed_flops_timed", 0.0) / 2.0 / (d["ms"] * 1e-3) / 1e12,
                    "launches": d["timed"], "avg_us": 1e3 * d["ms"] / max(1, d["timed"])}
    if world == 1 and not args.no_coords_roofline:
        with torch.no_grad():
            out["roofline_hbm"] = coords_roofline(scan_np, device)
    if world == 1 and not args.no_closed_loop:
        # beside the metric, never `value` (SURVEY.md 8d "also report the closed-loop run with seeded random weights"): one whole
        # scan through DiffCompletion.complete_scan -- the number the reference itself prints per scan (pipeline:198-203)
        with torch.no_grad():
            per_scan, phases, rows = pipeline_leg(pipe, device, [5000])
        out["closed_loop"] = {
            "s_per_scan": per_scan[0], "scans_per_s": 1.0 / per_scan[0], "denoising_steps_per_s": T_STEPS / phases["denoise_s"],
            "ms_per_denoising_step": 1e3 * phases["denoise_s"] / T_STEPS, "phases_s": phases,
            "points_diffused": rows[0][0], "points_refined": rows[0][1], "finite": rows[0][2],
            "workload": "complete_scan (pipeline:117-132) on the bundled scan: range filter + FPS 119035 -> 18000, x10 + N(0, I), CLOSED "
                        "T=50 CFG loop (each step voxelises the points the previous one produced), post-filter, MinkUNet refinement; "
                        "seeded random-init weights -- the offsets do not contract as with trained weights (they grow to ~1 / alpha_T), "
                        "so the maps are sparser than on the metric's sigma_t trajectory; parity of this loop against the oracle: "
                        "tests/test_gpu_baseline.py::test_closed_loop_c2_chamfer_vs_oracle"}
    if world == 1 and not args.no_train:
        del pipe
        torch.cuda.empty_cache()
        out["train"] = train_leg(scan_np, device, steps=10, warmup=2)
        # config_refine.yaml's batch (8 items, ~110 GiB at this size) when the device has the room; else its per-item size at B = 2
        free_gib = torch.cuda.mem_get_info(device)[0] / 2 ** 30
        out["train_refine"] = train_refine_leg(scan_np, device, items=8 if free_gib > 160 else 2)
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(scan_np, threads=args.cpu_threads)
    emit(json.dumps(out))


if __name__ == "__main__":
    main()
