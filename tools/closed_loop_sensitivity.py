#!/usr/bin/env python
"""How sensitive is the CLOSED T = 50 loop of BASELINE configs[1] (180 000 points, seeded random weights) to rounding-level
perturbations?  Runs DiffCompletion.completion_loop on the device twice with the same scheduler noise -- once from x_T, once from
x_T + 1e-6 m * N(0, I) -- and reports the reference's Chamfer distance between the two completions and the share of points that
moved by more than 0.1 mm / 1 mm.  Context for tests/test_gpu_baseline.py::test_closed_loop_c2_chamfer_vs_oracle, where the CPU
oracle and the device differ by fp32 rounding and by ~5 ppm of differently rounded voxel coordinates per step.
    python tools/closed_loop_sensitivity.py [--eps 1e-6]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--eps", type=float, default=1e-6)
    a = ap.parse_args()
    import heavy_oracle as heavy
    from conftest import build_seeded_models
    from lidiff_amd.evaluation import ChamferDistance
    from lidiff_amd.pipeline import DiffCompletion
    dev = torch.device("cuda:0")
    fps = np.load(os.path.join(ROOT, "tests", "golden", "scan_000123_fps18000.npy"))
    enc, unet, refine = build_seeded_models(42)
    pipe = DiffCompletion(denoising_steps=heavy.CLOSED_STEPS, cond_weight=6.0, device=dev)
    pipe.partial_enc, pipe.model, pipe.model_refine = enc.to(dev), unet.to(dev), refine.to(dev)
    scan_np, noisy_np = heavy.closed_inputs(fps)
    n = scan_np.shape[0]
    scan = torch.from_numpy(scan_np).double()[None].to(dev)
    noises = [torch.from_numpy(heavy.closed_noise(i, n)).to(dev) for i in range(heavy.CLOSED_STEPS)]
    rng = np.random.default_rng(99)
    outs = []
    for eps in (0.0, a.eps):
        x0 = noisy_np.astype(np.float64) + eps * rng.standard_normal(noisy_np.shape)
        pipe.new_scheduler()
        outs.append(pipe.completion_loop(scan, pipe.points_to_tensor(torch.from_numpy(x0)[None].to(dev)), pipe.points_to_tensor(scan),
                                         pipe.points_to_tensor(torch.zeros_like(scan)), noises=noises))
    err = np.abs(outs[0] - outs[1]).max(axis=1)
    cd = ChamferDistance(device=dev)
    cd.update(outs[0], outs[1])
    print(f"closed loop T = {heavy.CLOSED_STEPS}, {n} points, x_T perturbed by {a.eps:g} m: Chamfer {cd.compute()[0]:.3e} m; points moved "
          f"> 0.1 mm {np.mean(err > 1e-4):.5f}, > 1 mm {np.mean(err > 1e-3):.5f}, > 1 cm {np.mean(err > 1e-2):.5f}; median {np.median(err):.2e} m, "
          f"max {err.max():.2e} m; offsets: std {np.std(outs[0] - scan_np):.2f} m")
