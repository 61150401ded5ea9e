#!/usr/bin/env python
"""Times the diffusion training step (models.py:180-217: forward + loss + backward + Adam) on the BASELINE
config-5 shape per GPU: B = 2 scans of 180 000 points, 18 000-point partial scans, fp32 or (--precision bf16) bf16 conv operands,
random-init weights.

    python tools/train_probe.py [--steps 5] [--batch 2] [--points 180000]
Prints ms per phase (HIP events on the current stream) and the peak allocated memory.
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--points", type=int, default=180000)
    ap.add_argument("--precision", default="32", choices=["32", "bf16"])
    a = ap.parse_args()
    from lidiff_amd.diffusion import DiffusionPoints
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    module = DiffusionPoints(device=dev, precision=a.precision)
    module.train()
    opt, _ = module.configure_optimizers()       # the LR schedule steps per epoch (train_loop), not here
    scan = np.load(os.path.join(ROOT, "tests", "golden", "scan_000123_fps18000.npy")).astype(np.float32)
    rng = np.random.default_rng(0)
    rep = a.points // scan.shape[0]
    batches = []
    for _ in range(2):
        part = np.stack([scan + 0.01 * rng.standard_normal(scan.shape).astype(np.float32) for _ in range(a.batch)])
        full = np.tile(part, (1, rep, 1)) + 0.05 * rng.standard_normal((a.batch, rep * scan.shape[0], 3)).astype(np.float32)
        batches.append({"pcd_full": torch.from_numpy(full), "pcd_part": torch.from_numpy(part)})
    gen = torch.Generator(device=dev).manual_seed(1)
    ev = lambda: torch.cuda.Event(enable_timing=True)
    tot = {"forward+loss": 0.0, "backward": 0.0, "optimizer": 0.0}
    torch.cuda.reset_peak_memory_stats()
    for step in range(a.warmup + a.steps):
        e0, e1, e2, e3 = ev(), ev(), ev(), ev()
        e0.record()
        loss = module.training_step(batches[step % 2], step, generator=gen)
        e1.record()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        e2.record()
        opt.step()
        e3.record()
        torch.cuda.synchronize()
        if step >= a.warmup:
            tot["forward+loss"] += e0.elapsed_time(e1)
            tot["backward"] += e1.elapsed_time(e2)
            tot["optimizer"] += e2.elapsed_time(e3)
        print(f"step {step}: loss {float(loss):.4f}", flush=True)
    ms = {k: v / a.steps for k, v in tot.items()}
    total = sum(ms.values())
    print(f"precision {a.precision}  B={a.batch} x {a.points} points: " + "  ".join(f"{k} {v:.1f} ms" for k, v in ms.items()) +
          f"  | step {total:.1f} ms = {1e3 / total:.2f} steps/s, {a.batch * 1e3 / total:.2f} scans/s; "
          f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")


if __name__ == "__main__":
    main()
