#!/usr/bin/env python
"""Which torch operators the training step's "torch elementwise / reductions / copies" class comes from: torch.profiler over two
bf16 steps, device time by aten operator and input shapes.   python tools/debug/train_ops.py [--precision bf16]"""
import argparse
import os
import sys

import numpy as np
import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="bf16")
    a = ap.parse_args()
    from lidiff_amd.diffusion import DiffusionPoints
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    module = DiffusionPoints(device=dev, precision=a.precision)
    module.train()
    opt, _ = module.configure_optimizers()
    scan = np.load(os.path.join(ROOT, "tests", "golden", "scan_000123_fps18000.npy")).astype(np.float32)
    rng = np.random.default_rng(0)
    part = np.stack([scan + 0.01 * rng.standard_normal(scan.shape).astype(np.float32) for _ in range(2)])
    full = np.tile(part, (1, 10, 1)) + 0.05 * rng.standard_normal((2, 180000, 3)).astype(np.float32)
    batch = {"pcd_full": torch.from_numpy(full), "pcd_part": torch.from_numpy(part)}
    gen = torch.Generator(device=dev).manual_seed(1)

    def step(i):
        loss = module.training_step(batch, i, generator=gen)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()

    for i in range(2):
        step(i)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        for i in range(2):
            step(2 + i)
        torch.cuda.synchronize()
    rows = []
    for e in prof.key_averages(group_by_input_shape=True):
        if not e.key.startswith("aten::"):
            continue
        dt = getattr(e, "device_time_total", None)          # inclusive: the kernels an operator launched
        if dt is None:
            dt = e.cuda_time_total
        if dt > 0:
            rows.append((dt / 2e3, e.count // 2, e.key, str(e.input_shapes)[:110]))
    rows.sort(reverse=True)
    print("ms/step  calls/step  operator  input shapes")
    for ms, n, key, shapes in rows[:70]:
        print(f"{ms:8.3f} {n:5d}  {key[:44]:44s} {shapes}")
    print(f"total self device time {sum(r[0] for r in rows):.1f} ms/step")


if __name__ == "__main__":
    main()
