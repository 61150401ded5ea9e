#!/usr/bin/env python
"""CPU-only calibration of the bars in tests/test_gpu_bf16_blocks.py: the oracle's bf16-operand emulation of every block run
twice, with float32 and with float64 sums.  The two runs differ by fp32 summation noise exactly as the device and the float64
oracle do, so the statistics printed here (share of elements within 1e-4, worst error over the tensor's scale, cosine) are what
a correct device kernel is expected to show -- including the few elements whose bf16 rounding flips between the two runs."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import test_gpu_bf16_blocks as tb  # noqa: E402
from lidiff_amd import minkunet as product  # noqa: E402
from oracle import me_cpu as me  # noqa: E402
from oracle import minkunet_cpu as net  # noqa: E402

if __name__ == "__main__":
    fps = np.load(os.path.join(ROOT, "tests", "golden", "scan_000123_fps18000.npy"))
    pts = tb.scene_points(fps)
    x0 = net.points_to_field(torch.from_numpy(pts)[None]).sparse()
    ts = 1
    for _ in range(4):
        ts = x0.mgr.stride(ts, 2)
    cmgr = x0.mgr
    m = lambda t: cmgr.maps[t].shape[0]
    print("voxels per level:", [m(1 << l) for l in range(5)])
    only = sys.argv[1:]
    for name, kind, level, cin, cout, cskip in tb.BLOCKS:
        if only and name not in only:
            continue
        ts_in = 1 << level
        ts_out = ts_in * 2 if kind == "stage" else ts_in // 2
        torch.manual_seed(1000 + level)
        block = product._stage(cin, cout, 3) if kind == "stage" else product._up(cin, cout, cskip, 3)
        for mod in block.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.weight.data.uniform_(0.8, 1.2)
                mod.bias.data.normal_(0, 0.1)
        sd = {"b." + k: v.detach().clone() for k, v in block.state_dict().items()}
        x, skip, cot = tb.block_inputs(name, m(ts_in), m(ts_out) if cskip else 0, m(ts_out), cin, cout, cskip)
        x_cpu = me.CpuSparseTensor(x, ts_in, cmgr)
        skip_cpu = me.CpuSparseTensor(skip, ts_out, cmgr) if cskip else None
        o64 = tb.oracle_block(sd, kind, x_cpu, skip_cpu, x, skip, cot, torch.float64)
        o32 = tb.oracle_block(sd, kind, x_cpu, skip_cpu, x, skip, cot, torch.float32)
        print(name, tb.compare("first conv", tb.first_conv(sd, kind, x_cpu, x, torch.float32), tb.first_conv(sd, kind, x_cpu, x)))
        print(name, tb.compare("block output", o32[0], o64[0]))
        print(name, tb.compare("dX", o32[1], o64[1]))
        worst = min((tb.compare("dW " + k, o32[2][k], o64[2][k]) for k in o64[2] if o64[2][k] is not None),
                    key=lambda s: s["cosine"])
        print(name, "worst dW:", worst, flush=True)
