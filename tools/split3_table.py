#!/usr/bin/env python
"""profiles/rNN_split3.txt: the split-operand convolution (spconv_split3.hip) next to the native fp32 kernel on the bench scan's own
maps -- time (CFG pair stacked), fp32-equivalent TFLOP/s, worst error against a float64 reference (torch, on the device, on a
sample of output rows) for both kernels, and the fraction of (16-row block, offset) pairs the kernel executes in table order and with
the rows sorted by their neighbour sets.     python tools/split3_table.py [--sigmas 1.0,0.3,0.05]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / iters


def block_fraction(nbr, blk=16, tile=256):
    """executed (block, offset) pairs / all pairs of the tiles' active offsets == all pairs (an offset absent from a whole tile
    costs nothing either way, so the denominator is every block x every offset)"""
    k, m = nbr.shape
    mp = (m // blk) * blk
    present = (nbr[:, :mp] >= 0).view(k, -1, blk).any(2)
    return present.float().mean().item()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sigmas", default="1.0,0.3,0.05")
    ap.add_argument("--f16x2", action="store_true", help="also the opt-in two-piece fp16 mode (ops.split_pieces(2)): time and error columns")
    args = ap.parse_args()
    from lidiff_amd import ops
    import lidiff_amd.MinkowskiEngine as ME
    dev = torch.device("cuda:0")
    scan = np.load(os.path.join(ROOT, "tests", "golden", "scan_000123_fps18000.npy"))
    layers = [(3, 256, 256), (3, 384, 256), (4, 256, 256), (3, 128, 128), (4, 128, 256), (2, 128, 128), (2, 192, 128), (2, 64, 64), (2, 32, 64)]
    print("# split-operand kernel vs native fp32 kernel, bench scan, CFG pair stacked (replicas 2); errors vs a float64 reference on 4096 output rows")
    print("# sigma level cin->cout | rows pairs occupancy | executed block fraction: table order / sorted | native us TF | split3 table-order us | "
          "split3 sorted us TF-equivalent | speed-up | max|err| native / split3 (outputs up to)" + (" | f16x2 sorted us TF-equivalent | max|err|" if args.f16x2 else ""))
    for sigma in [float(v) for v in args.sigmas.split(",")]:
        rng = np.random.default_rng(0)
        pts = np.tile(scan, (10, 1)) + sigma * rng.standard_normal((180000, 3)).astype(np.float32)
        feats = torch.from_numpy(pts.astype(np.float32)).to(dev)
        coord = torch.cat([torch.zeros(180000, 1, device=dev), torch.round(feats / 0.05)], 1)
        field = ME.TensorField(features=feats, coordinates=coord, device=dev)
        field.sparse()
        mgr = field.coordinate_manager
        ts = 1
        for _ in range(4):
            ts = mgr.stride(ts, 2)
        for level, cin, cout in layers:
            ts = 1 << level
            nbr = mgr.kernel_map(ts, ts, 3)
            m = nbr.shape[1]
            pairs = int((nbr >= 0).sum())
            nbr_s, order = ops.mask_sorted_map(nbr)
            g = torch.Generator(device="cpu").manual_seed(level * 1000 + cin + cout)
            x = (torch.randn(2 * m, cin, generator=g) * (torch.rand(cin, generator=g) * 2 + 0.1)).to(dev)
            w = (torch.randn(27, cin, cout, generator=g) / np.sqrt(cin * 9)).to(dev)
            x3 = ops.split3_rows(x)
            hint = mgr.is_sparse_map(ts, ts, 3, c_out=cout)
            t_nat = timed(lambda: ops.spconv_fwd(x, w, nbr, m, sparse_map=hint, replicas=2))
            t_tab = timed(lambda: ops.spconv_fwd_split3(x3, w, nbr, m, replicas=2))
            t_srt = timed(lambda: ops.spconv_fwd_split3(x3, w, nbr_s, m, replicas=2, row_order=order))
            out_n = ops.spconv_fwd(x, w, nbr, m, sparse_map=hint, replicas=2)
            out_s = ops.spconv_fwd_split3(x3, w, nbr_s, m, replicas=2, row_order=order)
            rows = torch.from_numpy(np.random.default_rng(1).choice(m, min(m, 4096), replace=False)).to(dev)
            ref = torch.zeros(rows.numel(), cout, dtype=torch.float64, device=dev)
            for k in range(27):
                idx = nbr[k, rows].long()
                ok = idx >= 0
                ref[ok] += x[:m][idx[ok]].double() @ w[k].double()
            en = (out_n[:m][rows].double() - ref).abs().max().item()
            es = (out_s[:m][rows].double() - ref).abs().max().item()
            fl = 2.0 * 2 * pairs * cin * cout
            extra = ""
            if args.f16x2:
                x2 = ops.split3_rows(x, 2)
                t_h = timed(lambda: ops.spconv_fwd_split3(x2, w, nbr_s, m, replicas=2, row_order=order, pieces=2))
                out_h = ops.spconv_fwd_split3(x2, w, nbr_s, m, replicas=2, row_order=order, pieces=2)
                eh = (out_h[:m][rows].double() - ref).abs().max().item()
                ops.split_check()
                extra = f" | {t_h:7.1f} {fl / t_h / 1e6:6.1f} | {eh:.2e}"
            print(f"{sigma:<4} {level} {cin:>3}->{cout:<3} | {m:>6} {pairs:>8} {pairs / (27.0 * m):.3f} | {block_fraction(nbr):.3f} / {block_fraction(nbr_s):.3f} | "
                  f"{t_nat:7.1f} {fl / t_nat / 1e6:6.1f} | {t_tab:7.1f} | {t_srt:7.1f} {fl / t_srt / 1e6:6.1f} | {t_nat / t_srt:4.2f}x | "
                  f"{en:.2e} / {es:.2e} ({ref.abs().max().item():.1f})" + extra, flush=True)


if __name__ == "__main__":
    main()
