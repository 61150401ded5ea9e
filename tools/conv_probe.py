#!/usr/bin/env python
"""Micro-benchmark of lidiff_spconv_fwd on the real sparsity of the bench workload: builds the
coordinate maps of the 180k-point scan at a given sigma, then times one conv shape on one level.

    python tools/conv_probe.py --sigma 1.0 --level 3 --cin 256 --cout 256 [--iters 20] [--kind k3|down|up|k1]
Prints pairs, avg us, TFLOP/s (algorithmic: 2*P*Cin*Cout) for the level (0 = stride 1 ... 4 = stride 16).
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
    ap.add_argument("--sigma", type=float, default=1.0)
    ap.add_argument("--level", type=int, default=3)
    ap.add_argument("--cin", type=int, default=256)
    ap.add_argument("--cout", type=int, default=256)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--kind", default="k3")
    ap.add_argument("--sweep", action="store_true", help="all levels x the network's channel pairs")
    ap.add_argument("--ordered", action="store_true", help="Morton-ordered tiles (row_order + permuted table)")
    ap.add_argument("--lpt", action="store_true", help="128-row tiles in descending pair-count order (row_order + permuted table)")
    ap.add_argument("--sparse-hint", type=int, default=-1, help="force the sparse-map hint (0/1); default: the manager's rule")
    ap.add_argument("--kernel", default="tile", choices=["tile", "bf16", "split3"], help="kernel of the dense 128-column layers")
    ap.add_argument("--pieces", type=int, default=3, choices=[2, 3], help="--kernel split3: 3 bf16 pieces (default) or 2 fp16 pieces per operand")
    ap.add_argument("--rows16", action="store_true", help="--kernel bf16: gather bf16 shadow rows (ops.cast_bf16) instead of fp32 rows")
    ap.add_argument("--planes", type=int, default=1, help="--kernel bf16: bf16 pieces per operand (1 = rounded, 2 / 3 = split)")
    ap.add_argument("--centre-tail", action="store_true", help="k3 layers as centre pass + tail rows (ops.spconv_centre_tail)")
    ap.add_argument("--up-ordered", action="store_true", help="'up' layers with their output rows grouped by offset (CoordinateManager.up_order)")
    ap.add_argument("--replicas", type=int, default=1, help="stacked feature matrices per launch (the bench runs the CFG pair: 2)")
    ap.add_argument("--flags", type=int, default=0, help="extra lidiff_spconv_fwd flag bits (8 = LIDIFF_CONV_SKEW)")
    ap.add_argument("--cases", default="", help="several cases in ONE process (maps built once): 'level,cin,cout,kind,hint,flags;...' "
                                                "(hint -1 = the manager's rule)")
    ap.add_argument("--timeline", action="store_true",
                    help="diagnostic build: per-workgroup cycle counters (prologue / main loop / epilogue / barrier / flush)")
    ap.add_argument("--probe", type=int, default=None,
                    help="diagnostic build (-DLIDIFF_CONV_PROBE): bit 0 = skip the A gather, 1 = skip the W loads, 2 = no barrier, 4 = no flush")
    args = ap.parse_args()
    if args.timeline and args.probe is None:
        args.probe = 0
    if args.probe is not None:
        import ctypes
        import subprocess
        from lidiff_amd import _lib
        csrc = os.path.join(ROOT, "lidiff_amd", "csrc")
        lib = os.path.join(csrc, "liblidiff_amd_probe.so")
        from lidiff_amd.csrc import build as _build
        srcs = [os.path.join(csrc, f) for f in _build.SOURCES]
        if not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(f) for f in srcs + [os.path.join(csrc, "spconv.h")]):
            subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-DLIDIFF_CONV_PROBE",
                            "-shared"] + srcs + ["-o", lib], check=True)
        _lib.LIB_PATH = lib
        _lib.load().lidiff_debug_set_conv_probe(ctypes.c_int(args.probe))
    from lidiff_amd import ops
    import lidiff_amd.MinkowskiEngine as ME
    ops.CONV_FLAGS = args.flags
    dev = torch.device("cuda:0")
    scan = np.load(os.path.join(ROOT, "tests", "golden", "scan_000123_fps18000.npy"))
    rng = np.random.default_rng(0)
    pts = np.tile(scan, (10, 1)) + args.sigma * rng.standard_normal((180000, 3)).astype(np.float32)
    feats = torch.from_numpy(pts.astype(np.float32)).to(dev)
    coord = torch.cat([torch.zeros(180000, 1, device=dev), torch.round(feats / 0.05)], 1)
    field = ME.TensorField(features=feats, coordinates=coord, device=dev)
    field.sparse()
    mgr = field.coordinate_manager
    ts = 1
    for _ in range(4):
        ts = mgr.stride(ts, 2)

    def run(level, cin, cout, kind):
        ts = 1 << level
        if kind == "k3":
            nbr, m_in = mgr.kernel_map(ts, ts, 3), mgr.maps[ts].coords.shape[0]
        elif kind == "down":
            nbr, m_in = mgr.kernel_map(ts, ts * 2, 2), mgr.maps[ts].coords.shape[0]
        elif kind == "up":
            nbr, m_in = mgr.kernel_map(ts * 2, ts, 2, True), mgr.maps[ts * 2].coords.shape[0]
        else:
            nbr, m_in = None, mgr.maps[ts].coords.shape[0]
        m_out = nbr.shape[1] if nbr is not None else m_in
        k = nbr.shape[0] if nbr is not None else 1
        pairs = int((nbr >= 0).sum()) if nbr is not None else m_in
        x = torch.randn(args.replicas * m_in, cin, device=dev)
        w = torch.randn(k, cin, cout, device=dev) * 0.05
        order = None
        if args.ordered and nbr is not None:
            order = ops.tile_order(mgr.maps[ts if kind != "down" else ts * 2].coords, ts if kind != "down" else ts * 2)
            nbr = nbr.index_select(1, order.long()).contiguous()
        if args.lpt and nbr is not None:
            m = nbr.shape[1]
            tiles = (m + 127) // 128
            cnt = torch.zeros(tiles * 128, dtype=torch.int32, device=dev)
            cnt[:m] = (nbr >= 0).sum(0).to(torch.int32)
            per_tile = cnt.view(tiles, 128).sum(1)
            last = per_tile[-1].clone()
            per_tile[-1] = -1                                   # the ragged last tile stays last
            tord = torch.argsort(per_tile, descending=True, stable=True)
            rows = (tord[:, None] * 128 + torch.arange(128, device=dev)[None, :]).reshape(-1)
            order = rows[rows < m].to(torch.int32)
            nbr = nbr.index_select(1, order.long()).contiguous()
        hint = {"k3": mgr.is_sparse_map(ts, ts, 3, c_out=cout), "down": mgr.is_sparse_map(ts, ts * 2, 2, c_out=cout),
                "up": mgr.is_sparse_map(ts * 2, ts, 2, True, c_out=cout), "k1": False}[kind]
        if args.sparse_hint >= 0:
            hint = bool(args.sparse_hint)
        if args.up_ordered and kind == "up":
            (nbr, order), hint = mgr.up_order(ts * 2, ts), False
        plist = None
        if args.up_ordered and kind == "up" and ops.pairs_kernel_applies(cin, 0, cout):
            plist = mgr.up_pairs(ts * 2, ts)
        if plist is not None:
            conv = lambda: ops.spconv_fwd_pairs(x, w, plist[0], plist[1], plist[2], m_out, replicas=args.replicas)
        elif args.centre_tail and kind == "k3":
            tmap = ops.TailMap(nbr)
            conv = lambda: ops.spconv_centre_tail(x, w, tmap, m_out, replicas=args.replicas, sparse_map=hint)
        elif args.kernel == "split3" and ops.split3_conv_applies(cin, 0, cout):
            x3 = ops.split3_rows(x, args.pieces)
            nbr_s, order_s = (ops.mask_sorted_map(nbr) if args.flags & 1 else (nbr, None))
            conv = lambda: ops.spconv_fwd_split3(x3, w, nbr_s, m_out, replicas=args.replicas, row_order=order_s, pieces=args.pieces)
        elif args.kernel == "bf16" and ops.bf16_conv_applies(cin, 0, cout):
            xin = ops.cast_bf16(x) if args.rows16 else x
            conv = lambda: ops.spconv_fwd_bf16(xin, w, nbr, m_out, planes=args.planes, replicas=args.replicas,
                                               kernel={0: None, 2: "ring", 3: "two_stage", 4: "wide"}.get(args.flags))
        else:
            conv = lambda: ops.spconv_fwd(x, w, nbr, m_out, sparse_map=hint, row_order=order, replicas=args.replicas)
        for _ in range(3):
            conv()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(args.iters):
            conv()
        e.record()
        torch.cuda.synchronize()
        us = 1e3 * s.elapsed_time(e) / args.iters
        tf = 2.0 * args.replicas * pairs * cin * cout / (us * 1e-6) / 1e12
        print(f"sigma={args.sigma} level={level} kind={kind} {cin}->{cout} m_in={m_in} m_out={m_out} pairs={pairs} "
              f"nbrs/row={pairs / m_out:.2f} hint={int(hint)} reps={args.replicas} flags={args.flags} avg_us={us:.1f} TFLOP/s={tf:.2f}", flush=True)
        if args.timeline:
            import ctypes
            from lidiff_amd import _lib
            tl = torch.zeros(1 << 15, 2, 10, dtype=torch.int64, device=dev)
            _lib.load().lidiff_debug_set_conv_timeline(ctypes.c_void_p(tl.data_ptr()))
            ops.spconv_fwd(x, w, nbr, m_out, sparse_map=hint, row_order=order, replicas=args.replicas)
            torch.cuda.synchronize()
            _lib.load().lidiff_debug_set_conv_timeline(ctypes.c_void_p(0))
            raw = tl.cpu().numpy()
            live = raw[:, 0, 5] > 0
            q = raw[live, 0].astype(np.float64)
            mhz = (q[:, 0] + q[:, 1] + q[:, 2]) / (q[:, 7] / 100.0)      # s_memtime ticks per us of the 100 MHz wall clock
            print(f"  shader clock while the kernel runs (s_memtime / s_memrealtime): mean {mhz.mean():.0f} MHz, "
                  f"min {mhz.min():.0f}, max {mhz.max():.0f}; mean workgroup duration {q[:, 7].mean() / 100:.1f} us")
            t = raw.astype(np.float64)
            t = t[t[:, 0, 5] > 0]
            stages = (t[:, 0, 5] * t[:, 0, 6]).mean()
            print(f"  timeline (s_memtime ticks; the two waves of SIMD 0): workgroups {len(t)}, items {t[:, 0, 5].mean():.1f} x "
                  f"slabs {t[:, 0, 6].mean():.0f} = {stages:.0f} stages")
            for wv, name in ((0, "wave 0"), (1, f"wave NW/2")):
                q = t[:, wv]
                print(f"  {name}: prologue {q[:, 0].mean():.0f}  main loop {q[:, 1].mean():.0f}  epilogue {q[:, 2].mean():.0f} | per stage: "
                      f"loop {q[:, 1].mean() / stages:.0f} = issue {q[:, 8].mean() / stages:.0f} + mma {q[:, 9].mean() / stages:.0f} + "
                      f"flush {q[:, 4].mean() / stages:.0f} + barrier {q[:, 3].mean() / stages:.0f} + rest "
                      f"{(q[:, 1] - q[:, 8] - q[:, 9] - q[:, 4] - q[:, 3]).mean() / stages:.0f}")

    if args.cases:
        for case in args.cases.split(";"):
            lv, ci, co, kind, hint, flags = case.split(",")
            args.sparse_hint, args.flags = int(hint), int(flags)
            ops.CONV_FLAGS = args.flags
            run(int(lv), int(ci), int(co), kind)
    elif args.sweep:
        shapes = [(0, 32, 32), (1, 32, 32), (1, 32, 64), (2, 64, 64), (2, 64, 128), (3, 128, 128), (3, 128, 256),
                  (3, 256, 256), (3, 384, 256), (4, 256, 256), (2, 192, 128), (2, 128, 128), (1, 128, 96), (1, 96, 96),
                  (0, 128, 96), (0, 96, 96)]
        for lv, ci, co in shapes:
            run(lv, ci, co, "k3")
        run(3, 256, 256, "up"); run(2, 128, 128, "down"); run(3, 384, 256, "k1")
    else:
        run(args.level, args.cin, args.cout, args.kind)


if __name__ == "__main__":
    main()
