#!/usr/bin/env python
"""Per-phase cycle breakdown of the sparse-conv slab loop (debug build liblidiff_amd_timing.so, built
with -DLIDIFF_CONV_TIMING): each wave accumulates cycle-counter deltas around the counted vmcnt wait, the
LDS stores, the prefetch issue, the MFMA half, the flush and the barrier.  Diagnostic only.

    python tools/conv_timing.py --level 3 --cin 256 --cout 256 --kind k3
"""
import argparse
import ctypes
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "lidiff_amd", "csrc")
TIMING_LIB = os.path.join(CSRC, "liblidiff_amd_timing.so")


def build_timing_lib():
    if os.path.exists(TIMING_LIB) and os.path.getmtime(TIMING_LIB) > os.path.getmtime(os.path.join(CSRC, "spconv.hip")):
        return
    objs = []
    for src, flags in (("spconv.hip", ["-DLIDIFF_CONV_TIMING"]), ("coords.hip", [])):
        obj = os.path.join("/tmp", src + ".timing.o")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", *flags, "-c",
                        os.path.join(CSRC, src), "-o", obj], check=True)
        objs.append(obj)
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", TIMING_LIB, *objs], check=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sigma", type=float, default=1.0)
    ap.add_argument("--level", type=int, default=3)
    ap.add_argument("--cin", type=int, default=256)
    ap.add_argument("--cout", type=int, default=256)
    ap.add_argument("--kind", default="k3")
    a = ap.parse_args()
    build_timing_lib()
    from lidiff_amd import _lib
    _lib.LIB_PATH = TIMING_LIB
    import lidiff_amd.MinkowskiEngine as ME
    from lidiff_amd import ops
    lib = _lib.load()
    lib.lidiff_debug_set_conv_timing_buffer.argtypes = [ctypes.c_void_p]
    dev = torch.device("cuda:0")
    scan = np.load(os.path.join(ROOT, "tests", "golden", "scan_000123_fps18000.npy"))
    rng = np.random.default_rng(0)
    pts = np.tile(scan, (10, 1)) + a.sigma * rng.standard_normal((180000, 3)).astype(np.float32)
    feats = torch.from_numpy(pts.astype(np.float32)).to(dev)
    coord = torch.cat([torch.zeros(180000, 1, device=dev), torch.round(feats / 0.05)], 1)
    field = ME.TensorField(features=feats, coordinates=coord, device=dev)
    field.sparse()
    mgr = field.coordinate_manager
    ts = 1
    for _ in range(4):
        ts = mgr.stride(ts, 2)
    ts = 1 << a.level
    m_in = mgr.maps[ts].coords.shape[0]
    nbr = mgr.kernel_map(ts, ts, 3) if a.kind == "k3" else None
    m_out = m_in
    k = 27 if nbr is not None else 1
    x = torch.randn(m_in, a.cin, device=dev)
    w = torch.randn(k, a.cin, a.cout, device=dev) * 0.05
    for _ in range(2):
        ops.spconv_fwd(x, w, nbr, m_out)
    bn = 128 if a.cout % 128 == 0 else 96 if a.cout % 96 == 0 else 64 if a.cout % 64 == 0 else 32
    nwg = ((m_out + 127) // 128 + 7) // 8 * 8 * (a.cout // bn)
    dbg = torch.zeros(nwg * 8 * 9, dtype=torch.int64, device=dev)
    lib.lidiff_debug_set_conv_timing_buffer(dbg.data_ptr())
    ops.spconv_fwd(x, w, nbr, m_out)
    torch.cuda.synchronize()
    lib.lidiff_debug_set_conv_timing_buffer(None)
    d = dbg.cpu().numpy().reshape(nwg, 8, 9).astype(np.float64)
    live = d[:, :, 8] > 0
    names = ["vmcnt wait", "lds store", "prefetch issue", "mfma half", "flush", "barrier", "stage total", "loop total"]
    print(f"{a.kind} {a.cin}->{a.cout} level {a.level}: workgroups {int(live.any(1).sum())}, "
          f"mean stages/wave {d[:, :, 8][live].mean():.1f}")
    for grp, sel in (("waves 0-3 (io first)", slice(0, 4)), ("waves 4-7 (mfma first)", slice(4, 8))):
        dd, ll = d[:, sel, :], live[:, sel]
        print(grp)
        for q, n in enumerate(names):
            print(f"   {n:15s} {(dd[:, :, q][ll] / dd[:, :, 8][ll]).mean():9.0f} cycles/stage")


if __name__ == "__main__":
    main()
