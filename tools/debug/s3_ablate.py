#!/usr/bin/env python
"""Ablations of spconv_fwd_split3_kernel (probe build; LIDIFF_S3_ABLATE bits: 1 no requests, 2 no MFMAs, 4 no barrier, 8 / 16 no A / W
fragment reads -- results are wrong with any bit set).  One process per value: LIDIFF_S3_ABLATE=n python tools/debug/s3_ablate.py"""
import argparse, os, subprocess, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--level", type=int, default=3); ap.add_argument("--cin", type=int, default=256); ap.add_argument("--cout", type=int, default=256)
ap.add_argument("--sigma", type=float, default=1.0); ap.add_argument("--sorted", type=int, default=1)
a = ap.parse_args()
from lidiff_amd import _lib
from lidiff_amd.csrc import build as _build
csrc = os.path.join(ROOT, "lidiff_amd", "csrc")
lib = os.path.join(csrc, "liblidiff_amd_probe.so")
srcs = [os.path.join(csrc, f) for f in _build.SOURCES]
if not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(f) for f in srcs + [os.path.join(csrc, "spconv.h")]):
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-DLIDIFF_CONV_PROBE", "-shared"] + srcs + ["-o", lib], check=True)
_lib.LIB_PATH = lib
from lidiff_amd import ops
import lidiff_amd.MinkowskiEngine as ME
from tools.split3_table import timed
dev = torch.device("cuda:0")
scan = np.load(os.path.join(ROOT, "tests", "golden", "scan_000123_fps18000.npy"))
rng = np.random.default_rng(0)
pts = np.tile(scan, (10, 1)) + a.sigma * rng.standard_normal((180000, 3)).astype(np.float32)
feats = torch.from_numpy(pts.astype(np.float32)).to(dev)
coord = torch.cat([torch.zeros(180000, 1, device=dev), torch.round(feats / 0.05)], 1)
field = ME.TensorField(features=feats, coordinates=coord, device=dev); field.sparse()
mgr = field.coordinate_manager
ts = 1
for _ in range(4): ts = mgr.stride(ts, 2)
ts = 1 << a.level
nbr = mgr.kernel_map(ts, ts, 3); m = nbr.shape[1]
order = None
if a.sorted: nbr, order = ops.mask_sorted_map(nbr)
x = torch.randn(2 * m, a.cin, device=dev); w = torch.randn(27, a.cin, a.cout, device=dev) * 0.05
x3 = ops.split3_rows(x)
t = timed(lambda: ops.spconv_fwd_split3(x3, w, nbr, m, replicas=2, row_order=order))
abl = int(os.environ.get("LIDIFF_S3_ABLATE", "0"))
names = [n for b, n in ((1, "no requests"), (2, "no MFMAs"), (4, "no barrier"), (8, "no A reads"), (16, "no W reads")) if abl & b]
print(f"level {a.level} {a.cin}->{a.cout} sorted {a.sorted} ablate={abl:2d} {t:8.1f} us  ({', '.join(names) or 'the kernel'})", flush=True)
