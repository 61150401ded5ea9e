#!/usr/bin/env python
"""Per-stage cycle budget of spconv_fwd_split3_kernel (probe build, -DLIDIFF_CONV_PROBE): waves 0 and 4 of every workgroup stamp
the phases of the stage loop.  python tools/debug/s3_timeline.py [--level 3 --cin 256 --cout 256 --sigma 1.0 --sorted 1]"""
import argparse, ctypes, os, subprocess, sys
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
for _ in range(3): ops.spconv_fwd_split3(x3, w, nbr, m, replicas=2, row_order=order)
tl = torch.zeros(1 << 14, 2, 10, dtype=torch.int64, device=dev)
_lib.load().lidiff_debug_set_split3_timeline(ctypes.c_void_p(tl.data_ptr()))
ops.spconv_fwd_split3(x3, w, nbr, m, replicas=2, row_order=order)
torch.cuda.synchronize()
_lib.load().lidiff_debug_set_split3_timeline(ctypes.c_void_p(0))
t = tl.cpu().numpy().astype(np.float64); t = t[t[:, 0, 9] > 0]
st = t[:, 0, 7].mean()
print(f"level {a.level} {a.cin}->{a.cout} sigma {a.sigma} sorted {a.sorted}: workgroups {len(t)}, stages/tile {st:.0f}, loop cycles/tile {t[:, 0, 0].mean():.0f}")
for wv, name in ((0, "wave 0"), (1, "wave 4")):
    q = t[:, wv]
    print(f"  {name}: per stage {q[:, 0].mean() / st:.0f} = barrier {q[:, 1].mean() / st:.0f} + head {q[:, 2].mean() / st:.0f} + requests {q[:, 3].mean() / st:.0f} + "
          f"W reads {q[:, 4].mean() / st:.0f} + reads+MFMAs {q[:, 5].mean() / st:.0f} + fold {q[:, 6].mean() / st:.0f}")
