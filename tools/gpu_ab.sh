#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
python - <<'PY' 2>&1 | grep -v amdgpu
import numpy as np, torch, os, sys
sys.path.insert(0, ".")
import bench
from lidiff_amd import ops
import lidiff_amd.MinkowskiEngine as ME
dev = torch.device("cuda:0")
scan = bench.load_scan()
for sigma in (1.0, 0.2):
    rng = np.random.default_rng(0)
    pts = np.tile(scan, (10, 1)) + sigma * rng.standard_normal((180000, 3)).astype(np.float32)
    def mgr_of(p):
        f = torch.from_numpy(p.astype(np.float32)).to(dev)
        c = torch.cat([torch.zeros(len(p), 1, device=dev), torch.round(f / 0.05)], 1)
        fld = ME.TensorField(features=f, coordinates=c, device=dev); fld.coordinate_manager.pyramid = True; fld.sparse()
        return fld.coordinate_manager
    mf, mp = mgr_of(pts), mgr_of(scan)
    part = mp.maps[16].coords
    for edge in (32, 64, 128):
        cells = ops.MatchCells(part, edge)
        for ts in (1, 4, 16):
            full = mf.maps[ts].coords
            for name, fn in (("exhaustive", lambda: ops.nn_match(full, part)), (f"cells{edge}", lambda: ops.nn_match(full, part, cells=cells))):
                if name == "exhaustive" and edge != 32: continue
                fn(); torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(10): fn()
                e.record(); torch.cuda.synchronize()
                print(f"sigma {sigma} stride {ts:2d} full {full.shape[0]:6d} part {part.shape[0]} {name:>10}: {s.elapsed_time(e) * 100:.1f} us")
PY
for V in 0 64; do LIDIFF_MATCH_CELL_EDGE=$V timeout 300 python tools/train_probe.py --steps 4 --precision bf16 2>&1 | grep -v amdgpu | tail -1; done
