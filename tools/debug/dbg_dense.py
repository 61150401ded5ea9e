"""dense vs tile kernel on a few shapes (debug aid): prints max |dense - tile| and where the rows differ."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import random_cloud
from oracle import me_cpu as me
from lidiff_amd import ops
dev = torch.device("cuda:0")
coords = random_cloud(2000, 5, 23)
uniq, _, _ = me.voxelize(coords)
nbr = torch.from_numpy(me.kernel_map(uniq, uniq, 3, 1)).to(dev)
m = uniq.shape[0]
g = torch.Generator().manual_seed(0)
for cin, split, cout in [(192, 0, 128), (192, 128, 128), (192, 64, 128), (256, 128, 128), (128, 64, 128), (384, 256, 256), (64, 0, 128), (64, 32, 128)]:
    x = torch.randn(m, cin, generator=g).to(dev)
    w = (torch.randn(27, cin, cout, generator=g) * 0.05).to(dev)
    a = x[:, :split].contiguous() if split else x
    b = x[:, split:].contiguous() if split else None
    for rep in range(2):
        d = ops.spconv_fwd(a, w, nbr, m, in_b=b, kernel="dense")
        t = ops.spconv_fwd(a, w, nbr, m, in_b=b, kernel="tile")
        err = (d - t).abs()
        bad = (err.max(1).values > 1e-4).nonzero().flatten()
        print(f"cin={cin} split={split} cout={cout} run{rep}: max err {err.max().item():.3e}  bad rows {bad.numel()}/{m}"
              f"  first bad {bad[:8].tolist()}  bad cols {(err.max(0).values > 1e-4).nonzero().flatten()[:8].tolist()}", flush=True)
