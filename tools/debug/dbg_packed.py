import sys, numpy as np, torch
import os; R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from conftest import random_cloud
from oracle import me_cpu as me
from lidiff_amd import ops
dev = torch.device("cuda:0")
for (n, ext, cin, cout) in [(300, 40, 32, 32), (3000, 40, 32, 32), (3000, 40, 64, 128), (3000, 40, 96, 96)]:
    coords = random_cloud(n, ext, 3, dup=0.05)
    uniq, _, _ = me.voxelize(coords)
    nbr = me.kernel_map(uniq, uniq, 3, 1)
    M = uniq.shape[0]
    g = torch.Generator().manual_seed(0)
    x = torch.randn(M, cin, generator=g); w = torch.randn(27, cin, cout, generator=g) * 0.1
    want = me.conv_forward(x.double(), w.double(), nbr)
    got = ops.spconv_fwd(x.to(dev), w.to(dev), torch.from_numpy(nbr).to(dev), M, sparse_map=True).cpu().double()
    err = (got - want).abs()
    bad_rows = torch.nonzero(err.max(1).values > 1e-3).squeeze(1)
    print(f"n={n} M={M} {cin}->{cout}: max err {err.max():.3e}, bad rows {bad_rows.numel()}/{M}; pairs/row {(nbr>=0).sum()/M:.2f}")
    if bad_rows.numel():
        r = bad_rows[:8].tolist()
        print("  first bad rows", r, "nbr counts", [(nbr[:, i] >= 0).sum() for i in r], "tile", [i // 128 for i in r])
        # test hypothesis: got = want - contribution of some offsets
        i = r[0]
        ks = np.nonzero(nbr[:, i] >= 0)[0]
        for k in ks:
            contrib = x[nbr[k, i]].double() @ w[k].double()
            print("   k", k, "in row", nbr[k, i], "resid if missing:", float((got[i] - (want[i] - contrib)).abs().max()))
        badcols = torch.nonzero(err[i] > 1e-3).squeeze(1).tolist()
        print("   bad cols of row", i, badcols[:40])
