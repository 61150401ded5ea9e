import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import build_seeded_models
from lidiff_amd.pipeline import DiffCompletion
from lidiff_amd import ops
dev = torch.device("cuda:0")
enc, unet, refine = build_seeded_models(42)
enc, unet, refine = enc.to(dev).eval(), unet.to(dev).eval(), refine.to(dev).eval()
fps = np.load(os.path.join(ROOT, "tests", "golden", "scan_000123_fps18000.npy"))
scan = torch.from_numpy(np.tile(fps, (10, 1))).double()[None].to(dev)
g = torch.Generator(device="cpu").manual_seed(5)
x0 = (scan.cpu() + torch.randn(scan.shape, generator=g, dtype=torch.float64)).to(dev)
zs = [torch.randn(scan.shape, generator=g, dtype=torch.float64).to(dev) for _ in range(4)]
def run(overlap, split3=True, srt=True):
    ops.SPLIT3, ops.SPLIT3_SORTED = split3, srt
    pipe = DiffCompletion(denoising_steps=4, cond_weight=6.0, device=dev)
    pipe.partial_enc, pipe.model, pipe.model_refine = enc, unet, refine
    pipe.overlap_maps = overlap
    pipe.new_scheduler()
    with torch.no_grad():
        o = pipe.completion_loop(scan, pipe.points_to_tensor(x0), pipe.points_to_tensor(scan), pipe.points_to_tensor(torch.zeros_like(scan)), noises=zs)
    torch.cuda.synchronize()
    return o
import warnings
for name, kw, pre in (("presort on the map lane", {}, True),):
    ops.SPLIT3_PRESORT = pre
    ref = run(False, **kw)
    for i in range(24):
        with warnings.catch_warnings(record=True) as wl:
            warnings.simplefilter("always")
            o = run(True, **kw)
        nd = int((o != ref).any(1).sum())
        if nd or wl:
            print("run", i, "points differing", nd, "warnings:", [str(w.message)[:160] for w in wl], flush=True)
    print("done")
